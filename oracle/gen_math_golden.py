"""Exhaustive sweep of the path's transcendental functions: oracle/rm_math_ref.c against torch CPU.

For exp, log and x.pow(1/2.33) every one of the 2^32 fp32 bit patterns is evaluated; for atan2, 2^32 hashed
(y, x) pairs.  Per block of 2^24 inputs the fixture tests/golden/math_sweep.json records
  ours      checksum of the restatement (== the device function, tests/test_math_sweep.py -m gpu)
  torch     checksum of torch's CPU result on the generating host
  n_diff    inputs on which the two differ,  max_ulp  their largest distance in ulps
Run in the build container:  python oracle/gen_math_golden.py [--out tests/golden/math_sweep.json]
(about 10 minutes on 8 cores).  Test infrastructure only."""
from __future__ import annotations

import argparse
import json
import os
import platform
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import math_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "math_sweep.json"))
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--only", nargs="*", help="regenerate these functions only, keep the rest of the file")
    args = ap.parse_args()
    math_ref.build(force=True)
    cpu = subprocess.run("lscpu | grep 'Model name' | cut -d: -f2", shell=True, capture_output=True, text=True).stdout.strip()
    out = {"host": {"cpu": cpu, "torch": torch.__version__, "capability": torch.backends.cpu.get_cpu_capability(),
                    "machine": platform.machine()},
           "block": math_ref.BLOCK, "checksum": "sum canon(out_bits) * (2 i + 1) mod 2^64", "functions": {}}
    if args.only and os.path.isfile(args.out):
        with open(args.out) as f:
            out["functions"] = json.load(f)["functions"]
    for fn in (args.only or math_ref.FN):
        t0 = time.time()
        rows = {"ours": [], "torch": [], "n_diff": [], "max_ulp": []}
        for blk in range(args.blocks):
            a, b = math_ref.sweep_inputs(fn, blk)
            ref = math_ref.torch_eval(fn, a, b).contiguous()
            s, osum, nd, mu = math_ref.sweep_block(fn, blk, ref)
            rows["ours"].append(f"{s:016x}"); rows["torch"].append(f"{osum:016x}")
            rows["n_diff"].append(nd); rows["max_ulp"].append(mu)
        if fn in math_ref.CHECKED_BLOCKS:
            rows["checked_blocks"] = math_ref.CHECKED_BLOCKS[fn]
        rows["total_diff"] = sum(rows["n_diff"])
        rows["worst_ulp"] = max(rows["max_ulp"])
        out["functions"][fn] = rows
        print(f"{fn}: {rows['total_diff']} of {args.blocks * math_ref.BLOCK} inputs differ from torch on this host, "
              f"worst {rows['worst_ulp']} ulp  ({time.time() - t0:.0f} s)", flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.abspath(args.out))


if __name__ == "__main__":
    main()
