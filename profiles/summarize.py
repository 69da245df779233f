"""Turn a gpurun_out/prof_<tag>/ directory (made by profiles/collect.sh) into the committed
summaries: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json and
profiles/traffic.json (read by bench.py for roofline.traffic).

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reads exactly 1/2 of a coalesced stream.
The factor is calibrated in the same run on k_camera_fwd, a kernel with a known byte count and
the same 12-byte-per-lane access pattern (reads 2 x [1,1080,1920,3] fp32, writes the same)."""
import csv
import glob
import json
import os
import shutil


def newest(pattern):
    """gpurun merges every run into gpurun_out/: take the latest file of a kind"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]

import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
out = os.path.join(root, "profiles")

stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, f"{tag}_bench_under_rocprof.json"))


def counters(sub):
    f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    agg = defaultdict(list)
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        key = "k_render_fwd" if "k_render_fwd" in name else ("k_camera_fwd" if "k_camera_fwd" in name else None)
        if key:
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v)


fetch, write, sq = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
KNOWN = 2 * 1080 * 1920 * 3 * 4            # bytes read (and written) by k_camera_fwd
cal_fetch = mean(fetch[("k_camera_fwd", "FETCH_SIZE")]) * 1024
cal_write = mean(write[("k_camera_fwd", "WRITE_SIZE")]) * 1024
fetch_factor = KNOWN / cal_fetch            # ~2.0 on gfx950
write_factor = KNOWN / cal_write            # ~1.0
r_fetch = mean(fetch[("k_render_fwd", "FETCH_SIZE")]) * 1024 * fetch_factor
r_write = mean(write[("k_render_fwd", "WRITE_SIZE")]) * 1024 * write_factor
summary = {
    "tag": tag,
    "calibration_kernel": "k_camera_fwd (49766400 B read, 49766400 B written, float3 per lane)",
    "FETCH_SIZE_correction": fetch_factor, "WRITE_SIZE_correction": write_factor,
    "k_render_fwd": {"read_bytes_per_launch": r_fetch, "written_bytes_per_launch": r_write,
                     "hbm_bytes_per_launch": r_fetch + r_write,
                     "algorithmic_bytes_per_launch": 1920 * 1080 * 36,
                     "traffic_over_algorithmic": (r_fetch + r_write) / (1920 * 1080 * 36)},
}
for name in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    v = sq.get(("k_render_fwd", name))
    if v:
        summary["k_render_fwd"][name] = mean(v)
k = summary["k_render_fwd"]
if "SQ_INSTS_VALU" in k and "GRBM_GUI_ACTIVE" in k:
    cycles = k["GRBM_GUI_ACTIVE"] / 8          # counter sums the 8 XCDs
    k["gpu_cycles_per_launch"] = cycles
    k["valu_wave_instructions_per_simd_cycle"] = k["SQ_INSTS_VALU"] / 1024 / cycles
    k["valu_issue_utilisation"] = 2 * k["SQ_INSTS_VALU"] / 1024 / cycles   # wave64 on SIMD32 = 2 cycles/instr
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
sys.path.insert(0, root)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("_rm_build", os.path.join(root, "ray_marching_amd", "_build.py"))
_build = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_build)
json.dump({"source": f"profiles/{tag}_pmc_summary.json",
           # bench.py only quotes these numbers while the kernel sources are the ones they were measured on
           "sources_hash": _build.sources_hash(),
           "k_render_fwd_hbm_bytes_per_launch": r_fetch + r_write,
           "k_render_fwd_valu_wave_instructions_per_launch": k.get("SQ_INSTS_VALU")},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
