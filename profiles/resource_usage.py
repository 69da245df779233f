"""Compiler-reported resources of the specialised kernels of a benchmark scene (no GPU needed):

    python profiles/resource_usage.py make_closed_test_scene > profiles/r03_resource_usage_closed_scene1.txt

Compiles the per-scene library exactly as ray_marching_amd/specialize.py does, plus
-Rpass-analysis=kernel-resource-usage, and prints VGPRs / AGPRs / SGPRs / scratch / occupancy / LDS per kernel."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_marching_amd import specialize  # noqa: E402
from ray_marching_amd.compiler import compile_scene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "make_closed_test_scene"
extra = sys.argv[2:]
cs = compile_scene(specialize.default_scenes()[name])
with tempfile.TemporaryDirectory() as tmp:
    header = os.path.join(tmp, "code.h")
    open(header, "w").write(specialize.code_header(cs))
    cmd = [specialize._hipcc(), *specialize.variant("exact")[1], f'-DRM_STATIC_CODE="{header}"',
           "-Rpass-analysis=kernel-resource-usage", *extra]
    if not specialize.static_backward(cs):
        cmd.append("-DRM_NO_BACKWARD")
    cmd += [os.path.join(specialize.CSRC, "rm_abi.hip"), "-o", os.path.join(tmp, "lib.so")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=specialize.CSRC)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+(?:\[.*?\])?): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print(f"# {name}: {cs.n_instr} instructions, {cs.n_params} parameters + {cs.n_derived} derived; flags {' '.join(specialize.variant('exact')[1] + extra)}")
print(f"# {'kernel':58s} VGPR AGPR SGPR  spillS spillV scratch[B]  occ[waves/SIMD]  LDS[B]")
for k in rows:
    short = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", short.replace("void ", "").replace("rm::", ""))
    short = re.sub(r"StaticCfg<RmStaticCode, (\d+), false>", r"S\1", short)
    short = re.sub(r"StaticCfg<RmStaticCode, (\d+), true>", r"S\1v", short)      # v: parameters in VGPRs
    if not short.startswith("k_"):
        continue
    print(f"{short[:60]:60s} {k.get('VGPRs', '?'):>4s} {k.get('AGPRs', '?'):>4s} {k.get('TotalSGPRs', '?'):>4s}  {k.get('SGPRs Spill', '?'):>6s} "
          f"{k.get('VGPRs Spill', '?'):>6s} {k.get('ScratchSize [bytes/lane]', '?'):>10s}  {k.get('Occupancy [waves/SIMD]', '?'):>15s}  {k.get('LDS Size [bytes/block]', '?'):>6s}")
