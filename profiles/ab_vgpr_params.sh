#!/bin/bash
# A/B of where the specialised frame kernel keeps the scene parameters: SGPRs (readfirstlane hoist,
# the default; an SGPR source makes its consumer a half-rate instruction, profiles/r03_valu_issue_bench.txt) against
# VGPR-resident copies (-DRM_VGPR_PARAMS).  Frame times at both config-2 poses, then executed VALU instructions and
# busy cycles of k_render_fwd from a --pmc pass per variant and pose.   bash profiles/ab_vgpr_params.sh > gpurun_out/r03_ab_vgpr_params.txt
ROOT=$(pwd)
python3 profiles/ab_probe.py "sgpr_params(default)=" "vgpr_params=-DRM_VGPR_PARAMS"
cd /tmp && export TMPDIR=/tmp
for variant in base vgpr; do
  if [ $variant = vgpr ]; then export RM_HIPCC_EXTRA="-DRM_VGPR_PARAMS"; export RM_LIB_DIR=/tmp/rm_ab_vgpr_params; else export RM_HIPCC_EXTRA=""; export RM_LIB_DIR=/tmp/rm_ab_sgpr_params_default_; fi
  export RM_SPECIALIZE=jit
  for z in -3 1; do
    OUT=$ROOT/gpurun_out/prof_abvgpr/${variant}_z$z; mkdir -p $OUT
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $ROOT/profiles/pmc_driver.py $z > $OUT/log.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/pmc_driver.py $z > $OUT/trace.log 2>&1
    python3 - $OUT $variant $z <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
us = None
for f in glob.glob(sys.argv[1] + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_fwd" in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
cyc = us * 1e-6 * 2.4e9 * 1024 / m["SQ_INSTS_VALU"] if us and m.get("SQ_INSTS_VALU") else float("nan")
print(f"{sys.argv[2]:5s} z={sys.argv[3]:>2s}: k_render_fwd {us:.1f} us, VALU {m.get('SQ_INSTS_VALU', 0)/1e6:.2f} M wave-instr, SALU {m.get('SQ_INSTS_SALU', 0)/1e6:.2f} M, {cyc:.2f} SIMD-cycles per VALU instruction")
PY
  done
done
