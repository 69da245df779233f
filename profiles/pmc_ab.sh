#!/bin/bash
# instruction counters of the frame kernel for the current tree and the _ab_r1 copy (same box)
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for d in cur r1; do
  SRC=$ROOT; [ $d = r1 ] && SRC=$ROOT/_ab_r1
  OUT=$ROOT/gpurun_out/pmc_ab_$d
  mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 $SRC/profiles/pmc_driver.py > $OUT/log.txt 2>&1 || true
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for d in ("cur", "r1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_ab_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "render" in k:
            print(d, k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
