#!/bin/bash
# per-kernel durations of the config-4 training step without an optimiser (profiles/train_driver.py):  bash profiles/trace_train.sh <tag> [size]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_train_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/profiles/train_driver.py ${2:-512} 24 > $OUT/log.txt 2>&1
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0.0
for r in rows[:12]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
