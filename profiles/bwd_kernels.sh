#!/bin/bash
# per-kernel durations of the config-4 training step (rocprofv3 --kernel-trace --stats)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_bwd_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/profiles/bwd_probe.py > $OUT/log.txt 2>&1
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
