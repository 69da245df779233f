set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_auto; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/regen_driver.py -3 auto 64 > $OUT/log.txt 2>&1
cd $ROOT
f=$(ls $OUT/trace/*/*_kernel_stats.csv | head -1); cut -d, -f1-4 $f | sed -n 1,14p
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_auto/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# timeline of the last 40 kernels: start offset (us), duration, name
prev_end = None
for r in rows[-70:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:60]}")
    prev_end = e
PY
