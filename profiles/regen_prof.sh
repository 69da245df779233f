set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_regen; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for z in 1 -3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_z$z -- python3 $ROOT/profiles/regen_driver.py $z > $OUT/log_z$z.txt 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_z1 -- python3 $ROOT/profiles/regen_driver.py 1 > $OUT/pmc_z1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_z1_tile -- python3 $ROOT/profiles/regen_driver.py 1 0 > $OUT/pmc_z1_tile.txt 2>&1
cd $ROOT
for z in 1 -3; do f=$(ls $OUT/trace_z$z/*/*_kernel_stats.csv | head -1); echo "== z=$z"; cut -d, -f1-4 $f | sed -n 1,8p; done
python3 - <<'PY'
import csv, glob, collections
for tag in ("pmc_z1", "pmc_z1_tile"):
    f = glob.glob(f"gpurun_out/prof_regen/{tag}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("k_march_regen", "k_render_finish", "k_render_fwd"):
            if k in n: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(tag, {f"{k[0]}.{k[1]}": round(sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
