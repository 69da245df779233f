#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root:  bash profiles/collect_r03.sh [tag]
#   backward (config-4 step): kernel trace at 512^2 and 2048^2, FETCH_SIZE / WRITE_SIZE / SQ passes at 512^2 and 2048^2
#   config 5 (one 7680x540 band): kernel trace, FETCH / WRITE / SQ passes, fp64 instruction counters
# Counter passes are their own runs (--pmc only); summarised by profiles/summarize_r03.py
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU"
for size in 512 2048; do
  n=12; [ $size = 2048 ] && n=4
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bwd${size}_trace -- python3 $ROOT/profiles/train_driver.py $size $n > $OUT/bwd${size}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bwd${size}_fetch -- python3 $ROOT/profiles/train_driver.py $size $n > $OUT/bwd${size}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bwd${size}_write -- python3 $ROOT/profiles/train_driver.py $size $n > $OUT/bwd${size}_write.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/bwd${size}_sq -- python3 $ROOT/profiles/train_driver.py $size $n > $OUT/bwd${size}_sq.log 2>&1
  echo "bwd $size done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -- python3 $ROOT/profiles/config5_driver.py 540 3 0 > $OUT/c5_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -- python3 $ROOT/profiles/config5_driver.py 540 2 0 > $OUT/c5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -- python3 $ROOT/profiles/config5_driver.py 540 2 0 > $OUT/c5_write.log 2>&1
rocprofv3 --pmc $SQ --output-format csv -d $OUT/c5_sq -- python3 $ROOT/profiles/config5_driver.py 540 2 0 > $OUT/c5_sq.log 2>&1
rocprofv3 --pmc $F64 --output-format csv -d $OUT/c5_f64 -- python3 $ROOT/profiles/config5_driver.py 540 2 0 > $OUT/c5_f64.log 2>&1
echo "config 5 done"
cd $ROOT
find $OUT -name "*.csv" | wc -l
