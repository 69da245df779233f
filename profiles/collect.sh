#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh <tag>
# 1. kernel-trace + stats of the bench's timed workload only (warm-up + timed 1080p launches; the
#    secondary probes -- backward, pipelined, CPU baseline -- are switched off so that the per-kernel
#    average is comparable with roofline.kernel_ms)
# 2. PMC passes (separate runs, --pmc only): FETCH_SIZE, WRITE_SIZE
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --skip-backward --no-pipelined > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/profiles/pmc_driver.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/profiles/pmc_driver.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/profiles/pmc_driver.py > $OUT/pmc_sq.log 2>&1 || true
cd $ROOT
find $OUT -name "*.csv" | head -20
