#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root:  bash profiles/collect_r02.sh
# (1) headline frame kernel: kernel trace + PMC passes (profiles/collect.sh), summarised into profiles/r02_*
# (2) config-4 training step: per-kernel stats + HBM traffic of the backward kernels
# (3) fp16 frame: kernel list of a RenderLoop.to(float16) frame (one frame kernel, no cast passes)
# (4) probes quoted in DESIGN.md: host math, fp64 issue cost, settle statistics, backward distribution
set -e
ROOT=$(pwd)
bash profiles/collect.sh r02 > gpurun_out/collect_r02.log 2>&1
python3 profiles/summarize.py r02 > gpurun_out/summarize_r02.log 2>&1 || true
OUT=$ROOT/gpurun_out/prof_r02b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/bwd_probe.py > $OUT/bwd.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/profiles/bwd_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/profiles/bwd_probe.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp16 -- python3 $ROOT/profiles/fp16_frame_driver.py > $OUT/fp16.log 2>&1
# (5) ray regeneration at the reference's pose (0,0,1): kernel stats of 64 frames with regen=True, HBM traffic and
#     instruction counters of its kernels, and the timeline of the default regen="auto" loop in front of the scene
OUTC=$ROOT/gpurun_out/prof_r02c; mkdir -p $OUTC
rocprofv3 --kernel-trace --stats --output-format csv -d $OUTC/trace -- python3 $ROOT/profiles/regen_driver.py 1 1 64 > $OUTC/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUTC/pmc_fetch -- python3 $ROOT/profiles/regen_driver.py 1 1 24 > $OUTC/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUTC/pmc_write -- python3 $ROOT/profiles/regen_driver.py 1 1 24 > $OUTC/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUTC/pmc_sq -- python3 $ROOT/profiles/regen_driver.py 1 1 24 > $OUTC/pmc_sq.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUTC/auto -- python3 $ROOT/profiles/regen_driver.py -3 auto 128 > $OUTC/auto.log 2>&1
cd $ROOT
python3 profiles/regen_probe.py default= > gpurun_out/r02_regen_probe.txt 2>&1 || true
python3 profiles/regen_stale_probe.py > gpurun_out/r02_regen_stale_probe.txt 2>&1 || true
python3 profiles/host_math_probe.py > gpurun_out/r02_host_math_gpu_box.txt 2>&1 || true
./profiles/micro/f64_issue_bench > gpurun_out/r02_f64_issue_bench.txt 2>&1
python3 profiles/settle_probe.py > gpurun_out/r02_settle_probe.txt 2>&1 || true
python3 profiles/bwd_dist_probe.py > gpurun_out/r02_bwd_dist_probe.txt 2>&1 || true
python3 profiles/config5_probe.py > gpurun_out/r02_config5_probe.txt 2>&1 || true
find gpurun_out/prof_r02 gpurun_out/prof_r02b -name "*.csv" | head -40
