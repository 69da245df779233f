#!/bin/bash
# Everything the round-3 numbers in DESIGN.md / profiles/ come from, in one GPU call:  bash profiles/collect_final_r03.sh
set -x
bash profiles/collect.sh r03 > gpurun_out/collect_r03_headline.log 2>&1
bash profiles/collect_r03.sh r03 > gpurun_out/collect_r03_bwd_c5.log 2>&1
bash profiles/pmc_stalls.sh r03 > gpurun_out/r03_stalls.txt 2>&1
python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python3 bench.py --config 5 --steps 5 --warmup 2 --repeats 3 > gpurun_out/r03_config5_1gpu.json 2> gpurun_out/r03_config5_1gpu.err; echo "bench c5 rc=$?"
python3 bench.py --gpus 2 --share-gpu --backend gloo --steps 5 --warmup 2 --skip-backward --no-pipelined > gpurun_out/r03_rehearsal_selflaunch_2rank.json 2> gpurun_out/r03_rehearsal_selflaunch_2rank.err; echo "selflaunch rc=$?"
timeout -k 10 500 python3 tests/fuzz_cull_lse.py 100 > gpurun_out/r03_fuzz_cull_lse.txt 2>&1; echo "fuzz rc=$?"
RM_CULL_LSE_MIN=2 timeout -k 10 500 python3 tests/fuzz_cull.py 200 > gpurun_out/r03_fuzz_cull.txt 2>&1; echo "fuzz cull rc=$?"
timeout -k 10 400 python3 profiles/scale_probe.py off 8 32 48 100 200 300 400 450 2>&1 | grep "^n=" > gpurun_out/r03_scale_probe.txt; echo "scale rc=$?"
timeout -k 10 200 python3 profiles/config5_kernels.py 2>&1 | grep " ms" > gpurun_out/r03_config5_kernels.txt; echo "c5 kernels rc=$?"
