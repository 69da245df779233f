#!/bin/bash
# Average latency of vector-memory, scalar-memory and LDS instructions of the training-step kernels:
# SQ_INST_LEVEL_* accumulates the number of instructions in flight per cycle, so LEVEL / INSTS = cycles an instruction is outstanding.
#   bash profiles/pmc_latency.sh <tag> [size]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_lat_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/vmem -- python3 $ROOT/profiles/train_driver.py ${2:-512} 6 > $OUT/vmem.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/smem -- python3 $ROOT/profiles/train_driver.py ${2:-512} 6 > $OUT/smem.log 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].replace("rm::StaticCfg<RmStaticCode, ", "S<").replace("rm::", "")
        if k.startswith("k_render") or k.startswith("k_bwd"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    vm = m.get("SQ_INSTS_VMEM_RD", 0) + m.get("SQ_INSTS_VMEM_WR", 0)
    print(f"{k[:44]:44s} VMEM {vm/1e3:8.1f}k instr, {m.get('SQ_INST_LEVEL_VMEM',0)/max(vm,1):8.0f} cycles each | SMEM {m.get('SQ_INSTS_SMEM',0)/1e3:7.1f}k, {m.get('SQ_INST_LEVEL_SMEM',0)/max(m.get('SQ_INSTS_SMEM',1),1):7.0f} cycles | LDS {m.get('SQ_INSTS_LDS',0)/1e3:7.1f}k, {m.get('SQ_INST_LEVEL_LDS',0)/max(m.get('SQ_INSTS_LDS',1),1):6.0f} cycles | wave-cycles {m.get('SQ_WAVE_CYCLES',0)/1e6:.1f}M")
PY
