#!/bin/bash
# Where do the waves of the training-step kernels wait?  I-cache hits / misses and wait cycles per kernel:
#   bash profiles/pmc_stalls.sh <tag> [size]       (summary on stdout; counters in their own --pmc passes)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_stalls_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/icache -- python3 $ROOT/profiles/train_driver.py ${2:-512} 6 > $OUT/icache.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/wait -- python3 $ROOT/profiles/train_driver.py ${2:-512} 6 > $OUT/wait.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_BUSY_CYCLES --output-format csv -d $OUT/mem -- python3 $ROOT/profiles/train_driver.py ${2:-512} 6 > $OUT/mem.log 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        k = k.replace("rm::StaticCfg<RmStaticCode, 64>", "S").replace("rm::", "")
        if k.startswith("k_render") or k.startswith("k_bwd"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    line = f"{k:28s}"
    if "SQC_ICACHE_REQ" in m:
        line += f" icache req {m['SQC_ICACHE_REQ']/1e6:7.2f}M hit {m.get('SQC_ICACHE_HITS',0)/max(m['SQC_ICACHE_REQ'],1):.3f} miss {m.get('SQC_ICACHE_MISSES',0)/1e6:6.2f}M dup {m.get('SQC_ICACHE_MISSES_DUPLICATE',0)/1e6:6.2f}M |"
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        line += f" wave-cycles {wc/1e6:8.1f}M wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} act_valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} act_sca {m.get('SQ_ACTIVE_INST_SCA',0)/wc:.2f} VALU {m.get('SQ_INSTS_VALU',0)/1e6:.1f}M SALU {m.get('SQ_INSTS_SALU',0)/1e6:.1f}M SMEM {m.get('SQ_INSTS_SMEM',0)/1e6:.2f}M |"
    if "SQ_INSTS_LDS" in m:
        line += f" LDS {m['SQ_INSTS_LDS']/1e6:.2f}M VMEM rd {m.get('SQ_INSTS_VMEM_RD',0)/1e6:.2f}M wr {m.get('SQ_INSTS_VMEM_WR',0)/1e6:.2f}M ifetch {m.get('SQ_IFETCH',0)/1e6:.2f}M wait_lds {m.get('SQ_WAIT_INST_LDS',0)/1e6:.1f}M"
    print(line)
PY
