# Same-box A/B against the round-2 tree (git archive d9d0c53 | tar -x -C _ab_r2; build it there), alternating runs:
#   bash profiles/ab_r2.sh            headline kernel time, Mrays/s at both poses, config-4 graph step
# third leg: this tree with the tile kernel's parameters in VGPRs (-DRM_FWD_VGPR_PARAM_LIMIT=32)
line='
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=j.get("poses",[{},{}])
print(sys.argv[1].ljust(12), "value",round(j["value"]),"kernel_ms",round(j["roofline"]["kernel_ms"],4),"poses ms",[round(x.get("ms_per_frame",0),4) for x in p], "tile kernel", [round(x["fixed_choices"]["tile kernel"]["ms_per_frame"],4) for x in p], "graph fwd+bwd", round(j["fwd_bwd"]["graph_fwd_bwd_ms"],4))
'
ARGS="--steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --skip-config3"
for i in 1 2 3; do
  python3 bench.py $ARGS 2>/dev/null | python3 -c "$line" this_tree
  (cd _ab_r2 && python3 bench.py $ARGS 2>/dev/null | python3 -c "$line" round2_tree)
  RM_HIPCC_EXTRA=-DRM_FWD_VGPR_PARAM_LIMIT=32 RM_LIB_DIR=/tmp/rm_ab_vgpr_tile RM_SPECIALIZE=jit python3 bench.py $ARGS 2>/dev/null | python3 -c "$line" this_vgpr
done
