# Same-box A/B against the round-1 tree: first `git worktree add _ab_r1 240ae82` (or copy that commit's tree into _ab_r1/,
# which .gitignore excludes), build it there (python -c 'import __graft_entry__ as g; g.build()'), then run this on the GPU box.
for i in 1 2 3; do
 for d in . _ab_r1; do
  (cd $d && python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', 'value',round(j['value']),'kernel_ms',round(j['roofline']['kernel_ms'],4),'other',round(j['other_camera']['value']),'fwdbwd',round(j['fwd_bwd']['fwd_bwd_ms'],4), 'graph', round(j['fwd_bwd']['graph_fwd_bwd_ms'],4))
")
 done
done
