for i in 1 2 3; do
 for d in . _ab_r1; do
  (cd $d && python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', 'value',round(j['value']),'kernel_ms',round(j['roofline']['kernel_ms'],4),'other',round(j['other_camera']['value']),'fwdbwd',round(j['fwd_bwd']['fwd_bwd_ms'],4), 'graph', round(j['fwd_bwd']['graph_fwd_bwd_ms'],4))
")
 done
done
