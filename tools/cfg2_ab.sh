# cfg2 iteration time: rows of M per workgroup of the one-launch iteration, graph replay vs direct launches, two-launch iteration
for env in "LPVS_NO_GRAPH=0" "LPVS_NO_GRAPH=1" "LPVS_ITERATION=two"; do
  echo "== $env"
  env $env python bench.py --workload cfg2 --no-cpu-baseline --no-concurrent --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('signals/s %.2f  iteration_us %.3f  kernel %s' % (d['value'], d['roofline']['iteration_us'], d['roofline']['kernel'][:30]))"
done
