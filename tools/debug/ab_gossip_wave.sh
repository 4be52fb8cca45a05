#!/bin/bash
# A/B of the wave-autonomous gossip kernel against the block form:  ab_gossip_wave.sh "<workload> <replicas>" ...
for WL in "$@"; do
  set -- $WL
  for WV in 1 0 1 0; do
    DESCO_GOSSIP_WAVE=$WV python bench.py --workload $1 --replicas $2 --steps 5 --warmup 2 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 WAVE=$WV', round(d['value']), round(d['ms_per_step'],2), [(k, round(v['ms']/v['calls'],3)) for k,v in d['kernels'].items() if 'gossip' in k and 'scal' not in k])"
  done
done
