#!/bin/bash
# A/B of an environment switch on one box:  ab_env.sh VAR "v1 v2 .." "<workload> <replicas>" ...
VAR=$1; VALS=$2; shift; shift
for WL in "$@"; do set -- $WL
for rep in 1 2; do for V in $VALS; do
  export $VAR=$V
  python bench.py --workload $1 --replicas $2 --steps 5 --warmup 2 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 x$2 $VAR=$V', round(d['value']), round(d['ms_per_step'],2), [(k, round(v['ms']/v['calls'],3)) for k,v in d['kernels'].items() if 'shmp' in k and '3,2' in k])"
done; done; done
