#!/bin/bash
# A/B one environment switch on the SAME box: tools/debug/ab_env.sh VAR "v0 v1" -- <bench.py flags>
VAR=$1; VALS=$2; shift 3
for round in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-x1 --no-attainable --no-train "$@" 2>/dev/null | V="$VAR=$v" python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); tot = sum(k['ms'] for k in d['kernels'].values())
        parts = ['%s %.4f' % (n.replace('_kernel', ''), k['ms'] / k['calls']) for n, k in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms']) if k['ms'] > 0.03 * tot]
        print(os.environ['V'], 'step %.2f ms, value %d, partition %.1f s |' % (d['ms_per_step'], d['value'], d['config'].get('partition_build_s', 0)), '; '.join(parts))"
  done
done
