#!/bin/bash
# A/B two versions of gossip_fused.hip on the same box (box-to-box clocks differ by ~2 %):
# tools/debug/_ab/{base,new}.hip, alternating twice; prints ms per launch from bench.py --by-shape.
for round in 1 2; do
  for v in base new; do
    cp tools/debug/_ab/$v.hip desco_amd/csrc/gossip_fused.hip
    make -C desco_amd/csrc > /dev/null 2>&1
    python bench.py --by-shape --steps 4 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']['gossip_fused_kernel']; print('$v', round(k['ms'] / k['calls'], 4), 'ms/launch; step', round(d['ms_per_step'], 2))"
  done
done
