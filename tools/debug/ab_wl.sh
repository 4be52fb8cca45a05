#!/bin/bash
# variants tools/debug/_ab/<name>.hip of one kernel source on several workloads (same box)
F=$1; shift
for v in "$@"; do
  cp tools/debug/_ab/$v.hip desco_amd/csrc/$F; make -C desco_amd/csrc > /dev/null 2>&1
  for wl in "cox2 64" "msrc_imdb 8" "syn_1827 2"; do
    set -- $wl
    python bench.py --workload $1 --replicas $2 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | V=$v W=$1 python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']['shmp_layer_f32_kernel<3,2,x6>']
        print(os.environ['V'], os.environ['W'], 'shmp<3,2>', round(k['ms'] / d['steps'], 2), 'ms/step; step', round(d['ms_per_step'], 2))"
  done
done
cp tools/debug/_ab/base.hip desco_amd/csrc/$F; make -C desco_amd/csrc > /dev/null 2>&1
