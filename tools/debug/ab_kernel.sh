#!/bin/bash
# A/B two versions of one kernel source on the SAME box (box-to-box clocks differ by a few percent):
#   tools/debug/ab_kernel.sh <file under desco_amd/csrc> '<kernel key of bench.py --by-shape>'
# expects tools/debug/_ab/{base,new}.hip (e.g. `git show HEAD:desco_amd/csrc/x.hip > tools/debug/_ab/base.hip`),
# alternates them twice and prints ms per launch of that kernel and ms per step.
F=$1; K=$2
for round in 1 2; do
  for v in base new; do
    cp tools/debug/_ab/$v.hip desco_amd/csrc/$F
    make -C desco_amd/csrc > /dev/null 2>&1
    python bench.py --by-shape --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | K="$K" V=$v python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels'][os.environ['K']]
        print(os.environ['V'], round(k['ms'] / k['calls'], 4), 'ms/launch; step', round(d['ms_per_step'], 2))"
  done
done
cp tools/debug/_ab/new.hip desco_amd/csrc/$F; make -C desco_amd/csrc > /dev/null 2>&1
