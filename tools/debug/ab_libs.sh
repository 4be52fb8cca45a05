#!/bin/bash
# A/B whole-library builds on the SAME box: tools/debug/ab_libs.sh A B C ... [-- extra bench.py flags]
# expects tools/debug/_ab/lib<name>.so (built in the container with different flags, or from patched copies of a kernel
# source as tools/debug/gf16_hazard/make_ablations.py does; such timing-only builds are compiled with
# -DDESCO_DEBUG_ABLATION, which makes the library report a debug ABI version -- accepted here only); alternates
# them twice through bench.py (DESCO_LIB) and prints ms per launch of every kernel above 2 % plus ms per step.
LIBS=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; LIBS+=("$1"); shift; done
for round in 1 2; do
  for v in "${LIBS[@]}"; do
    DESCO_ALLOW_DEBUG_LIB=1 DESCO_LIB=$PWD/tools/debug/_ab/lib$v.so python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary \
      --no-x1 --no-attainable --no-train "${EXTRA[@]}" 2>/dev/null | V=$v python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); tot = sum(k['ms'] for k in d['kernels'].values())
        parts = ['%s %.4f' % (n.split('<')[0].replace('_kernel', '') + ('<' + n.split('<')[1] if '<' in n else ''), k['ms'] / k['calls'])
                 for n, k in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms']) if k['ms'] > 0.02 * tot]
        print(os.environ['V'], 'step %.2f ms, value %d |' % (d['ms_per_step'], d['value']), '; '.join(parts))"
  done
done
