#!/bin/bash
# A/B of alternative builds of the library on one box:  ab_lib.sh "<lib suffixes, 'base' = the shipped one>" "<workload> <replicas>" ...
LIBS=$1; shift
cat > /tmp/_ab_fmt.py <<'PY'
import sys, json
tag = sys.argv[1]
d = json.loads(sys.stdin.read())
ks = [(k, round(v['ms'] / v['calls'], 3)) for k, v in d['kernels'].items()
      if ('shmp' in k and '3,2' in k) or 'gossip_f' in k or 'gemm_f16' in k]
print(tag, round(d['value']), round(d['ms_per_step'], 2), ks)
PY
for WL in "$@"; do set -- $WL
for rep in 1 2; do for LIBV in $LIBS; do
  if [ $LIBV = base ]; then unset DESCO_LIB; else export DESCO_LIB=$PWD/desco_amd/libdesco_$LIBV.so; fi
  python bench.py --workload $1 --replicas $2 --steps 5 --warmup 2 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train 2>/dev/null | tail -1 | python /tmp/_ab_fmt.py "$1 x$2 $LIBV"
done; done; done
