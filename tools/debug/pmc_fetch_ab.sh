#!/bin/bash
# FETCH_SIZE / TCC hit-miss of the Syn_1827 x2 layer kernel under an environment switch:  pmc_fetch_ab.sh VAR "v1 v2 .."
VAR=$1; VALS=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ab; mkdir -p $OUT
ARGS="--workload syn_1827 --replicas 2 --steps 2 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train --no-profile"
cd /tmp && export TMPDIR=/tmp
for V in $VALS; do
  export $VAR=$V
  i=0
  for P in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/${V}_p$i -- python3 $ROOT/bench.py $ARGS > $OUT/${V}_p$i.log 2>&1
  done
done
python3 - "$OUT" "$VALS" <<'PY'
import csv, glob, collections, sys
out, vals = sys.argv[1], sys.argv[2].split()
for v in vals:
    agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob(f"{out}/{v}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "shmp_layer16_kernel" in r["Kernel_Name"] and ", 2, " in r["Kernel_Name"].split("(")[0]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print(v, {k: round(agg[k] / cnt[k] / 1e6, 2) for k in agg}, "(millions per launch; FETCH_SIZE in GB-ish KB/1e6)")
PY
rm -rf $OUT
