#!/bin/bash
# Collect the per-round evidence on the GPU box (run via gpurun from the repo root):
#   tools/profile_round.sh <tag> [workloads]      e.g. r2_d "cox2:64 syn_1827:4 msrc_imdb:8"
# Per workload: bench line, rocprofv3 --kernel-trace --stats of the same command, separate --pmc
# FETCH_SIZE / WRITE_SIZE passes folded into pmc_traffic.json (keyed "<workload>_x<replicas>", what
# bench.py looks up), and three SQ counter passes.  Everything lands in gpurun_out/<tag>/.
set -u
TAG=${1:-round}
WLS=${2:-"cox2:64 syn_1827:4 msrc_imdb:8"}
[ "$WLS" = "none" ] && WLS=""      # (only the training-leg PMC passes and the x1 line)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
for WL in $WLS; do
  W=${WL%%:*}; R=${WL##*:}; K=${W}_x${R}
  ARGS="--workload $W --replicas $R"
  # the default workload's line is the driver's (CPU baseline, training legs, secondary shapes); the other shapes run
  # their own pass only -- the training legs and the CPU baseline do not depend on the inference workload
  EXTRA=""
  [ "$W" != "cox2" ] && EXTRA="--no-train --no-cpu-baseline --no-secondary --no-x1"
  python3 bench.py $ARGS --steps 10 --warmup 3 $EXTRA > $OUT/bench_$K.json 2> $OUT/bench_$K.err
  head -c 300 $OUT/bench_$K.json; echo
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$K -- python3 $ROOT/bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train > $OUT/stats_$K.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$K -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train --no-profile > $OUT/pmc_fetch_$K.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$K -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train --no-profile > $OUT/pmc_write_$K.log 2>&1
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/sq_${K}_p$i -- python3 $ROOT/bench.py $ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train --no-profile > $OUT/sq_${K}_p$i.log 2>&1
  done
  cd $ROOT
  F=$(find $OUT/pmc_fetch_$K -name "*counter_collection.csv" | head -1)
  Wf=$(find $OUT/pmc_write_$K -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$F" "$Wf" $OUT/pmc_traffic.json "python3 bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train --no-profile" $K $TAG 3 > /dev/null
  S=$(find $OUT/stats_$K -name "*kernel_stats.csv" | head -1)
  cp "$S" $OUT/kernel_stats_$K.csv 2>/dev/null
  python3 - "$OUT" "$K" <<'PY'
import csv, glob, collections, sys, json
out, key = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{out}/sq_{key}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").replace("gf16::", "").replace("small::", "").strip()
        if k.startswith("shmp_layer16_kernel<"):           # <NW, KB, ST, LD64, POOL, F16[, SELFDEG]>
            a = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            k = f"shmp_layer16_kernel<{a[1]},{a[2]}{',f16x3' if len(a) > 5 and a[5] == 'true' else ''}{',selfdeg' if len(a) > 6 and a[6] == 'true' else ''}>"
        elif k.startswith("shmp_layer_f32_kernel<"):
            a = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            k = f"shmp_layer_f32_kernel<{a[0]},{a[1]},{'x6' if a[2] == 'true' else 'f32'}>"
        else:
            k = k.split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
res = {k: dict(d) for k, d in agg.items()}
for k in res:
    res[k]["launches"] = max(cnt[k].values())
json.dump(res, open(f"{out}/sq_counters_{key}.json", "w"), indent=1)
PY
  # keep the merge-back small: drop the raw per-dispatch traces
  find $OUT/stats_$K $OUT/pmc_fetch_$K $OUT/pmc_write_$K $OUT/sq_${K}_p1 $OUT/sq_${K}_p2 $OUT/sq_${K}_p3 -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
  head -6 $OUT/kernel_stats_$K.csv
done
# training legs: FETCH_SIZE / WRITE_SIZE per kernel, one leg per run (their dominant kernels share names), folded into
# pmc_traffic.json under train_fp32 / train_bf16 / train_gossip (bench.py's train_traffic)
if [ "${PROFILE_TRAIN:-1}" = "1" ]; then
  for LEG in fp32 bf16 gossip; do
    TA="--train-only --train-leg $LEG --train-stride 16"
    cd /tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_train_$LEG -- python3 $ROOT/bench.py $TA > $OUT/pmc_fetch_train_$LEG.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train_$LEG -- python3 $ROOT/bench.py $TA > $OUT/pmc_write_train_$LEG.log 2>&1
    cd $ROOT
    F=$(find $OUT/pmc_fetch_train_$LEG -name "*counter_collection.csv" | head -1)
    Wf=$(find $OUT/pmc_write_train_$LEG -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summary.py "$F" "$Wf" $OUT/pmc_traffic.json "python3 bench.py $TA" train_$LEG $TAG 1 > /dev/null
    find $OUT/pmc_fetch_train_$LEG $OUT/pmc_write_train_$LEG -name "*.csv" -delete 2>/dev/null
  done
fi
python3 bench.py --steps 50 --warmup 5 --replicas 1 --no-cpu-baseline --no-train > $OUT/bench_cox2_x1.json 2>> $OUT/bench_cox2_x64.err
