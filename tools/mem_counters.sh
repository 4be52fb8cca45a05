#!/bin/bash
# Memory-pipe counters (TA / TCP / TCC / UTCL1) per kernel of one bench pass (run via gpurun from the repo root):
#   tools/mem_counters.sh <tag> [workload] [replicas]
set -u
TAG=${1:-mem}; W=${2:-syn_1827}; R=${3:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "TA_BUSY_avr GRBM_GUI_ACTIVE" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  # (a counter set the hardware cannot collect aborts and then hangs in finalize: bound every pass)
  timeout 150 rocprofv3 --pmc $P --output-format csv -d $OUT/mem_${W}_p$i -- python3 $ROOT/bench.py --workload $W --replicas $R --steps 1 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-profile > $OUT/mem_${W}_p$i.log 2>&1
done
cd $ROOT
python3 - "$OUT" "$W" <<'PY' | tee $OUT/mem_counters_$W.txt
import csv, glob, collections, sys
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{out}/mem_{w}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").replace("gf16::", "").replace("small::", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("TCP_TOTAL_READ_sum", 0) * 0 - sum(kv[1].get("TD_TD_BUSY_sum", 0) for _ in (0,))):
    if not any(t in k for t in ("shmp", "gossip_fused", "gemm_split", "linear64", "count_head")):
        continue
    n = max(cnt[k].values())
    print(k[:90], "dispatches", n)
    for name in sorted(c):
        print(f"    {name:45s} {c[name] / cnt[k][name]:16.1f} per dispatch")
PY
find $OUT -name "*.csv" -path "*mem_${W}_p*" -delete 2>/dev/null
