#!/bin/bash
# LDS bank-conflict rate per kernel of one bench pass (run via gpurun from the repo root):
#   tools/lds_conflicts.sh <tag> [workload] [replicas]
set -u
TAG=${1:-lds}; W=${2:-cox2}; R=${3:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/lds_$W -- python3 $ROOT/bench.py --workload $W --replicas $R --steps 1 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-profile > $OUT/lds_$W.log 2>&1
cd $ROOT
python3 - "$OUT/lds_$W" <<'PY' | tee $OUT/lds_conflicts_$W.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print(f"{k[:70]:70s} conflict/active {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}  "
              f"lds active / busy cycles {c['SQ_LDS_IDX_ACTIVE'] / max(c['SQ_BUSY_CYCLES'], 1):.3f}")
PY
find $OUT/lds_$W -name "*.csv" -delete 2>/dev/null
