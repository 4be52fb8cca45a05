#!/bin/bash
# Where do the SHMP layer kernel's HBM bytes go on Syn_1827-shaped input?  
#   tools/syn_traffic.sh <tag> [replicas]
# For DESCO_DEGREE_SORT = 1 (default order) and 0 (partition order): FETCH_SIZE, WRITE_SIZE, L2 hit / miss / EA read
# requests and L1->L2 requests of the layer kernels, per launch, next to the bench line's algorithmic bytes.
set -u
TAG=${1:-syn_traffic}; R=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
ARGS="--workload syn_1827 --replicas $R --steps 2 --warmup 1 --no-cpu-baseline --no-x1 --no-secondary --no-attainable --no-train"
for DS in 1 0; do
  export DESCO_DEGREE_SORT=$DS
  cd $ROOT
  python3 bench.py $ARGS --steps 5 > $OUT/bench_ds$DS.json 2> $OUT/bench_ds$DS.err
  cd /tmp && export TMPDIR=/tmp
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/ds${DS}_p$i -- python3 $ROOT/bench.py $ARGS --no-profile > $OUT/ds${DS}_p$i.log 2>&1
  done
done
cd $ROOT
python3 - "$OUT" <<'PY' | tee $OUT/syn_traffic.txt
import csv, glob, collections, json, sys
out = sys.argv[1]
for ds in (1, 0):
    try:
        b = json.loads(open(f"{out}/bench_ds{ds}.json").read().strip().splitlines()[-1])
        g = b["roofline"].get("gather") or {}
        print(f"DESCO_DEGREE_SORT={ds}: {b['value']:.0f} graphs/s  {b['ms_per_step']:.2f} ms/pass; layer kernel "
              f"{g.get('kernel')}: {g.get('avg_launch_ms', 0):.3f} ms/launch, algorithmic "
              f"{g.get('algorithmic_bytes_per_launch', 0) / 1e9:.3f} GB/launch, frac of HBM peak {g.get('frac', 0):.3f}")
    except Exception as e:
        print("bench line unreadable:", e)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"{out}/ds{ds}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").replace("gf16::", "").replace("small::", "").strip()
            if not k.startswith("shmp_layer16_kernel<"):
                continue
            a = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            k = f"shmp_layer16_kernel<{a[1]},{a[2]}{',f16x3' if len(a) > 5 and a[5] == 'true' else ''}{',selfdeg' if len(a) > 6 and a[6] == 'true' else ''}>"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    for k, c in agg.items():
        n = max(cnt[k].values())
        print(f"  {k}: {n} dispatches")
        for name in sorted(c):
            print(f"      {name:34s} {c[name] / cnt[k][name]:18.1f} per launch")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f_, w_ = c["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"], c["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"]
            print(f"      HBM bytes per launch (2 x FETCH + WRITE, KB -> B): {(2 * f_ + w_) * 1024 / 1e9:.3f} GB "
                  f"(read {2 * f_ * 1024 / 1e9:.3f}, write {w_ * 1024 / 1e9:.3f})")
PY
find $OUT -name "*.csv" -delete 2>/dev/null
