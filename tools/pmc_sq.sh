#!/bin/bash
# Diagnostic SQ counter passes over one bench step (run on the GPU box via gpurun).
# usage: tools/pmc_sq.sh <out-subdir under gpurun_out> ["script.py args" relative to the repo root]
set -u
OUT=${1:-sq}
CMD=${2:-"bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
mkdir -p $ROOT/gpurun_out/$OUT
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $ROOT/gpurun_out/$OUT/p$i -- python3 $ROOT/$CMD > $ROOT/gpurun_out/$OUT/p$i.log 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"gpurun_out/{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("desco::", "").split("<")[0].strip()
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
res = {k: {c: v for c, v in d.items()} for k, d in agg.items()}
for k in res:
    res[k]["launches"] = max(cnt[k].values())
json.dump(res, open(f"gpurun_out/{out}/sq_summary.json", "w"), indent=1)
for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0))[:6]:
    d = res[k]; wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k, "launches", d["launches"])
    for c in sorted(d):
        if c != "launches":
            print(f"   {c:32s} {d[c]:.4g}  ({d[c]/wc:.3f} of WAVE_CYCLES)")
PY
