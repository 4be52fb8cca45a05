#!/bin/bash
# no vendor-library / torch kernels inside the timed inference pass.  rocprofv3 cannot cut a trace at
# the warm-up boundary, so the check is differential: the same bench command with 2 and with 6 timed steps -- every
# kernel that is NOT one of this library's must have the SAME call count in both traces (it runs in set-up only), and
# every kernel whose count grows with the steps must be one of ours.
#   tools/check_pass_is_native.sh [out dir under gpurun_out]            the inference pass
#   tools/check_pass_is_native.sh --train [out dir]                       the training steps: the
#       neighborhood leg (fp32, eager and hipGraph replay) and the gossip leg of bench.py --train-only, 1 against 3 timed
#       passes over the same batches (the untimed first pass -- indices, address tables, caches -- is the same in both)
MODE=pass
if [ "$1" = "--train" ]; then MODE=train; shift; fi
OUT=${1:-gpurun_out/native_$MODE}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--warmup 1 --no-cpu-baseline --no-train --no-secondary --no-x1 --no-attainable --no-profile"
for K in 2 6; do
  rm -rf $GRAFT_REPO_ROOT/$OUT/s$K
  if [ $MODE = train ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/s$K -- python3 $GRAFT_REPO_ROOT/bench.py --train-only --train-precision fp32 --train-stride 16 --train-epochs $((K / 2)) > $GRAFT_REPO_ROOT/$OUT/bench_s$K.log 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/s$K -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K $COMMON > $GRAFT_REPO_ROOT/$OUT/bench_s$K.log 2>&1
  fi
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
def counts(k):
    f = glob.glob(f"{out}/s{k}/**/*kernel_stats.csv", recursive=True)
    assert f, f"no kernel stats for --steps {k}"
    return [{r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(p))} for p in f]
# (--train-only also starts the world-of-one nccl child, which runs the same steps whatever --train-epochs says: one trace
#  per process -- a process with the SAME table in both runs is that child, the bench process is what remains)
ta, tb = counts(2), counts(6)
same = [t for t in ta if t in tb]
ta, tb = [t for t in ta if t not in same] or ta, [t for t in tb if t not in same] or tb
a, b = max(ta, key=lambda t: sum(t.values())), max(tb, key=lambda t: sum(t.values()))
ours = lambda n: "desco" in n
grow = {n: (a.get(n, 0), c) for n, c in b.items() if c != a.get(n, 0)}
foreign_in_pass = {n: v for n, v in grow.items() if not ours(n)}
print(f"kernels in the traces: {len(b)}; call count grows with --steps: {len(grow)} (per pass: "
      f"{sum((v[1] - v[0]) for v in grow.values()) // 4} launches)")
for n, v in sorted(grow.items(), key=lambda kv: -(kv[1][1] - kv[1][0])):
    print(f"  {'OURS   ' if ours(n) else 'FOREIGN'} +{(v[1] - v[0]) // 4:4d} per pass  {n[:110]}")
fixed = {n: c for n, c in b.items() if n not in grow and not ours(n)}
print(f"foreign kernels with a step-independent count (set-up only): {len(fixed)}, {sum(fixed.values())} launches")
print("RESULT:", "PASS -- the timed pass launches only this library's kernels" if not foreign_in_pass else
      "FAIL -- foreign kernels inside the pass")
sys.exit(1 if foreign_in_pass else 0)
PY
