#!/bin/bash
# Collect the per-round evidence on the GPU box (run via gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r1_e  -> gpurun_out/<tag>/...
# 1. full GPU test log, 2. bench lines (x64 default, x1), 3. rocprofv3 --kernel-trace --stats of the
# bench command, 4. separate --pmc FETCH_SIZE / WRITE_SIZE passes folded into pmc_traffic.json.
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --steps 50 --warmup 5 --replicas 1 --no-cpu-baseline > $OUT/bench_x1.json 2>> $OUT/bench.err
python3 bench.py --steps 20 --warmup 3 --graph --no-cpu-baseline > $OUT/bench_graph.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/pmc_write.log 2>&1
cd $ROOT
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py "$F" "$W" $OUT/pmc_traffic.json "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile" > /dev/null
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp "$S" $OUT/kernel_stats.csv 2>/dev/null
# keep the merge-back small: drop the raw per-dispatch traces
find $OUT/stats $OUT/pmc_fetch $OUT/pmc_write -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete 2>/dev/null
head -c 600 $OUT/bench.json; echo
head -8 $OUT/kernel_stats.csv
