#!/bin/bash
# two seeded training runs must give identical validation-loss histories.
#   tools/check_train_reproducible.sh <neigh epochs> <gossip epochs> <out dir under gpurun_out>
NE=${1:-6}; GE=${2:-3}; OUT=${3:-gpurun_out/repro}
mkdir -p $OUT
for R in a b; do
  rm -rf /tmp/desco_results_$R /tmp/desco_ckn_$R /tmp/desco_ckg_$R
  python main.py --data_root /tmp/desco_data --output_dir /tmp/desco_results_$R \
      --train_dataset Syn_1827_train --valid_dataset Syn_1827_val --test_dataset Syn_1827_test \
      --train_neigh --train_gossip --test_gossip --use_hetero --use_tconv --zero_node_feat \
      --neigh_epoch_num $NE --gossip_epoch_num $GE --graph_capture --seed 0 \
      --neigh_model_path /tmp/desco_ckn_$R --gossip_model_path /tmp/desco_ckg_$R > $OUT/run_$R.log 2>&1
  grep -E "val_loss|val loss|epoch [0-9]+:" $OUT/run_$R.log | sed -E 's/[0-9.]+ ?s\b//g; s/\([0-9.]+ s[^)]*\)//g' > $OUT/hist_$R.txt
done
wc -l $OUT/hist_a.txt $OUT/hist_b.txt | head -3
if cmp -s $OUT/hist_a.txt $OUT/hist_b.txt; then echo "RESULT: PASS -- identical loss histories ($(wc -l < $OUT/hist_a.txt) lines)"; head -3 $OUT/hist_a.txt; tail -2 $OUT/hist_a.txt
else echo "RESULT: FAIL -- histories differ"; diff $OUT/hist_a.txt $OUT/hist_b.txt | head -10; fi
