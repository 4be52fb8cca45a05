#!/bin/bash
# Convergence evidence (missing #5): main.py end to end on the Syn_1827-shaped split
# (train 456 / val 456 / test 915 graphs, random.seed(0) split of data.py:206-227) with exact device ground
# truth, enough epochs for the norm-MSE of the reference's README metric to mean something.
#   tools/train_convergence.sh <neigh epochs> <gossip epochs> <out dir under gpurun_out>
NE=${1:-100}; GE=${2:-15}; OUT=${3:-gpurun_out/conv}; SEED=${4:-}
mkdir -p $OUT
( time python main.py --data_root /tmp/desco_data --output_dir /tmp/desco_results \
    --train_dataset Syn_1827_train --valid_dataset Syn_1827_val --test_dataset Syn_1827_test \
    --train_neigh --train_gossip --test_gossip --use_hetero --use_tconv --zero_node_feat \
    --neigh_epoch_num $NE --gossip_epoch_num $GE --graph_capture ${SEED:+--seed $SEED} \
    --neigh_model_path /tmp/desco_ckn --gossip_model_path /tmp/desco_ckg ) > $OUT/main_train.log 2>&1
cp /tmp/desco_results/analyze_results_*.txt $OUT/ 2>/dev/null
grep -E "epoch (0|[0-9]*9):|best|test|norm_mse|mae|real" $OUT/main_train.log | tail -60
