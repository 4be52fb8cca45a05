#!/bin/bash
# Train briefly, then evaluate the SAME checkpoints with the round-3 work-order features on and off: the held-out metrics
# must agree (the features only reorder work).   tools/debug/eval_ab.sh <neigh epochs> <gossip epochs>
NE=${1:-40}; GE=${2:-8}
COMMON="--data_root /tmp/desco_data --train_dataset Syn_1827_train --valid_dataset Syn_1827_val --test_dataset Syn_1827_test --use_hetero --use_tconv --zero_node_feat"
python main.py $COMMON --output_dir /tmp/desco_res_train --train_neigh --train_gossip --test_gossip --neigh_epoch_num $NE --gossip_epoch_num $GE \
   --graph_capture --neigh_model_path /tmp/desco_ckn --gossip_model_path /tmp/desco_ckg > /tmp/train.log 2>&1
grep -E "best|norm_mse|mae" /tmp/train.log
NC=$(grep "best neighborhood model path" /tmp/train.log | awk '{print $NF}'); GC=$(grep "best gossip model path" /tmp/train.log | awk '{print $NF}')
for mode in "DESCO_DEGREE_SORT=1 DESCO_GOSSIP_TILE_ORDER=1" "DESCO_DEGREE_SORT=0 DESCO_GOSSIP_TILE_ORDER=0"; do
  echo "== test only, $mode"
  env $mode python main.py $COMMON --output_dir /tmp/desco_res_eval --neigh_checkpoint $NC --gossip_checkpoint $GC --test_gossip > /tmp/eval.log 2>&1
  grep -E "test|norm_mse|mae" /tmp/eval.log
done
