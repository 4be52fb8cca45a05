#!/usr/bin/env python3
"""End-to-end DeSCo driver on the MI355X-native path, with the reference's CLI
(``python main.py --test_dataset COX2 --neigh_checkpoint ... --gossip_checkpoint ... --test_gossip``,
``--train_neigh --train_gossip``; flags of subgraph_counting/config.py).  Mirrors the stages of the
reference's main.py (31-517) on ``desco_amd``: ground truth -> pipeline datasets -> neighborhood
model (train / load, test, predict) -> apply_neighborhood_count -> gossip model (train / load, test,
predict) -> CSV dumps + norm-MSE / MAE report.
"""
from __future__ import annotations

import argparse
import datetime
import os
import pickle

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F

from desco_amd.analysis import mae, norm_mse
from desco_amd.config import parse_gossip, parse_neighborhood, parse_optimizer, split_namespaces
from desco_amd.data import gen_query_ids, graph_atlas_plus, load_data
from desco_amd.lightning_data import LightningDataLoader
from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
from desco_amd.trainer import ModelCheckpoint, Trainer
from desco_amd.transforms import ToTconvHetero
from desco_amd.workload import Workload


def build_workload(name, query_ids, nx_queries, depth, transform, num_cpu, root="data", node_feat_len=-1):
    w = Workload(load_data(name, root_folder=root), os.path.join(root, name), hetero_graph=True,
                 node_feat_len=node_feat_len)
    if w.exist_groundtruth(query_ids=query_ids, queries=nx_queries):
        w.canonical_count_truth = w.load_groundtruth(query_ids=query_ids, queries=nx_queries)
    else:
        w.canonical_count_truth = w.compute_groundtruth(query_ids=query_ids, queries=nx_queries,
                                                        num_workers=num_cpu, save_to_file=True)
    w.generate_pipeline_datasets(depth_neigh=depth, neighborhood_transform=transform)
    return w


def main(args_neighborhood, args_gossip, args_opt, train_neighborhood=True, train_gossip=True,
         test_gossip=True, neighborhood_checkpoint=None, gossip_checkpoint=None, nx_queries=None,
         atlas_query_ids=None, output_dir="results/raw", data_root="data"):
    if nx_queries is None and atlas_query_ids is None:
        raise ValueError("nx_queries and atlas_query_ids cannot be both None")
    query_ids = atlas_query_ids
    if nx_queries is None:
        nx_queries = [graph_atlas_plus(i) for i in atlas_query_ids]
        if getattr(args_neighborhood, "use_node_feature", False):           # main.py:51-64
            from desco_amd.data import add_node_feat_to_networkx
            eye = [t for t in np.eye(args_neighborhood.input_dim).tolist()]
            nx_queries = [g for q in nx_queries for g in add_node_feat_to_networkx(q, eye, "feat")]
            query_ids = None
            print("query_ids set to None because node features are used")
    else:
        query_ids = None
    node_feat_len = args_neighborhood.input_dim if getattr(args_neighborhood, "use_node_feature", False) else -1
    transform = ToTconvHetero() if args_neighborhood.use_tconv else None
    assert args_neighborhood.use_hetero if args_neighborhood.use_tconv else True
    depth, ncpu = args_neighborhood.depth, args_opt.num_cpu

    devices = args_opt.gpu if isinstance(args_opt.gpu, list) else [args_opt.gpu]
    # len(devices) > 1: one process per GPU (started by __main__ below or by torch.distributed.run);
    # every rank runs this function, training and prediction are data parallel over RCCL
    # (Trainer(strategy="ddp")), and rank 0 alone writes the output files.  The reference
    # parallelises only neighborhood training (main.py:242-255) and refuses it for the gossip model
    # (main.py:353-356); both stages and both predict passes are sharded here.
    from desco_amd import distributed as D
    _, env_world, _ = D.env_world()
    if len(devices) > 1 and env_world != len(devices):
        raise RuntimeError(f"--gpu {devices}: {len(devices)} ranks expected, WORLD_SIZE={env_world}; run "
                           "main.py directly (it starts the ranks) or under torch.distributed.run")
    strategy = "ddp" if len(devices) > 1 else None
    if strategy:
        D.init_from_env(D.local_device(devices))

    def build_all():
        tw = vw = None
        if train_neighborhood or train_gossip:
            tw = build_workload(args_opt.train_dataset, query_ids, nx_queries, depth, transform, ncpu, data_root,
                                node_feat_len)
            vw = build_workload(args_opt.valid_dataset, query_ids, nx_queries, depth, transform, ncpu, data_root,
                                node_feat_len)
        return tw, vw, build_workload(args_opt.test_dataset, query_ids, nx_queries, depth, transform, ncpu,
                                      data_root, node_feat_len)

    # rank 0 computes the ground truth / partitions and writes the on-disk caches; the others read
    # them.  They wait on the process group's store, not in a collective: a cold-cache build may
    # take longer than any collective watchdog allows
    train_w, valid_w, test_w = D.rank0_first(build_all)

    # ---------------- neighborhood counting ----------------
    neigh_loader = LightningDataLoader(
        train_dataset=train_w.neighborhood_dataset if train_w else None,
        val_dataset=valid_w.neighborhood_dataset if valid_w else None,
        test_dataset=test_w.neighborhood_dataset, batch_size=args_neighborhood.batch_size,
        num_workers=ncpu, shuffle=False)
    neigh_ckpt = ModelCheckpoint(monitor="neighborhood_counting_val_loss", mode="min", save_top_k=1,
                                 save_last=True)
    neigh_trainer = Trainer(max_epochs=args_neighborhood.epoch_num, accelerator="gpu", devices=devices,
                            default_root_dir=args_neighborhood.model_path, callbacks=[neigh_ckpt],
                            strategy=strategy, grad_reduce="mean", verbose=True,
                            precision=getattr(args_opt, "precision", "fp32"),
                            graph_capture=getattr(args_opt, "graph_capture", False))
    if train_neighborhood and neighborhood_checkpoint is None:
        neigh_model = NeighborhoodCountingModel(input_dim=args_neighborhood.input_dim,
                                                hidden_dim=args_neighborhood.hidden_dim,
                                                args=args_neighborhood)
        neigh_model = neigh_model.to_hetero_old(tconv_target=args_neighborhood.use_tconv,
                                                tconv_query=args_neighborhood.use_tconv)
    else:
        assert neighborhood_checkpoint is not None
        print("loading neighborhood model from checkpoint: ", neighborhood_checkpoint)
        neigh_model = NeighborhoodCountingModel.load_from_checkpoint(neighborhood_checkpoint)
    neigh_model.to(neigh_trainer.device)
    neigh_model.set_queries(query_ids=query_ids, queries=nx_queries, transform=transform)
    if train_neighborhood:
        neigh_trainer.fit(model=neigh_model, datamodule=neigh_loader)
        print("best neighborhood model path: ", neigh_ckpt.best_model_path)
        neigh_model = NeighborhoodCountingModel.load_from_checkpoint(neigh_ckpt.best_model_path)
        neigh_model.to(neigh_trainer.device)
        neigh_model.set_queries(query_ids=query_ids, queries=nx_queries, transform=transform)
    print("neighborhood test:", neigh_trainer.test(model=neigh_model, datamodule=neigh_loader))

    # ---------------- gossip counting ----------------
    skip_gossip = not (train_gossip or test_gossip)
    if train_gossip:
        for w, loader in ((train_w, neigh_loader.train_dataloader()), (valid_w, neigh_loader.val_dataloader())):
            w.apply_neighborhood_count(torch.cat(neigh_trainer.predict(neigh_model, loader), dim=0))
    neighborhood_count_test = torch.cat(neigh_trainer.predict(neigh_model, neigh_loader.test_dataloader()), dim=0)
    if test_gossip or train_gossip:
        test_w.apply_neighborhood_count(neighborhood_count_test)

    gossip_model = gossip_trainer = gossip_loader = None
    if not skip_gossip:
        gossip_loader = LightningDataLoader(
            train_dataset=train_w.gossip_dataset if train_gossip else None,
            val_dataset=valid_w.gossip_dataset if train_gossip else None,
            test_dataset=test_w.gossip_dataset, batch_size=args_gossip.batch_size, num_workers=ncpu,
            shuffle=False)
        args_gossip.use_hetero = False
        if train_gossip and gossip_checkpoint is None:
            gossip_model = GossipCountingModel(1, args_gossip.hidden_dim, args_gossip,
                                               emb_channels=args_neighborhood.hidden_dim,
                                               input_pattern_emb=True)
        else:
            assert gossip_checkpoint is not None
            print("loading gossip model from checkpoint: ", gossip_checkpoint)
            gossip_model = GossipCountingModel.load_from_checkpoint(gossip_checkpoint)
        gossip_ckpt = ModelCheckpoint(monitor="gossip_counting_val_loss", mode="min", save_top_k=1,
                                      save_last=True)
        # the gossip loss is a SUM over nodes and queries (lightning_model.py:607): sum-reduce
        gossip_trainer = Trainer(max_epochs=args_gossip.epoch_num, accelerator="gpu", devices=devices,
                                 default_root_dir=args_gossip.model_path, callbacks=[gossip_ckpt],
                                 strategy=strategy, grad_reduce="sum", verbose=True,
                                 precision=getattr(args_opt, "precision", "fp32"))
        gossip_model.to(gossip_trainer.device)
        gossip_model.set_query_emb(neigh_model.get_query_emb())
        if train_gossip:
            gossip_trainer.fit(model=gossip_model, datamodule=gossip_loader)
            print("best gossip model path: ", gossip_ckpt.best_model_path)
            gossip_model = GossipCountingModel.load_from_checkpoint(gossip_ckpt.best_model_path)
            gossip_model.to(gossip_trainer.device)
            gossip_model.set_query_emb(neigh_model.get_query_emb())
        elif test_gossip:
            print("gossip test:", gossip_trainer.test(gossip_model, datamodule=gossip_loader))

    # ---------------- outputs (main.py:381-515) ----------------
    gossip_count_test = None
    if not skip_gossip:
        gossip_count_test = torch.cat(gossip_trainer.predict(gossip_model, gossip_loader.test_dataloader()), dim=0)
    if D.rank() != 0:          # every rank holds the full predictions; rank 0 writes the files
        D.barrier()
        return None
    os.makedirs(output_dir, exist_ok=True)
    ds = args_opt.test_dataset
    with open(os.path.join(output_dir, f"config_{ds}.txt"), "w") as f:
        f.write(f"args_opt: \n{args_opt}\nargs_neighborhood:\n{args_neighborhood}\nargs_gossip:\n{args_gossip}"
                f"\ntime:\n{datetime.datetime.now()}")
    graphlet_neigh = test_w.neighborhood_dataset.aggregate_neighborhood_count(neighborhood_count_test)
    pd.DataFrame(torch.round(F.relu(graphlet_neigh)).cpu().numpy()).to_csv(
        os.path.join(output_dir, f"neighborhood_graphlet_{ds}.csv"))
    graphlet_gossip = None
    if not skip_gossip:
        graphlet_gossip = test_w.gossip_dataset.aggregate_neighborhood_count(gossip_count_test)
        pd.DataFrame(torch.round(F.relu(graphlet_gossip)).cpu().numpy()).to_csv(
            os.path.join(output_dir, f"gossip_graphlet_{ds}.csv"))
        gates = gossip_model._gate_value(gossip_model.query_emb).squeeze(dim=-1)
        pd.DataFrame(gates.cpu().numpy()).to_csv(os.path.join(output_dir, f"gossip_gate_{ds}.csv"))
        pd.DataFrame(gossip_count_test.cpu().numpy()).to_csv(
            os.path.join(output_dir, f"gossip_node_{ds}_results.csv"))
    pd.DataFrame(neighborhood_count_test.cpu().numpy()).to_csv(
        os.path.join(output_dir, f"neighborhood_node_{ds}_results.csv"))
    pd.DataFrame(test_w.neighborhood_dataset.nx_neighs_index).to_csv(
        os.path.join(output_dir, f"neighborhood_node_{ds}_index.csv"))
    with open(os.path.join(output_dir, f"test_nxgraph_{ds}.pk"), "wb") as f:
        pickle.dump(test_w.to_networkx(), f)

    sizes = sorted({len(q) for q in nx_queries})
    groupby = [[i for i, q in enumerate(nx_queries) if len(q) == s] for s in sizes]
    truth_graphlet = test_w.gossip_dataset.aggregate_neighborhood_count(test_w.canonical_count_truth).numpy()
    report = {}
    pred_n = torch.round(F.relu(graphlet_neigh)).cpu().numpy()
    report["graphlet_norm_mse_neighborhood"] = norm_mse(pred_n, truth_graphlet, groupby)
    report["graphlet_mae_neighborhood"] = mae(pred_n, truth_graphlet, groupby)
    if not skip_gossip:
        pred_g = torch.round(F.relu(graphlet_gossip)).cpu().numpy()
        report["graphlet_norm_mse_gossip"] = norm_mse(pred_g, truth_graphlet, groupby)
        report["graphlet_mae_gossip"] = mae(pred_g, truth_graphlet, groupby)
        pd.DataFrame(pred_g).to_csv(os.path.join(output_dir, f"graphlet_count_{ds}.csv"))
        pd.DataFrame(truth_graphlet).to_csv(os.path.join(output_dir, f"graphlet_truth_{ds}.csv"))
    with open(os.path.join(output_dir, f"analyze_results_{ds}.txt"), "w") as f:
        for k, v in report.items():
            print(f"{k}: {v}")
            f.write(f"{k}: {v}\n")
    print("done")
    D.barrier()
    return report


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="DeSCo argument parser (MI355X-native path)")
    parse_optimizer(parser)
    parse_neighborhood(parser)
    parse_gossip(parser)
    parser.add_argument("--data_root", type=str, default="data")
    parser.add_argument("--precision", type=str, default="fp32", choices=["fp32", "bf16"],
                        help="matrix-product precision of the training steps (bf16: BASELINE config 3)")
    parser.add_argument("--graph_capture", action="store_true",
                        help="replay each neighborhood training batch's step from a hipGraph after epoch 0")
    parser.add_argument("--seed", type=int, default=None,
                        help="seed of the model initialisation and the batch order (the reference seeds neither: runs differ)")
    args = parser.parse_args()
    gpus = args.gpu if isinstance(args.gpu, list) else [args.gpu]
    if len(gpus) > 1 and "WORLD_SIZE" not in os.environ:
        # --gpu 0 1 ..: start one process per GPU (what Lightning's "ddp" strategy does for
        # main.py:242-255), before this process touches the GPU; LOCAL_RANK r uses gpus[r]
        import sys
        from desco_amd import distributed as D
        sys.exit(D.launch([os.path.abspath(__file__)] + sys.argv[1:], len(gpus), devices=gpus))
    print(args)
    if args.seed is not None:
        import random as _random
        import numpy as _np
        import torch as _torch
        _random.seed(args.seed)
        _np.random.seed(args.seed)
        _torch.manual_seed(args.seed)
    args_neighborhood, args_gossip, args_opt = split_namespaces(args)
    args_opt.precision = args.precision          # this build's flags (not in the reference's groups)
    args_opt.graph_capture = args.graph_capture
    assert args_neighborhood.use_hetero
    query_ids = gen_query_ids(query_size=[3, 4, 5])
    output_dir = args_opt.output_dir or os.path.join(
        "results/wdsm24/raw", datetime.datetime.now().strftime("%Y%m%d_%H:%M:%S"))
    main(args_neighborhood, args_gossip, args_opt, train_neighborhood=args_opt.train_neigh,
         train_gossip=args_opt.train_gossip, test_gossip=args_opt.test_gossip,
         neighborhood_checkpoint=args_opt.neigh_checkpoint, gossip_checkpoint=args_opt.gossip_checkpoint,
         atlas_query_ids=query_ids, output_dir=output_dir, data_root=args.data_root)
