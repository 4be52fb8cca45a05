"""One rank of the 2-rank world of tests/test_multirank_gpu.py.  Started by
desco_amd.distributed.launch with the torchrun environment; DESCO_SHARE_GPU=1 puts every rank on
cuda:0 with the gloo backend (the GPU box has one GPU; RCCL needs one device per rank)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from desco_amd import distributed as D  # noqa: E402


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    dev = D.local_device()
    D.init_from_env(dev)
    rank, world = D.rank(), D.world_size()
    assert world == 2
    import multirank_common as C
    from desco_amd.graphs import GraphSet
    res = {}
    if mode in ("pipeline", "pipeline_chunks"):
        from desco_amd.pipeline import InferencePipeline
        nm, gm, qids, queries = C.models(dev)
        gs = GraphSet.from_edge_lists(C.mixed_graphs())
        pipe = InferencePipeline(nm, gm, gs, depth=4, device=dev,      # rank / world from the group
                                 chunks=C.CHUNKS if mode == "pipeline_chunks" else None)
        assert (pipe.rank, pipe.world) == (rank, world)
        out = pipe.run()
        full = pipe.gather(out, node_level=True)
        res = {"range": pipe.graph_range}
        if rank == 0:
            res.update({k: v.cpu() for k, v in full.items()})
        else:
            assert full is None
    elif mode == "tiny":
        # fewer graphs than ranks: one rank's shard is empty and must still take part in the gather
        from desco_amd.pipeline import InferencePipeline
        nm, gm, qids, queries = C.models(dev)
        gs = GraphSet.from_edge_lists(C.mixed_graphs()[:1])
        pipe = InferencePipeline(nm, gm, gs, depth=4, device=dev)
        full = pipe.gather(pipe.run(), node_level=True)
        res = {"range": pipe.graph_range}
        if rank == 0:
            res.update({k: v.cpu() for k, v in full.items()})
    elif mode == "grads":
        from desco_amd.batch import GossipBatch, NeighborhoodBatch
        from desco_amd.partition import build_partition
        nm, gm, qids, queries = C.models(dev)
        graphs = C.mixed_graphs()[:C.TRAIN_GRAPHS]
        gs = GraphSet.from_edge_lists(graphs)
        part = build_partition(gs, 4)
        y = C.neigh_labels(part.num_neigh, len(queries))
        cut = int(part.num_neigh * 0.6)
        bounds = [(0, cut), (cut, part.num_neigh)]
        sizes = [b - a for a, b in bounds]
        a, b = bounds[rank]
        bk = D.GradBuckets(list(nm.parameters()), 4)
        bk.zero()
        loss = nm.train_forward(NeighborhoodBatch(part.slice(a, b), dev, y=y[a:b]), 0)
        (loss * D.mean_loss_weight(sizes, [0, 1], rank)).backward()
        bk.finish()
        res["neigh"] = {n: p.grad.detach().cpu().clone() for n, p in nm.named_parameters()}
        bk.close()
        # gossip: sum loss, sum reduce; rank r takes graphs [8r, 8r+8)
        x, yg = C.gossip_inputs(gs.num_nodes, len(queries))
        gm.set_query_emb(nm.get_query_emb())
        g0, g1 = 8 * rank, 8 * rank + 8
        n0, n1 = int(gs.graph_ptr[g0]), int(gs.graph_ptr[g1])
        bk = D.GradBuckets(list(gm.parameters()), 2)
        bk.zero()
        gm.train_forward(GossipBatch(gs.subset(g0, g1), dev, x=x[n0:n1], y=yg[n0:n1]), 0).backward()
        bk.finish()
        res["gossip"] = {n: p.grad.detach().cpu().clone() for n, p in gm.named_parameters()}
        bk.close()
    elif mode == "fit":
        from desco_amd.lightning_data import LightningDataLoader
        from desco_amd.trainer import ModelCheckpoint, Trainer
        from desco_amd.workload import Workload
        nm, gm, qids, queries = C.models(dev)
        if rank == 1:      # fit() must re-synchronise the replicas from rank 0
            with torch.no_grad():
                for p in nm.parameters():
                    p.add_(0.5)
        gs = GraphSet.from_edge_lists(C.mixed_graphs()[:C.TRAIN_GRAPHS])
        w = Workload(gs, root=None)
        w.generate_pipeline_datasets(depth_neigh=4)
        nd = w.neighborhood_dataset
        nd.y = C.neigh_labels(len(nd), len(queries))
        loader = LightningDataLoader(train_dataset=nd, val_dataset=nd, test_dataset=nd,
                                     batch_size=C.NEIGH_BATCH)
        ck = ModelCheckpoint(monitor="neighborhood_counting_val_loss")
        tr = Trainer(max_epochs=2, devices=[0, 1], strategy="ddp", default_root_dir=out_path + ".ckpt",
                     callbacks=[ck])
        tr.fit(nm, loader)
        res["params"] = {k: v.detach().cpu().clone() for k, v in nm.state_dict().items()}
        res["history"] = tr.history
        res["best"] = ck.best_model_path
        res["pred"] = torch.cat(tr.predict(nm, loader.test_dataloader())).cpu()
        # gossip stage: sum loss
        x, yg = C.gossip_inputs(gs.num_nodes, len(queries))
        gd = w.gossip_dataset
        gd.x, gd.y = x, yg
        gm.set_query_emb(nm.get_query_emb())
        gl = LightningDataLoader(train_dataset=gd, val_dataset=gd, test_dataset=gd, batch_size=C.GOSSIP_BATCH)
        tg = Trainer(max_epochs=1, devices=[0, 1], strategy="ddp", default_root_dir=out_path + ".ckptg",
                     callbacks=[ModelCheckpoint(monitor="gossip_counting_val_loss")], grad_reduce="sum")
        tg.fit(gm, gl)
        res["gparams"] = {k: v.detach().cpu().clone() for k, v in gm.state_dict().items()}
    elif mode == "fit_replay":
        # Trainer(strategy="ddp") with and without graph_capture: the replayed data-parallel step (two hipGraphs around
        # the bucket all-reduces, trainer.DDPReplay) must leave the parameters of the eager one, bit for bit
        from desco_amd import ops
        from desco_amd.lightning_data import LightningDataLoader
        from desco_amd.trainer import Trainer
        from desco_amd.workload import Workload
        gs = GraphSet.from_edge_lists(C.mixed_graphs()[:C.TRAIN_GRAPHS])
        w = Workload(gs, root=None)
        w.generate_pipeline_datasets(depth_neigh=4)
        nd = w.neighborhood_dataset
        x, yg = C.gossip_inputs(gs.num_nodes, 29)
        for capture in (False, True):
            nm, gm, qids, queries = C.models(dev, gossip_dropout=0.01)
            nd.y = C.neigh_labels(len(nd), len(queries))
            loader = LightningDataLoader(train_dataset=nd, val_dataset=nd, test_dataset=nd, batch_size=C.NEIGH_BATCH)
            tr = Trainer(max_epochs=3, devices=[0, 1], strategy="ddp", default_root_dir=out_path + f".ck{int(capture)}",
                         graph_capture=capture)
            tr.fit(nm, loader)
            key = f"{'replay' if capture else 'eager'}"
            res[key + "_params"] = {k: v.detach().cpu().clone() for k, v in nm.state_dict().items()}
            res[key + "_history"] = tr.history
            gd = w.gossip_dataset
            gd.x, gd.y = x, yg
            gm.set_query_emb(nm.get_query_emb().detach())
            ops.manual_seed(77)
            gl = LightningDataLoader(train_dataset=gd, val_dataset=gd, test_dataset=gd, batch_size=C.GOSSIP_BATCH)
            tg = Trainer(max_epochs=3, devices=[0, 1], strategy="ddp", default_root_dir=out_path + f".cg{int(capture)}",
                         grad_reduce="sum", graph_capture=capture)
            tg.fit(gm, gl)
            res[key + "_gparams"] = {k: v.detach().cpu().clone() for k, v in gm.state_dict().items()}
            res[key + "_rng"] = ops.rng_state(dev).cpu().tolist()
    else:
        raise SystemExit(f"unknown mode {mode}")
    torch.save(res, f"{out_path}.rank{rank}")
    D.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
