"""The main.py-style driver end to end on tiny data: train both stages for an epoch, reload the
best checkpoints, predict, dump the reference's artefacts."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_tu(root, name, graphs):
    raw = os.path.join(root, name, "raw")
    os.makedirs(raw, exist_ok=True)
    a, gi, off = [], [], 0
    for g, (n, edges) in enumerate(graphs):
        gi += [g + 1] * n
        for u, v in edges:
            a += [(off + u + 1, off + v + 1), (off + v + 1, off + u + 1)]
        off += n
    np.savetxt(os.path.join(raw, name + "_A.txt"), np.array(a), fmt="%d", delimiter=", ")
    np.savetxt(os.path.join(raw, name + "_graph_indicator.txt"), np.array(gi), fmt="%d")


def test_driver_trains_and_writes_reference_artifacts(tmp_path):
    import main as driver
    from desco_amd import config
    from desco_amd.data import load_data
    from helpers import golden_graphs
    graphs = golden_graphs(max_n=30)
    root = str(tmp_path / "data")
    _write_tu(root, "TOY", graphs)
    gs = load_data("TOY", root_folder=root)
    assert gs.edge_lists() == [(n, sorted(tuple(sorted(e)) for e in es)) for n, es in graphs]
    assert load_data("TOY_train", root_folder=root).num_graphs == int(len(graphs) * 0.25)
    p = argparse.ArgumentParser()
    config.parse_optimizer(p)
    config.parse_neighborhood(p)
    config.parse_gossip(p)
    args = p.parse_args(["--train_dataset", "TOY_train", "--valid_dataset", "TOY_val", "--test_dataset",
                         "TOY_test", "--neigh_epoch_num", "2", "--gossip_epoch_num", "2",
                         "--neigh_batch_size", "64", "--gossip_batch_size", "4",      # (--gossip_dropout: default 0.01)
                         "--neigh_model_path", str(tmp_path / "ckpt_n"), "--gossip_model_path",
                         str(tmp_path / "ckpt_g"), "--train_neigh", "--train_gossip", "--test_gossip",
                         "--output_dir", str(tmp_path / "out")])
    an, ag, ao = config.split_namespaces(args)
    from desco_amd.data import STANDARD_QUERY_IDS
    rep = driver.main(an, ag, ao, train_neighborhood=True, train_gossip=True, test_gossip=True,
                      atlas_query_ids=STANDARD_QUERY_IDS, output_dir=str(tmp_path / "out"), data_root=root)
    out = tmp_path / "out"
    for f in ["config_TOY_test.txt", "neighborhood_graphlet_TOY_test.csv", "gossip_graphlet_TOY_test.csv",
              "gossip_gate_TOY_test.csv", "neighborhood_node_TOY_test_results.csv",
              "neighborhood_node_TOY_test_index.csv", "gossip_node_TOY_test_results.csv",
              "test_nxgraph_TOY_test.pk", "graphlet_count_TOY_test.csv", "graphlet_truth_TOY_test.csv",
              "analyze_results_TOY_test.txt"]:
        assert (out / f).exists(), f
    assert len(rep["graphlet_norm_mse_gossip"]) == 3 and all(np.isfinite(rep["graphlet_mae_gossip"]))
    assert (tmp_path / "ckpt_n" / "last.ckpt").exists() and (tmp_path / "ckpt_g" / "last.ckpt").exists()
    # second run: inference only from the saved checkpoints
    rep2 = driver.main(an, ag, ao, train_neighborhood=False, train_gossip=False, test_gossip=True,
                       neighborhood_checkpoint=str(tmp_path / "ckpt_n" / "last.ckpt"),
                       gossip_checkpoint=str(tmp_path / "ckpt_g" / "last.ckpt"),
                       atlas_query_ids=STANDARD_QUERY_IDS, output_dir=str(tmp_path / "out2"), data_root=root)
    assert len(rep2["graphlet_mae_neighborhood"]) == 3


def test_driver_with_two_gpus_starts_its_own_ranks(tmp_path):
    """``main.py --gpu 0 1`` (reference: main.py:242-255 hands the devices to a "ddp" Trainer): the
    driver starts one process per GPU itself, trains both stages data parallel, shards the predict
    passes, and rank 0 writes the reference's artefacts.  (DESCO_SHARE_GPU=1: both ranks on the one
    GPU of the test box, gloo.)"""
    import subprocess
    import sys
    from helpers import golden_graphs
    root = str(tmp_path / "data")
    _write_tu(root, "TOY", golden_graphs(max_n=30))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DESCO_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(repo, "main.py"), "--gpu", "0", "1", "--data_root", root,
           "--train_dataset", "TOY_train", "--valid_dataset", "TOY_val", "--test_dataset", "TOY_test",
           "--neigh_epoch_num", "2", "--gossip_epoch_num", "1",
           "--neigh_batch_size", "32", "--gossip_batch_size", "2",
           "--neigh_model_path", str(tmp_path / "ckpt_n"), "--gossip_model_path", str(tmp_path / "ckpt_g"),
           "--train_neigh", "--train_gossip", "--test_gossip", "--output_dir", str(tmp_path / "out")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = tmp_path / "out"
    for f in ["neighborhood_graphlet_TOY_test.csv", "gossip_graphlet_TOY_test.csv", "gossip_node_TOY_test_results.csv",
              "analyze_results_TOY_test.txt", "graphlet_truth_TOY_test.csv"]:
        assert (out / f).exists(), f
    assert (tmp_path / "ckpt_n" / "last.ckpt").exists() and (tmp_path / "ckpt_g" / "last.ckpt").exists()
    assert p.stdout.count("done") == 1            # only rank 0 reports
