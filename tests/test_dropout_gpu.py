"""Gossip training at the reference's DEFAULT configuration: --gossip_dropout 0.01 (config.py:316 of the reference) --
F.dropout behind each GossipConv layer's relu (gnn_model.py:274) and post_mp.1 = nn.Dropout (:46).  The HIP path
multiplies by counter-based factors inside the epilogues that produce the dropped tensors and regenerates them in the
backward kernels (csrc/common_device.hpp, dropout.hip); torch's mask stream cannot be reproduced, so parity is held in
two halves: the factor tensor is the documented function of (seed, step, site, row, col) (bit-exact against
oracle/dropout.py, which the Random123 known-answer vectors pin), and with THAT mask injected into the oracle, loss and
gradients agree within the suite's training gates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from desco_amd import autograd as AG  # noqa: E402
from desco_amd import gnn_model as GM  # noqa: E402
from desco_amd import ops  # noqa: E402
from desco_amd.batch import GossipBatch  # noqa: E402
from desco_amd.graphs import GraphSet  # noqa: E402
from oracle import dropout as OD  # noqa: E402
from oracle import model as OM  # noqa: E402

from helpers import (GOSSIP_GRAD_TOL, assert_grad_close, assert_logits_close, assert_loss_close, golden_graphs,  # noqa: E402
                     gossip_args, make_models, standard_queries)

DEV = "cuda"


def gossip_model(p, seed=0):
    """make_models' gossip model (same seeded weights) with dropout p"""
    from desco_amd.lightning_model import GossipCountingModel
    _, ref = make_models(seed=seed)
    gm = GossipCountingModel(1, 64, gossip_args(dropout=p), emb_channels=64, input_pattern_emb=True)
    gm.load_state_dict(ref.state_dict())
    return gm.to(DEV)


@pytest.fixture(scope="module")
def setup():
    nm, _ = make_models(seed=0)
    qids, queries = standard_queries()
    nm = nm.to(DEV)
    nm.set_queries(qids)
    graphs = golden_graphs(max_n=41)[:12]
    gs = GraphSet.from_edge_lists(graphs)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(gs.num_nodes, len(queries), generator=g) * 20
    y = torch.floor(torch.rand(gs.num_nodes, len(queries), generator=g) * 25)
    qemb = nm.get_query_emb().detach()
    return gs, x, y, qemb


def step(gm, gs, x, y, qemb, seed=None, step_no=0):
    """one training forward + backward; returns (loss, {name: grad})"""
    if seed is not None:
        ops.manual_seed(seed, step=step_no)
    gm.set_query_emb(qemb)
    batch = GossipBatch(gs, DEV, x=x, y=y)
    gm.zero_grad()
    gm.train()
    loss = gm.train_forward(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach(), {n: p.grad.detach().clone() for n, p in gm.named_parameters() if p.grad is not None}, batch


def test_mask_kernel_is_the_documented_function():
    """desco_dropout_mask_f32 == oracle/dropout.py bit for bit (ragged row counts, several sites / steps / p)."""
    for seed, stp, site, p, R, C in ((1234, 0, 0, 0.01, 1001, 64), (2 ** 40 + 17, 2 ** 33 + 5, 2, 0.5, 130, 64),
                                     (7, 3, 255, 0.25, 7, 256), (9, 1, 1, 0.0, 5, 64), (9, 1, 1, 1.0, 5, 64)):
        ops.manual_seed(seed, step=stp)
        key = ops.rng_next(torch.device(DEV, torch.cuda.current_device()))
        got = ops.dropout_mask(ops.DropSite(key, site, p), R, C).cpu().numpy()
        want = OD.dropout_factor(seed, stp, site, p, R, C)
        assert np.array_equal(got, want), (seed, stp, site, p)
        # the counter moved on: the next key is (seed, step + 1)
        assert ops.rng_state(DEV).cpu().tolist() == [seed, stp + 1]
        assert key.cpu().tolist() == [seed, stp]


def test_keep_rate_seeds_and_repeats():
    dev = torch.device(DEV, torch.cuda.current_device())
    p, R, C = 0.01, 200_000, 64
    ops.manual_seed(11)
    k0 = ops.rng_next(dev)
    m0 = ops.dropout_mask(ops.DropSite(k0, 0, p), R, C)
    keep = float((m0 > 0).double().mean())
    sigma = np.sqrt(p * (1 - p) / (R * C))
    print(f"[dropout] keep rate {keep:.6f} vs {1 - p} ({abs(keep - (1 - p)) / sigma:.2f} sigma)")
    assert abs(keep - (1 - p)) < 3 * sigma
    assert float(m0.max()) == pytest.approx(1.0 / (1.0 - p), rel=1e-7)
    k1 = ops.rng_next(dev)                                # next step of the same seed: another mask
    assert not torch.equal(ops.dropout_mask(ops.DropSite(k1, 0, p), R, C), m0)
    assert not torch.equal(ops.dropout_mask(ops.DropSite(k0, 1, p), R, C), m0)      # another site
    ops.manual_seed(12)
    assert not torch.equal(ops.dropout_mask(ops.DropSite(ops.rng_next(dev), 0, p), R, C), m0)   # another seed
    ops.manual_seed(11)
    assert torch.equal(ops.dropout_mask(ops.DropSite(ops.rng_next(dev), 0, p), R, C), m0)       # same seed: same bits


def test_p_zero_is_bit_identical_to_the_path_without_dropout(setup):
    gs, x, y, qemb = setup
    gm = gossip_model(0.0)
    l0, g0, _ = step(gm, gs, x, y, qemb)
    GM.DROPOUT_AT_ZERO = True
    try:
        l1, g1, _ = step(gm, gs, x, y, qemb, seed=5)
    finally:
        GM.DROPOUT_AT_ZERO = False
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys()
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


@pytest.mark.parametrize("p", [0.01, 0.3])
def test_gossip_training_with_dropout_vs_oracle_with_the_same_mask(setup, p):
    """loss and every gradient of one training step at dropout p (0.01 = the reference default) against torch autograd
    through the CPU oracle fed with the factor tensors the kernels used (exported by desco_dropout_mask_f32 for the
    step's key)."""
    gs, x, y, qemb = setup
    gm = gossip_model(p)
    seed = 4242
    loss, grads, batch = step(gm, gs, x, y, qemb, seed=seed, step_no=3)
    N, Q = x.shape
    key = torch.tensor([seed, 3], dtype=torch.int64, device=DEV)
    T = AG.GossipTrunk
    masks = [ops.dropout_mask(ops.DropSite(key, s, p), N * Q, 64).cpu().view(N, Q, 64)
             for s in (T.SITE_H1, T.SITE_H2, T.SITE_POST)]
    dropped = [float((m == 0).double().mean()) for m in masks]
    print(f"[dropout] p = {p}: dropped fractions per site {dropped}")
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in gm.state_dict().items()}
    ref_loss = OM.gossip_loss(sd, x, y, batch.edge_index.numpy(), qemb.cpu(), 2, masks=(masks[:2], masks[2]))
    ref_loss.backward()
    assert_loss_close(f"gossip train loss, dropout {p}", loss, ref_loss.detach())
    # the mask matters: without it the oracle's loss is measurably different
    plain = OM.gossip_loss({k: v.detach() for k, v in sd.items()}, x, y, batch.edge_index.numpy(), qemb.cpu(), 2)
    assert abs(float(plain) - float(ref_loss.detach())) / abs(float(ref_loss.detach())) > 1e-4
    worst = 0.0
    for name, prm in gm.named_parameters():
        ref = sd[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        worst = max(worst, assert_grad_close(name, grads[name], ref, tol=GOSSIP_GRAD_TOL))
    print(f"[parity] gossip worst relative gradient error at dropout {p}: {worst:.3e}")


def test_steps_draw_fresh_masks_and_eval_mode_draws_none(setup):
    gs, x, y, qemb = setup
    gm = gossip_model(0.3)
    la, ga, batch = step(gm, gs, x, y, qemb, seed=1)
    lb, gb, _ = step(gm, gs, x, y, qemb)                   # the counter moved on
    assert not torch.equal(la, lb)
    lc, gc, _ = step(gm, gs, x, y, qemb, seed=1)           # same (seed, step): the same step bit for bit
    assert torch.equal(la, lc)
    for n in ga:
        assert torch.equal(ga[n], gc[n]), n
    # eval mode (validation_step / test_step under Trainer's model.eval()): no dropout -- the training-path forward
    # equals the inference kernels' output and is repeatable
    gm.eval()
    before = ops.rng_state(DEV).cpu().tolist()
    with torch.no_grad():
        p1 = gm.emb_model(batch, query_emb=qemb)
        p2 = gm.graph_to_count(batch)
    assert ops.rng_state(DEV).cpu().tolist() == before
    assert_logits_close("eval-mode training path vs inference kernels (dropout model)", p1, p2)


def test_trainer_replays_the_gossip_step_with_dropout(tmp_path, setup):
    """Trainer(graph_capture=True) on a model with dropout 0.01: the captured step draws a new mask on every replay
    (the key is read from device memory), and the replayed run equals the eager run bit for bit."""
    from desco_amd.trainer import Trainer
    gs, x, y, qemb = setup

    class DM:
        def __init__(self):
            self.b = [GossipBatch(gs, DEV, x=x, y=y)]

        def train_dataloader(self):
            return self.b

        def val_dataloader(self):
            return self.b

    outs = []
    for capture in (False, True):
        gm = gossip_model(0.01)
        gm.set_query_emb(qemb)
        ops.manual_seed(99)
        tr = Trainer(max_epochs=4, default_root_dir=str(tmp_path / f"c{int(capture)}"), graph_capture=capture)
        tr.fit(gm, DM())
        torch.cuda.synchronize()
        outs.append(({k: v.detach().clone() for k, v in gm.state_dict().items()}, ops.rng_state(DEV).cpu().tolist(),
                     [h["gossip_counting_val_loss"] for h in tr.history]))
    (sa, ra, ha), (sb, rb, hb) = outs
    assert ra == rb == [99, 4]                               # four training steps drew four keys in both runs
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert ha == hb


@pytest.mark.parametrize("p", [0.2])
def test_neighborhood_training_with_dropout_vs_oracle_with_the_same_masks(p):
    """--neigh_dropout p (default 0.0, config.py:251): F.dropout behind every SAGE layer's relu of BOTH models (target
    and query, gnn_model.py:274) and post_mp.1 (:46) inside the fused training nodes (autograd.ShmpTrunk: the layer
    products' epilogues; ShmpTrunkSmall: the per-graph kernels; Mlp).  One training step's loss and gradients against
    torch autograd through the CPU oracle fed with the factor tensors of the step's two keys (query model: the first
    key drawn, target model: the second)."""
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.lightning_model import NeighborhoodCountingModel
    from desco_amd.partition import build_partition
    from desco_amd import gnn_model as GM
    from oracle import partition as OP
    from helpers import GRAD_TOL, neigh_args
    ref0, _ = make_models(seed=0)
    nm = NeighborhoodCountingModel(1, 64, neigh_args(dropout=p)).to_hetero_old(True, True)
    nm.load_state_dict(ref0.state_dict())
    nm = nm.to(DEV)
    qids, queries = standard_queries()
    nm.set_queries(qids)
    graphs = golden_graphs(max_n=41)[:10]
    part = build_partition(GraphSet.from_edge_lists(graphs), 4)
    _, _, neighs = OP.neighborhood_dataset(graphs, 4)
    g = torch.Generator().manual_seed(9)
    y = torch.floor(torch.rand(part.num_neigh, len(queries), generator=g) ** 3 * 40)
    batch = NeighborhoodBatch(part, DEV, y=y)
    seed = 31337
    ops.manual_seed(seed, step=10)
    nm.train()
    nm.zero_grad()
    loss = nm.train_forward(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert ops.rng_state(DEV).cpu().tolist() == [seed, 12]             # two keys: query model, target model
    kq = torch.tensor([seed, 10], dtype=torch.int64, device=DEV)
    kt = torch.tensor([seed, 11], dtype=torch.int64, device=DEV)
    Nc, B, nq = batch.num_count, batch.num_graphs, sum(n for n, _ in queries)

    def fac(key, site, pp, rows):
        return ops.dropout_mask(ops.DropSite(key, site, pp), rows, 64).cpu()
    masks_t = ([{"count": fac(kt, 2 * l, p, Nc), "canonical": fac(kt, 2 * l + 1, p, B)} for l in range(8)],
               fac(kt, GM.POST_DROP_SITE, p, B))
    masks_q = ([{"union_node": fac(kq, 2 * l, p, nq)} for l in range(8)], fac(kq, GM.POST_DROP_SITE, p, len(queries)))
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in nm.state_dict().items()}
    ref_loss = OM.neighborhood_loss(sd, OP.neighborhood_batch(neighs), OP.query_batch(queries), y, emulate_quirk=False,
                                    masks_t=masks_t, masks_q=masks_q)
    ref_loss.backward()
    assert_loss_close(f"neighborhood train loss, dropout {p}", loss.detach(), ref_loss.detach())
    plain = OM.neighborhood_loss({k: v.detach() for k, v in sd.items()}, OP.neighborhood_batch(neighs),
                                 OP.query_batch(queries), y, emulate_quirk=False)
    assert abs(float(plain) - float(ref_loss.detach())) / abs(float(ref_loss.detach())) > 1e-3     # the masks matter
    worst, checked = 0.0, 0
    for name, prm in nm.named_parameters():
        ref = sd[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        worst = max(worst, assert_grad_close(name, prm.grad, ref, tol=GRAD_TOL))
        checked += 1
    print(f"[parity] neighborhood worst relative gradient error at dropout {p} over {checked} tensors: {worst:.3e}")
    assert checked > 150
    # eval mode: no dropout, no key drawn
    nm.eval()
    before = ops.rng_state(DEV).cpu().tolist()
    with torch.no_grad():
        a = nm.graph_to_count(batch)
        b = nm.graph_to_count(batch)
    assert torch.equal(a, b) and ops.rng_state(DEV).cpu().tolist() == before


def test_trainer_replays_the_neighborhood_step_with_dropout(tmp_path):
    """Trainer(graph_capture=True) on the neighborhood model at --neigh_dropout 0.1: both models' keys are drawn inside
    the captured step (two rng_next launches on the capturing stream, the query pass forked after them), so every replay
    draws fresh masks, and the replayed run equals the eager run bit for bit."""
    from desco_amd.batch import NeighborhoodBatch
    from desco_amd.lightning_model import NeighborhoodCountingModel
    from desco_amd.partition import build_partition
    from desco_amd.trainer import Trainer
    from helpers import neigh_args
    qids, queries = standard_queries()
    part = build_partition(GraphSet.from_edge_lists(golden_graphs(max_n=41)[:12]), 4)
    g = torch.Generator().manual_seed(9)
    y = torch.floor(torch.rand(part.num_neigh, len(queries), generator=g) ** 3 * 40)
    cuts = [0, part.num_neigh // 2, part.num_neigh]

    class DM:
        def _mk(self):
            return [NeighborhoodBatch(part.slice(a, b), DEV, y=y[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]

        def train_dataloader(self):
            return self._mk()

        def val_dataloader(self):
            return self._mk()[:1]

    outs = []
    for capture in (False, True):
        ref0, _ = make_models(seed=2)
        nm = NeighborhoodCountingModel(1, 64, neigh_args(dropout=0.1)).to_hetero_old(True, True)
        nm.load_state_dict(ref0.state_dict())
        nm = nm.to(DEV)
        nm.set_queries(qids)
        ops.manual_seed(5)
        tr = Trainer(max_epochs=3, default_root_dir=str(tmp_path / f"n{int(capture)}"), graph_capture=capture)
        tr.fit(nm, DM())
        torch.cuda.synchronize()
        outs.append(({k: v.detach().clone() for k, v in nm.state_dict().items()}, ops.rng_state(DEV).cpu().tolist(),
                     [h["neighborhood_counting_val_loss"] for h in tr.history]))
    (sa, ra, ha), (sb, rb, hb) = outs
    assert ra == rb == [5, 12]                    # 6 training steps x 2 keys (query model, target model)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert ha == hb
