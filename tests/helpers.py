"""Shared helpers for the parity tests."""
import argparse
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def neigh_args(**over):
    a = dict(layer_num=8, conv_type="SAGE", use_hetero=True, dropout=0.0, depth=4, lr=1e-4,
             weight_decay=0.0, use_tconv=True, hidden_dim=64, input_dim=1, batch_size=512)
    a.update(over)
    return argparse.Namespace(**a)


def gossip_args(**over):
    a = dict(layer_num=2, conv_type="GOSSIP", use_hetero=False, dropout=0.0, lr=1e-3,
             weight_decay=0.0, hidden_dim=64, batch_size=256)
    a.update(over)
    return argparse.Namespace(**a)


def make_models(seed=0, scale_bias=True, gains=(1.3, 1.4)):
    """Seeded NeighborhoodCountingModel + GossipCountingModel (reference key names).
    ``gains``: widening of the (neighborhood, gossip) weight matrices (dense Syn-shaped batches sum
    over many more neighbours per row and want a smaller one to stay O(1))."""
    from desco_amd.lightning_model import GossipCountingModel, NeighborhoodCountingModel
    torch.manual_seed(seed)
    nm = NeighborhoodCountingModel(1, 64, neigh_args()).to_hetero_old(True, True)
    gm = GossipCountingModel(1, 64, gossip_args(), emb_channels=64, input_pattern_emb=True)
    if scale_bias:
        # default nn.Linear init makes 8 relu layers collapse to ~constant outputs; widen the
        # weights so that parity tests see structure-dependent, O(1) values
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for m, gain in ((nm, gains[0]), (gm, gains[1])):
                for name, p in m.named_parameters():
                    if p.dim() == 2:
                        p.mul_(gain)
                    else:
                        p.add_(0.1 * torch.randn(p.shape, generator=g))
    return nm, gm


def golden_graphs(max_n=10 ** 9):
    with open(os.path.join(GOLDEN, "partition_golden.json")) as f:
        d = json.load(f)
    return [(g["n"], [tuple(e) for e in g["edges"]]) for g in d["graphs"] if g["n"] <= max_n]


def standard_queries():
    with open(os.path.join(GOLDEN, "queries.json")) as f:
        d = json.load(f)
    return d["query_ids"], [(q["n"], [tuple(e) for e in q["edges"]]) for q in d["queries"]]


def cpu_sd(model):
    return {k: v.detach().cpu().float() for k, v in model.state_dict().items()}


def report(name, got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    d = (got - ref).abs()
    rel = d / ref.abs().clamp_min(1e-6)
    print(f"[parity] {name}: max|d|={d.max().item():.3e} max rel={rel.max().item():.3e} "
          f"ref max={ref.abs().max().item():.3e}")
    return d.max().item()


# ---- THE float tolerance of the HIP-vs-oracle gates (quoted in BASELINE.md section 2, DESIGN.md section 2, README) -------
# Every comparison of model outputs with the CPU oracle -- log-space head outputs ("logits"), embeddings, counts --
# uses ONE metric and ONE number:   max |got - ref| / (1 + |ref|)  <=  LOGIT_TOL,
# counts c = 2**logit - 1 being compared as sign(c) log2(1 + |c|) (so that the gate means the same for a count of 0.1 and of 1e6).
# Measured worst cases on the parity sets (printed by every test): logits 1.5e-5, node-level counts after both stages 3.4e-5 (neighborhood counts 1.8e-5),
# embeddings 1.3e-6, gossip corrections 3.8e-6 absolute on magnitude 2.5.  5e-5 is 1.5x the worst of them (3x the worst logit figure) and
# 20x below the smallest perturbation a wrong index or weight produces (see test_the_gate_catches_a_1e4_logit_error).
LOGIT_TOL = 5e-5


# Training gates (the only other tolerances of the GPU suite; each is used through the helper below it):
#   LOSS_TOL   relative difference of a scalar training loss (a mean over B*Q log-space errors) between the HIP step
#              and the oracle's autograd step: the loss inherits the logits' 5e-5 log-space bound divided by its own
#              magnitude (losses of 0.5-10), so 1e-4 relative.
#   GRAD_TOL   max |g - g_ref| / max |g_ref| per parameter tensor.  Gradients are sums over 1e4-1e6 fp32 terms in a
#              different order than autograd's (split-K weight gradients, fused backward kernels), with cancellation:
#              measured worst 4e-4 (neighborhood), 1.6e-3 (gossip: 29 query passes accumulate into one gradient).
LOSS_TOL = 1e-4
GRAD_TOL = 2e-3
GOSSIP_GRAD_TOL = 5e-3


def assert_loss_close(name, got, ref, tol=LOSS_TOL):
    g, r = float(got), float(ref)
    err = abs(g - r) / max(abs(r), 1e-12)
    print(f"[gate] {name}: loss {g:.6f} vs {r:.6f}, relative difference {err:.2e} (gate {tol:.0e})")
    assert err <= tol, f"{name}: loss differs by {err:.3e} relative (gate {tol:.0e})"
    return err


def assert_grad_close(name, got, ref, tol=GRAD_TOL):
    """max |got - ref| / max |ref| of one parameter's gradient"""
    scale = float(ref.abs().max()) + 1e-8
    err = float((got.detach().cpu() - ref).abs().max()) / scale
    assert err <= tol, f"{name}: gradient differs by {err:.3e} of its largest element {scale:.3e} (gate {tol:.0e})"
    return err


def log_space_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs() / (1.0 + ref.abs())).max().item() if ref.numel() else 0.0


def assert_logits_close(name, got, ref, tol=LOGIT_TOL):
    """Gate on log-space quantities (head outputs, embeddings): prints the measured value, fails above ``tol``."""
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = log_space_err(got, ref)
    print(f"[gate] {name}: max |got - ref| / (1 + |ref|) = {err:.2e} (gate {tol:.0e})")
    assert err <= tol, f"{name}: {err:.3e} exceeds the gate {tol:.0e}"
    return err


def assert_counts_close(name, got, ref, tol=LOGIT_TOL):
    """Gate on counts (2**logit - 1, their node / graph sums, gossip-corrected counts), compared in signed log space:
    slog(c) = sign(c) log2(1 + |c|).  For c >= 0 that is log2(1 + c) = the logit; for the rare negative values (counts
    of absent patterns are 2**logit - 1 in (-1, 0), the gossip stage adds a signed correction) it stays smooth where
    log2(1 + c) would blow a 4e-6 absolute difference near c = -1 up to any size."""
    slog = lambda c: torch.sign(c) * torch.log2(1.0 + c.abs())          # noqa: E731
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    return assert_logits_close(name + " [signed log2(1+|count|)]", slog(g), slog(r), tol)


def random_family_graphs(seed, count):
    """Graph families the fixtures do not hold: stars, wheels, paths, cycles, cliques with tails, grids, barbells, random
    trees, G(n,p) at three densities, graphs with isolated nodes and with several components."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        kind = i % 11
        n = int(rng.integers(4, 40))
        e = set()
        add = lambda a, b: e.add((min(a, b), max(a, b))) if a != b else None       # noqa: E731
        if kind == 0:                                   # star (hub row)
            c = int(rng.integers(n))
            [add(c, v) for v in range(n)]
        elif kind == 1:                                 # wheel
            [add(0, v) for v in range(1, n)]
            [add(v, v % (n - 1) + 1) for v in range(1, n)]
        elif kind == 2:                                 # path
            [add(v, v + 1) for v in range(n - 1)]
        elif kind == 3:                                 # cycle + chords
            [add(v, (v + 1) % n) for v in range(n)]
            [add(int(rng.integers(n)), int(rng.integers(n))) for _ in range(n // 4)]
        elif kind == 4:                                 # clique with a tail (all-triangle edges + tride edges)
            k = min(n, int(rng.integers(4, 9)))
            [add(a, b) for a in range(k) for b in range(a + 1, k)]
            [add(v, v + 1) for v in range(k - 1, n - 1)]
        elif kind == 5:                                 # grid
            w = max(2, int(np.sqrt(n)))
            n = w * w
            [add(r * w + c, r * w + c + 1) for r in range(w) for c in range(w - 1)]
            [add(r * w + c, (r + 1) * w + c) for r in range(w - 1) for c in range(w)]
        elif kind == 6:                                 # barbell
            k = max(3, n // 3)
            n = 2 * k + 2
            [add(a, b) for a in range(k) for b in range(a + 1, k)]
            [add(k + 2 + a, k + 2 + b) for a in range(k) for b in range(a + 1, k)]
            add(k - 1, k), add(k, k + 1), add(k + 1, k + 2)
        elif kind == 7:                                 # random tree, random labels
            [add(v, int(rng.integers(v))) for v in range(1, n)]
        else:                                           # G(n, p), p = 0.08 / 0.2 / 0.45; isolated nodes and components stay
            p = (0.08, 0.2, 0.45)[kind - 8]
            m = rng.random((n, n)) < p
            [add(a, b) for a in range(n) for b in range(a + 1, n) if m[a, b]]
        perm = rng.permutation(n)                       # canonical neighborhoods depend on the node ids: shuffle them
        edges = sorted((int(min(perm[a], perm[b])), int(max(perm[a], perm[b]))) for a, b in e)
        if edges:
            out.append((n, edges))
    return out
