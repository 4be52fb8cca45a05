"""The numpy restatement of the HIP path's counter-based dropout (oracle/dropout.py) against the published
known-answer vectors of Philox4x32-10 (Random123, kat_vectors: `philox4x32 10 ...`), and the oracle's mask arguments."""
import numpy as np
import torch

from oracle import dropout as OD
from oracle import model as OM
from helpers import cpu_sd, make_models  # noqa: E402


KAT = [
    ((0x00000000,) * 4, (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        got = tuple(int(w) for w in OD.philox4x32_10(ctr, key))
        assert got == want, (ctr, key, [hex(g) for g in got])


def test_philox_vectorised_equals_scalar():
    r4 = np.arange(7, dtype=np.uint64)[:, None]
    col = np.arange(5, dtype=np.uint64)[None, :]
    w = OD.philox4x32_10((r4, col, 3, 0), (11, 12))
    for i in range(7):
        for j in range(5):
            s = OD.philox4x32_10((i, j, 3, 0), (11, 12))
            assert [int(x[i, j]) for x in w] == [int(x) for x in s]


def test_dropout_factor_statistics_and_layout():
    p = 0.25
    f = OD.dropout_factor(seed=5, step=9, site=2, p=p, num_rows=4001, num_cols=64)
    assert f.shape == (4001, 64) and set(np.unique(f)) == {np.float32(0.0), np.float32(1.0 / (1.0 - p))}
    n = f.size
    keep = float((f > 0).mean())
    assert abs(keep - (1 - p)) < 3 * np.sqrt(p * (1 - p) / n)
    # rows 4k..4k+3 of a column are the four words of one Philox call
    b = OD.dropout_bits(5, 9, 2, 8, 3)
    w = OD.philox4x32_10((1, 2 | (2 << 24), 9, 0), (5, 0))
    assert [int(b[4 + s, 2]) for s in range(4)] == [int(x) for x in w]
    # other site / step / seed: other masks;  p = 0 keeps everything, p = 1 nothing
    assert (OD.dropout_factor(5, 9, 1, p, 64, 64) != f[:64]).any()
    assert (OD.dropout_factor(5, 10, 2, p, 64, 64) != f[:64]).any()
    assert (OD.dropout_factor(6, 9, 2, p, 64, 64) != f[:64]).any()
    assert (OD.dropout_factor(5, 9, 2, 0.0, 64, 64) == 1.0).all()
    assert (OD.dropout_factor(5, 9, 2, 1.0, 64, 64) == 0.0).all()


def test_oracle_masks_of_ones_change_nothing_and_zero_masks_cut_the_path():
    _, gm = make_models(seed=0)
    sd = cpu_sd(gm)
    g = torch.Generator().manual_seed(3)
    N, Q = 9, 4
    x = torch.rand(N, Q, generator=g) * 5
    ei = np.array([[0, 1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 8]])
    qe = torch.randn(Q, 64, generator=g)
    base = OM.gossip_graph_to_count(sd, x, ei, qe, 2)
    ones = torch.ones(N, Q, 64)
    same = OM.gossip_graph_to_count(sd, x, ei, qe, 2, masks=([ones, ones], ones))
    assert torch.equal(base, same)
    # post_mp.1 dropped everywhere: the correction is post_mp's response to a zero vector, the same for every node
    cut = OM.gossip_graph_to_count(sd, x, ei, qe, 2, masks=([ones, ones], torch.zeros(N, Q, 64))) - x
    assert torch.allclose(cut, cut[:1].expand_as(cut))
    assert not torch.allclose(base - x, cut)
