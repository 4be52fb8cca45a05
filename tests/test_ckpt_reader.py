"""The Lightning-free checkpoint reader (desco_amd/ckpt.py) on a hand-built file that mimics the
layout of a pytorch-lightning 1.6.4 ``.ckpt`` (lightning_model.py:508-532 loads such files through
pl.LightningModule.load_from_checkpoint): pytorch_lightning classes inside the pickle resolve to
inert stand-ins, foreign callables are refused, and the models rebuild with the reference's
hetero conversion and state-dict keys."""
import argparse
import collections
import os
import pickle
import sys
import types

import pytest
import torch

from desco_amd.ckpt import AttributeDict, load_checkpoint
from helpers import gossip_args, neigh_args


def _fake_lightning():
    """Temporarily importable pytorch_lightning.* classes so that torch.save can pickle references to
    them (the package itself is not installed)."""
    mods = {}
    for name in ("pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing",
                 "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint",
                 "pytorch_lightning.utilities.enums"):
        mods[name] = types.ModuleType(name)
    class AttributeDictPL(dict):
        pass
    AttributeDictPL.__name__ = AttributeDictPL.__qualname__ = "AttributeDict"
    AttributeDictPL.__module__ = "pytorch_lightning.utilities.parsing"
    mods["pytorch_lightning.utilities.parsing"].AttributeDict = AttributeDictPL
    class ModelCheckpoint:
        def __init__(self):
            self.best_model_path = "/authors/box/epoch=3.ckpt"
    ModelCheckpoint.__module__ = "pytorch_lightning.callbacks.model_checkpoint"
    ModelCheckpoint.__qualname__ = "ModelCheckpoint"
    mods["pytorch_lightning.callbacks.model_checkpoint"].ModelCheckpoint = ModelCheckpoint
    return mods, AttributeDictPL, ModelCheckpoint


def _write_lightning_like(path, model, use_attrdict=True):
    mods, AD, MC = _fake_lightning()
    sys.modules.update(mods)
    try:
        hp = dict(model.hparams_dict)
        ckpt = {
            "epoch": 3, "global_step": 1444, "pytorch-lightning_version": "1.6.4",
            "state_dict": collections.OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items()),
            "loops": {"fit_loop": {"state_dict": {}, "epoch_progress": {"total": {"ready": 4}}}},
            "callbacks": {"ModelCheckpoint{'monitor': 'neighborhood_counting_val_loss'}":
                          {"best_model_score": torch.tensor(0.5), "best_model_path": "x.ckpt", "cb": MC()}},
            "optimizer_states": [{"state": {0: {"step": torch.tensor(4.0)}}, "param_groups": [{"lr": 1e-4}]}],
            "lr_schedulers": [{"best": 0.5, "num_bad_epochs": 0}],
            "hparams_name": "kwargs",
            "hyper_parameters": AD(hp) if use_attrdict else hp,
        }
        torch.save(ckpt, path)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert "pytorch_lightning" not in sys.modules


def test_reads_lightning_layout_and_rebuilds_neighborhood_model(tmp_path):
    from desco_amd.lightning_model import NeighborhoodCountingModel
    torch.manual_seed(0)
    nm = NeighborhoodCountingModel(1, 64, neigh_args()).to_hetero_old(True, True)
    p = str(tmp_path / "neigh.ckpt")
    _write_lightning_like(p, nm)
    with pytest.raises(Exception):            # the stock unpickler needs pytorch_lightning
        torch.load(p, weights_only=False)
    ck = load_checkpoint(p)
    assert type(ck["hyper_parameters"]) is dict and isinstance(ck["hyper_parameters"]["args"], argparse.Namespace)
    assert ck["pytorch-lightning_version"] == "1.6.4"
    nm2 = NeighborhoodCountingModel.load_from_checkpoint(p)
    assert nm2.emb_model.gnn_core.node_types == ["count", "canonical"]        # hetero conversion re-applied
    sd, sd2 = nm.state_dict(), nm2.state_dict()
    assert list(sd) == list(sd2)
    assert "emb_model.gnn_core.convs.0.count__union_triangle__canonical.lin.weight" in sd2
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k


def test_reads_gossip_checkpoint_with_plain_dict_hparams(tmp_path):
    from desco_amd.lightning_model import GossipCountingModel
    torch.manual_seed(1)
    gm = GossipCountingModel(1, 64, gossip_args(), emb_channels=64, input_pattern_emb=True)
    p = str(tmp_path / "gossip.ckpt")
    _write_lightning_like(p, gm, use_attrdict=False)
    gm2 = GossipCountingModel.load_from_checkpoint(p)
    assert gm2.kwargs["baseline"] == "gossip" and gm2.kwargs["emb_channels"] == 64
    for k, v in gm.state_dict().items():
        assert torch.equal(v, gm2.state_dict()[k]), k


def test_own_checkpoints_round_trip(tmp_path):
    from desco_amd.lightning_model import NeighborhoodCountingModel
    torch.manual_seed(2)
    nm = NeighborhoodCountingModel(1, 64, neigh_args(use_tconv=False)).to_hetero_old(False, False)
    p = str(tmp_path / "own.ckpt")
    nm.save_checkpoint(p)
    nm2 = NeighborhoodCountingModel.load_from_checkpoint(p)
    assert "emb_model.gnn_core.convs.0.count__union__canonical.lin.weight" in nm2.state_dict()


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_refuses_foreign_callables(tmp_path):
    p = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {}, "hyper_parameters": {"args": argparse.Namespace()}, "x": _Evil()}, p)
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        load_checkpoint(p)


def _payload(module, name, *args):
    """A protocol-4 pickle that calls ``module.name(*args)`` on load (GLOBAL via STACK_GLOBAL)."""
    import pickletools  # noqa: F401
    out = pickle.PROTO + bytes([4])
    for s_ in (module, name):
        b = s_.encode()
        out += pickle.SHORT_BINUNICODE + bytes([len(b)]) + b
    out += pickle.STACK_GLOBAL + pickle.MARK
    for a in args:
        b = a.encode()
        out += pickle.SHORT_BINUNICODE + bytes([len(b)]) + b
    out += pickle.TUPLE + pickle.REDUCE + pickle.STOP
    return out


@pytest.mark.parametrize("module,name,args", [
    ("torch.utils.collect_env", "run", ("echo pwned > /tmp/desco_ckpt_pwned",)),
    ("torch.storage", "_load_from_bytes", ("x",)),
    ("numpy.testing._private.utils", "runstring", ("import os", "{}")),
    ("builtins", "object.__getattribute__", ()),
    ("collections", "_sys.modules", ()),
    ("functools", "partial", ()),
    ("builtins", "eval", ("1+1",)),
    ("builtins", "getattr", ()),
    ("torch", "load", ("x",)),
    ("os", "system", ("true",)),
])
def test_restricted_unpickler_is_an_exact_allow_list(module, name, args):
    """module-wide allow-lists let a checkpoint run code through torch / numpy helpers
    or dotted protocol-4 names.  Every one of these must be refused before anything is called."""
    from desco_amd.ckpt import _PickleModule
    if os.path.exists("/tmp/desco_ckpt_pwned"):
        os.remove("/tmp/desco_ckpt_pwned")
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        _PickleModule.loads(_payload(module, name, *args))
    assert not os.path.exists("/tmp/desco_ckpt_pwned")


def test_collect_env_payload_inside_a_checkpoint_file(tmp_path):
    class _Evil2:
        def __reduce__(self):
            import torch.utils.collect_env as ce
            return (ce.run, ("echo pwned > /tmp/desco_ckpt_pwned2",))
    p = str(tmp_path / "evil2.ckpt")
    torch.save({"state_dict": {}, "hyper_parameters": {"args": argparse.Namespace(), "x": _Evil2()}}, p)
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        load_checkpoint(p)
    assert not os.path.exists("/tmp/desco_ckpt_pwned2")


def test_attribute_dict_standin():
    d = AttributeDict(a=1)
    d.b = 2
    assert d.a == 1 and d["b"] == 2
    with pytest.raises(AttributeError):
        d.c
