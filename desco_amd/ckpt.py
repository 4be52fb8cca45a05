"""Reading the authors' Lightning ``.ckpt`` files without Lightning.

A Lightning 1.6.4 checkpoint is a ``torch.save`` of a dict: ``state_dict``, ``hyper_parameters``
(``input_dim``, ``hidden_dim``, ``args``: argparse.Namespace, extra kwargs) plus trainer state
(``epoch``, ``global_step``, ``pytorch-lightning_version``, ``callbacks``, ``optimizer_states``,
``lr_schedulers``, ``loops``, ``hparams_name``).  ``hyper_parameters`` is pickled as
``pytorch_lightning.utilities.parsing.AttributeDict`` (a dict subclass), and callback / loop state may
mention other ``pytorch_lightning.*`` or ``subgraph_counting.*`` classes -- unpickling those needs
packages this build does not have (reference: lightning_model.py:508-532 relies on
``pl.LightningModule.load_from_checkpoint``).

``load_checkpoint`` unpickles with a RESTRICTED class resolver: an exact allow-list of (module, name)
pairs -- the tensor rebuild functions, storages, dtypes, stdlib containers, Namespace, numpy arrays --
resolves normally, ``pytorch_lightning.*`` / ``lightning.*`` / ``subgraph_counting.*`` /
``torchmetrics.*`` names resolve to inert stand-ins (``AttributeDict`` -> a dict subclass, anything
else -> an attribute bag), and every other global is refused -- a checkpoint cannot run code here.
"""
from __future__ import annotations

import argparse
import collections
import io
import pickle
from typing import Any, Dict

import torch

_STANDIN_PREFIXES = ("pytorch_lightning", "lightning", "lightning_fabric", "subgraph_counting",
                     "torchmetrics", "deepsnap", "torch_geometric")
# Exact (module, name) pairs a tensor / container / Namespace checkpoint can legitimately refer to.
# Nothing else resolves: a module-wide allow-list is NOT safe (torch.utils.collect_env.run,
# torch.storage._load_from_bytes, numpy.testing._private.utils.runstring ... all execute code).
_TORCH_STORAGES = tuple(t + "Storage" for t in (
    "Float", "Double", "Half", "BFloat16", "Long", "Int", "Short", "Char", "Byte", "Bool",
    "ComplexFloat", "ComplexDouble", "Untyped"))
_TORCH_DTYPES = ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8",
                 "bool", "complex64", "complex128", "float", "double", "half", "long", "int", "short")
_ALLOWED = {
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"), ("torch", "strided"),
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "deque"),
    ("argparse", "Namespace"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"),
    ("copyreg", "_reconstructor"),
    ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"), ("pathlib", "Path"),
    ("datetime", "timedelta"), ("datetime", "datetime"),
}
_ALLOWED |= {("torch", n) for n in _TORCH_STORAGES} | {("torch.storage", "UntypedStorage"),
                                                       ("torch.storage", "TypedStorage")}
_ALLOWED |= {("torch", n) for n in _TORCH_DTYPES}
_ALLOWED |= {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool",
                                       "str", "bytes", "bytearray", "complex", "slice", "range", "object")}
_ALLOWED |= {("numpy.dtypes", n) for n in ("Float32DType", "Float64DType", "Int64DType", "Int32DType",
                                           "BoolDType", "UInt8DType", "Int8DType", "Int16DType")}


class AttributeDict(dict):
    """Stand-in of pytorch_lightning.utilities.parsing.AttributeDict: a dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Bag:
    """Inert stand-in for any other third-party class found in a checkpoint (callback / loop state)."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.state = state

    def __call__(self, *a, **k):      # enum-like reconstructors: Class(value)
        return self


def _standin(module: str, name: str):
    if name == "AttributeDict":
        return AttributeDict
    return type(name, (_Bag,), {"__module__": "desco_amd.ckpt.standin." + module})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        # protocol-4 dotted names ("object.__getattribute__", "_sys.modules") walk attributes of an
        # allowed global: never needed by a checkpoint, always refused
        if "." in name:
            raise pickle.UnpicklingError(f"checkpoint refers to the dotted name {module}:{name}: refused")
        top = module.split(".")[0]
        if top in _STANDIN_PREFIXES:
            return _standin(module, name)
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            f"checkpoint refers to {module}.{name}: not a tensor / container / Namespace -- refused "
            "(desco_amd.ckpt loads checkpoints with a restricted unpickler: exact allow-list)")


class _PickleModule:
    """The ``pickle_module`` interface torch.load expects."""
    __name__ = "desco_amd.ckpt.restricted_pickle"
    Unpickler = _Unpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return _Unpickler(f, **kwargs).load()

    @staticmethod
    def loads(b, **kwargs):
        return _Unpickler(io.BytesIO(b), **kwargs).load()


def _plain(o):
    """AttributeDict / OrderedDict-of-hparams -> plain containers (Namespace stays a Namespace)."""
    if isinstance(o, AttributeDict):
        return {k: _plain(v) for k, v in o.items()}
    return o


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    """{'state_dict': ..., 'hyper_parameters': {'input_dim', 'hidden_dim', 'args': Namespace, ...}, ...}
    from a file written by Lightning 1.6.x or by ``_LightningLike.save_checkpoint``."""
    ckpt = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a Lightning-style checkpoint (no 'state_dict')")
    hp = ckpt.get("hyper_parameters")
    if hp is None:
        raise ValueError(f"{path}: checkpoint has no 'hyper_parameters' (saved without save_hyperparameters)")
    hp = dict(_plain(hp))
    args = hp.get("args")
    if isinstance(args, (dict, AttributeDict)):      # Lightning can store a Namespace as a dict
        hp["args"] = argparse.Namespace(**dict(args))
    ckpt["hyper_parameters"] = hp
    ckpt["state_dict"] = collections.OrderedDict(
        (k, v) for k, v in ckpt["state_dict"].items() if isinstance(v, torch.Tensor))
    return ckpt
