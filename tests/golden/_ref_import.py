"""Import harness for the READ-ONLY reference at /root/reference (build container only).

The reference's integer path (canonical partition, query ids, VF2 ground truth, metrics)
is pure networkx/numpy/torch, but its modules import torch_geometric / pytorch_lightning /
torch_scatter / torch_sparse / deepsnap / ogb / seaborn at module scope, none of which are
installed here.  This harness serves *inert* placeholder modules for those names so that the
reference's own functions that never touch a stubbed symbol at run time can be executed
unmodified.  Nothing from the reference is copied; this file is only
used by make_golden.py to produce data fixtures and never ships to / runs on the GPU box.
"""
import importlib.abc
import importlib.machinery
import sys
import types

REF_ROOT = "/root/reference"
_STUBBED = {
    "torch_geometric", "pytorch_lightning", "torch_scatter", "torch_sparse",
    "deepsnap", "ogb", "seaborn",
}


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)

    def __getitem__(cls, item):
        return cls


def _make_dummy(name):
    return _Meta(name, (object,), {
        "__init__": lambda self, *a, **k: None,
        "__call__": lambda self, *a, **k: None,
        "__class_getitem__": classmethod(lambda cls, item: cls),
    })


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        return _make_dummy(name)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUBBED:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install():
    sys.dont_write_bytecode = True
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def load():
    """Returns the reference modules whose integer-path functions are usable."""
    install()
    import subgraph_counting.data as ref_data
    import subgraph_counting.workload as ref_workload
    import subgraph_counting.analysis as ref_analysis
    import subgraph_counting.config as ref_config
    return ref_data, ref_workload, ref_analysis, ref_config
