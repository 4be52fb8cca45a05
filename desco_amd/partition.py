"""Canonical-partition neighborhoods as flat 4-slot CSR blocks (host side of A1-A4).

``build_partition`` calls the native builder (desco_partition_* in libdesco_hip.so); the result
replaces the reference's list of per-neighborhood ``HeteroData`` objects plus the per-item
``ToTconvHetero`` transform and the PyG collate (workload.py:243-294, transforms.py:180-255).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from .graphs import GraphSet

# relation slots of the destination-major CSR: slot = 2*(src is canonical) + (tride)
SLOT_EDGE_TYPES_COUNT_DST = (
    ("count", "union_triangle", "count"),
    ("count", "union_tride", "count"),
    ("canonical", "union_triangle", "count"),
    ("canonical", "union_tride", "count"),
)
SLOT_EDGE_TYPES_CANON_DST = (
    ("count", "union_triangle", "canonical"),
    ("count", "union_tride", "canonical"),
)


@dataclass
class NeighborhoodPartition:
    """Host arrays for B canonical neighborhoods (see include/desco_hip.h for the layout)."""
    neigh_index: np.ndarray     # int64 [B,2]  (graph id, node id in graph) == nx_neighs_index
    indicator: np.ndarray       # bool  [num_nodes]                       == nx_neighs_indicator
    count_ptr: np.ndarray       # int32 [B+1]
    count_orig: np.ndarray      # int32 [N_c] global node id of each count row
    vrowptr: np.ndarray         # int32 [4*(N_c+B)+1]
    vcol: np.ndarray            # int32 [E]
    depth: int = 4
    quirk_batch: int = 0

    @property
    def num_neigh(self) -> int:
        return len(self.count_ptr) - 1

    @property
    def num_count(self) -> int:
        return int(self.count_ptr[-1])

    @property
    def num_rows(self) -> int:
        return self.num_count + self.num_neigh

    @property
    def num_edges(self) -> int:
        return int(self.vrowptr[-1])

    def __len__(self):
        return self.num_neigh

    def slice(self, b0: int, b1: int) -> "NeighborhoodPartition":
        """Neighborhoods [b0, b1) re-based to a self-contained block (a DataLoader batch)."""
        b0, b1 = max(0, b0), min(self.num_neigh, b1)
        Nc, c0, c1 = self.num_count, int(self.count_ptr[b0]), int(self.count_ptr[b1])
        nc = c1 - c0
        v = self.vrowptr
        ec0, ec1 = int(v[4 * c0]), int(v[4 * c1])
        eb0, eb1 = int(v[4 * (Nc + b0)]), int(v[4 * (Nc + b1)])
        vr = np.concatenate([v[4 * c0:4 * c1] - ec0,
                             v[4 * (Nc + b0):4 * (Nc + b1) + 1] - eb0 + (ec1 - ec0)])
        col = np.concatenate([self.vcol[ec0:ec1], self.vcol[eb0:eb1]]).astype(np.int64)
        col = np.where(col < Nc, col - c0, col - Nc - b0 + nc)
        return NeighborhoodPartition(
            neigh_index=self.neigh_index[b0:b1], indicator=self.indicator,
            count_ptr=(self.count_ptr[b0:b1 + 1] - c0).astype(np.int32),
            count_orig=self.count_orig[c0:c1], vrowptr=vr.astype(np.int32),
            vcol=col.astype(np.int32), depth=self.depth, quirk_batch=self.quirk_batch)

    # ---- PyG-convention view (tests / interop) ---------------------------------------------
    def edge_index_dict(self) -> Dict[Tuple[str, str, str], np.ndarray]:
        """The six typed ``edge_index`` arrays of the collated HeteroData batch (PyG convention:
        row 0 = source index inside its node type, row 1 = destination index inside its type)."""
        Nc, B = self.num_count, self.num_neigh
        N = Nc + B
        cnt = np.diff(self.vrowptr.astype(np.int64))
        vrow = np.repeat(np.arange(4 * N, dtype=np.int64), cnt)
        dst, slot = vrow // 4, vrow % 4
        src = self.vcol.astype(np.int64)
        out = {}
        for s, et in enumerate(SLOT_EDGE_TYPES_COUNT_DST):
            m = (dst < Nc) & (slot == s)
            so = src[m] - (Nc if et[0] == "canonical" else 0)
            out[et] = np.stack([so, dst[m]])
        for s, et in enumerate(SLOT_EDGE_TYPES_CANON_DST):
            m = (dst >= Nc) & (slot == s)
            out[et] = np.stack([src[m], dst[m] - Nc])
        return out


def build_partition(graphs: GraphSet, depth: int = 4, quirk_batch: int = 0,
                    num_threads: int = 0) -> NeighborhoodPartition:
    L = _lib.lib()
    handle = ctypes.c_void_p()
    gp, rp, col = graphs.graph_ptr, graphs.rowptr, graphs.col
    _lib.check(L.desco_partition_build(gp.ctypes.data, graphs.num_graphs, rp.ctypes.data,
                                       col.ctypes.data, depth, quirk_batch, num_threads,
                                       ctypes.byref(handle)), "desco_partition_build")
    try:
        B, Nc, E, Nt = (ctypes.c_int64() for _ in range(4))
        _lib.check(L.desco_partition_sizes(handle, ctypes.byref(B), ctypes.byref(Nc),
                                           ctypes.byref(E), ctypes.byref(Nt)))
        B, Nc, E, Nt = B.value, Nc.value, E.value, Nt.value
        neigh_index = np.empty((B, 2), dtype=np.int64)
        indicator = np.empty(Nt, dtype=np.uint8)
        count_ptr = np.empty(B + 1, dtype=np.int32)
        count_orig = np.empty(Nc, dtype=np.int32)
        vrowptr = np.empty(4 * (Nc + B) + 1, dtype=np.int32)
        vcol = np.empty(E, dtype=np.int32)
        _lib.check(L.desco_partition_export(handle, neigh_index.ctypes.data, indicator.ctypes.data,
                                            count_ptr.ctypes.data, count_orig.ctypes.data,
                                            vrowptr.ctypes.data, vcol.ctypes.data))
    finally:
        L.desco_partition_free(handle)
    return NeighborhoodPartition(neigh_index, indicator.astype(bool), count_ptr, count_orig,
                                 vrowptr, vcol, depth, quirk_batch)
