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

    def select(self, idx) -> "NeighborhoodPartition":
        """The neighborhoods ``idx`` (ascending indices, any subset) as a self-contained block: what
        ``slice`` does for a contiguous range."""
        idx = np.asarray(idx, dtype=np.int64)
        M = len(idx)
        cp, Nc = self.count_ptr.astype(np.int64), self.num_count
        n = cp[idx + 1] - cp[idx]
        cp2 = np.concatenate([[0], np.cumsum(n)])
        nc2 = int(cp2[-1])
        rows_old = np.repeat(cp[idx] - cp2[:-1], n) + np.arange(nc2, dtype=np.int64)
        v = self.vrowptr.astype(np.int64)
        four = np.arange(4, dtype=np.int64)
        vr_old = np.concatenate([(rows_old[:, None] * 4 + four).ravel(), ((Nc + idx)[:, None] * 4 + four).ravel()])
        deg = v[vr_old + 1] - v[vr_old]
        vr2 = np.concatenate([[0], np.cumsum(deg)])
        e_old = np.repeat(v[vr_old] - vr2[:-1], deg) + np.arange(int(vr2[-1]), dtype=np.int64)
        col_old = self.vcol[e_old].astype(np.int64)
        nb_of_vr = np.concatenate([np.repeat(np.repeat(np.arange(M, dtype=np.int64), n), 4),
                                   np.repeat(np.arange(M, dtype=np.int64), 4)])
        nb_of_e = np.repeat(nb_of_vr, deg)
        col_new = np.where(col_old < Nc, col_old + (cp2[:-1] - cp[idx])[nb_of_e], nc2 + nb_of_e)
        return NeighborhoodPartition(
            neigh_index=self.neigh_index[idx], indicator=self.indicator, count_ptr=cp2.astype(np.int32),
            count_orig=self.count_orig[rows_old], vrowptr=vr2.astype(np.int32), vcol=col_new.astype(np.int32),
            depth=self.depth, quirk_batch=self.quirk_batch)

    def degree_sorted(self, num_threads: int = 0) -> "NeighborhoodPartition":
        """The same block with the count rows of every neighborhood re-ordered by decreasing / increasing (by the parity
        of the canonical node's id inside its graph, so consecutive neighborhoods alternate) number of count -> count sources (``desco_partition_degree_sort``): fewer
        gather steps per 16-row tile of the layer kernel on dense shapes.  Row order inside a neighborhood is a
        convention of this repo; per-neighborhood results only change by fp32 summation order."""
        if self.num_count == 0:
            return self
        cp = np.ascontiguousarray(self.count_ptr, dtype=np.int32)
        vr = np.ascontiguousarray(self.vrowptr, dtype=np.int32)
        vc = np.ascontiguousarray(self.vcol, dtype=np.int32)
        co = np.ascontiguousarray(self.count_orig, dtype=np.int32)
        co2, vr2, vc2 = np.empty_like(co), np.empty_like(vr), np.empty_like(vc)
        # direction by a neighborhood-intrinsic key, the canonical node's id INSIDE ITS GRAPH: the same neighborhood gets
        # the same row order -- the same fp32 summation order -- wherever the block cuts fall and whichever rank's shard
        # holds the graph.  (Until round 5 the key was graph id + node id; graph ids are relative to the shard, so a
        # shard that started at an odd graph flipped every direction and the 2-rank chunked run differed from the 1-rank
        # run in the last bit -- tests/test_multirank_gpu.py caught it when the shard cuts moved.)
        nkey = np.ascontiguousarray(self.neigh_index[:, 1].astype(np.int64))
        _lib.check(_lib.lib().desco_partition_degree_sort(
            cp.ctypes.data, self.num_neigh, vr.ctypes.data, vc.ctypes.data, co.ctypes.data, co2.ctypes.data,
            vr2.ctypes.data, vc2.ctypes.data, nkey.ctypes.data, num_threads), "desco_partition_degree_sort")
        return NeighborhoodPartition(
            neigh_index=self.neigh_index, indicator=self.indicator, count_ptr=self.count_ptr, count_orig=co2,
            vrowptr=vr2, vcol=vc2, depth=self.depth, quirk_batch=self.quirk_batch)

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


def build_partition_device(graphs: GraphSet, depth: int = 4, device="cuda",
                           num_waves: int = 0) -> NeighborhoodPartition:
    """Same result as ``build_partition`` (quirk_batch = 0), computed on the GPU by
    desco_partition_dev_* (csrc/partition_dev.hip).  The flat CSR is produced in
    device memory; the returned object also carries host copies (the host-side API of
    NeighborhoodPartition) and keeps the device tensors in ``device_arrays`` so that
    ``NeighborhoodBatch`` does not upload them again.  Graphs too large for the per-wave LDS
    workspace raise ``ValueError`` (use ``build_partition``)."""
    import torch
    L = _lib.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("build_partition_device needs the MI355X (cuda) device; there is no CPU fallback")
    V = graphs.num_nodes
    sizes = np.diff(graphs.graph_ptr)
    n_max = int(sizes.max()) if len(sizes) else 1
    if graphs.rowptr[-1] > np.iinfo(np.int32).max:
        raise ValueError("build_partition_device: more than 2^31 directed edges")
    gp = torch.from_numpy(graphs.graph_ptr).to(dev)
    node_graph = torch.from_numpy(graphs.node_graph_ids().astype(np.int32)).to(dev)
    rowptr = torch.from_numpy(graphs.rowptr.astype(np.int32)).to(dev)
    col = torch.from_numpy(graphs.col).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if num_waves <= 0:
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        num_waves = int(min(max(4, ((V + 3) // 4) * 4), cus * 16))
    i32 = dict(device=dev, dtype=torch.int32)
    i64 = dict(device=dev, dtype=torch.int64)
    nsize, ecc, eck = (torch.empty(V, **i32) for _ in range(3))
    _lib.check(L.desco_partition_dev_count(gp.data_ptr(), node_graph.data_ptr(), rowptr.data_ptr(),
                                           col.data_ptr(), V, depth, n_max, num_waves,
                                           nsize.data_ptr(), ecc.data_ptr(), eck.data_ptr(), st),
               "desco_partition_dev_count")
    b_index, row_off, eoc, eok = (torch.empty(V, **i64) for _ in range(4))
    totals = torch.zeros(4, **i64)
    _lib.check(L.desco_partition_dev_scan(nsize.data_ptr(), ecc.data_ptr(), eck.data_ptr(), V,
                                          b_index.data_ptr(), row_off.data_ptr(), eoc.data_ptr(),
                                          eok.data_ptr(), totals.data_ptr(), st),
               "desco_partition_dev_scan")
    B, Nc, Ec, Ek = (int(t) for t in totals.cpu())          # the one host sync: output sizes
    neigh_index = torch.empty((B, 2), **i64)
    indicator = torch.empty(V, device=dev, dtype=torch.uint8)
    count_ptr = torch.empty(B + 1, **i32)
    count_orig = torch.empty(Nc, **i32)
    vrowptr = torch.empty(4 * (Nc + B) + 1, **i32)
    vcol = torch.empty(Ec + Ek, **i32)
    _lib.check(L.desco_partition_dev_fill(gp.data_ptr(), node_graph.data_ptr(), rowptr.data_ptr(),
                                          col.data_ptr(), V, depth, n_max, num_waves,
                                          b_index.data_ptr(), row_off.data_ptr(), eoc.data_ptr(),
                                          eok.data_ptr(), B, Nc, Ec, Ek, neigh_index.data_ptr(),
                                          indicator.data_ptr(), count_ptr.data_ptr(),
                                          count_orig.data_ptr(), vrowptr.data_ptr(), vcol.data_ptr(), st),
               "desco_partition_dev_fill")
    part = NeighborhoodPartition(neigh_index.cpu().numpy(), indicator.cpu().numpy().astype(bool),
                                 count_ptr.cpu().numpy(), count_orig.cpu().numpy(),
                                 vrowptr.cpu().numpy(), vcol.cpu().numpy(), depth, 0)
    part.device_arrays = {"device": dev, "count_ptr": count_ptr, "vrowptr": vrowptr, "vcol": vcol}
    return part
