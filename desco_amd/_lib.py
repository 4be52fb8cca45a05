"""ctypes binding of libdesco_hip.so (C ABI: include/desco_hip.h).

There is NO fallback: if the library is missing the import of any compute entry point raises, and
every device op raises when handed a CPU tensor.  Build with ``python -c "import __graft_entry__ as
g; g.build()"`` or ``make -C desco_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DESCO_LIB: another build of the same library (A/B runs of kernel variants, tools/debug/ab_libs.sh)
LIB_PATH = os.environ.get("DESCO_LIB") or os.path.join(_HERE, "libdesco_hip.so")
ABI_VERSION = 6

_lib = None

i64, i32, f32, f64 = c_int64, c_int, c_float, c_double
vp = c_void_p

class Dropout(ctypes.Structure):
    """desco_dropout (include/desco_hip.h)"""
    _fields_ = [("key", vp), ("site", c_uint32), ("threshold", c_uint32), ("scale", f32)]


class GemmDesc(ctypes.Structure):
    """desco_gemm_desc (include/desco_hip.h)"""
    _fields_ = [("a1", vp), ("lda1", i64), ("k1", i32), ("a2", vp), ("lda2", i64), ("k2", i32), ("wt", vp), ("n", i32),
                ("bias", vp), ("bias_rows", i32), ("s", vp), ("ns", i32), ("ws", vp), ("act", i32), ("slope", f32),
                ("c", vp), ("ldc", i64), ("m", i64), ("gate", vp), ("ldg", i64), ("gate_act", i32), ("gate_slope", f32),
                ("accum", i32), ("drop", Dropout)]


class BwdWDesc(ctypes.Structure):
    """desco_bwd_w_desc (include/desco_hip.h)"""
    _fields_ = [("a1", vp), ("lda1", i64), ("k1", i32), ("a2", vp), ("lda2", i64), ("k2", i32), ("dz", vp),
                ("lddz", i64), ("m", i64), ("n", i32), ("dwt", vp), ("dbias", vp)]


class Copy2dDesc(ctypes.Structure):
    """desco_copy2d_desc (include/desco_hip.h)"""
    _fields_ = [("src", vp), ("lds", i64), ("dst", vp), ("ldd", i64), ("rows", i32), ("cols", i32), ("transpose", i32),
                ("accumulate", i32)]


class GossipFoldParams(ctypes.Structure):
    """desco_gossip_fold_params (include/desco_hip.h)"""
    _fields_ = [(n, vp) for n in ("E", "w_pre", "b_pre", "C0", "cb0", "D0", "db0", "C1", "cb1", "D1", "db1")] + \
               [("G0", vp * 2), ("gb0", vp * 2), ("g2", vp * 2), ("gb2", vp * 2)] + \
               [(n, vp) for n in ("P0", "p0", "P3", "P5")] + [("num_q", i32)]


class GossipFoldOut(ctypes.Structure):
    """desco_gossip_fold_out"""
    _fields_ = [(n, vp) for n in ("V0", "V1", "Vp", "wt1", "wtp", "w3t", "w5t", "g0", "g1", "g1c", "a", "h0", "h1")]


class GossipFoldGrads(ctypes.Structure):
    """desco_gossip_fold_grads"""
    _fields_ = [(n, vp) for n in ("dV0", "dV1", "dVp", "dwt1", "dwtp", "dw3t", "dw5t", "dg1",
                                  "dC0", "dcb0", "dD0", "ddb0", "dC1", "dcb1", "dD1", "ddb1")] + \
               [("dG0", vp * 2), ("dgb0", vp * 2), ("dg2", vp * 2), ("dgb2", vp * 2)] + \
               [(n, vp) for n in ("dP0", "dp0", "dP3", "dP5", "scratch")]


# name -> (restype, argtypes); mirrors include/desco_hip.h one to one
SIGNATURES = {
    "desco_abi_version": (c_int, []),
    "desco_device_count": (c_int, []),
    "desco_last_error": (c_char_p, []),
    "desco_partition_build": (c_int, [vp, i64, vp, vp, i32, i32, i32, POINTER(vp)]),
    "desco_partition_sizes": (c_int, [vp, POINTER(i64), POINTER(i64), POINTER(i64), POINTER(i64)]),
    "desco_partition_export": (c_int, [vp, vp, vp, vp, vp, vp, vp]),
    "desco_partition_free": (None, [vp]),
    "desco_partition_degree_sort": (c_int, [vp, i64, vp, vp, vp, vp, vp, vp, vp, i32]),
    "desco_canonical_counts": (c_int, [vp, i64, vp, vp, vp, vp, vp, i32, i32, vp]),
    "desco_canonical_class_table": (c_int, [vp, vp, vp, i32, vp, vp]),
    "desco_canonical_counts_dev": (c_int, [vp, i64, i64, vp, i64, vp, vp, vp, vp, i64, vp, i32, i32, vp, vp]),
    "desco_linear_smallk_f32": (c_int, [vp, i64, i32, vp, vp, vp, i64, i64, i32, vp]),
    "desco_csr_gather_sum_f32": (c_int, [vp, i64, vp, vp, i64, i32, vp, vp]),
    "desco_gemm_f32": (c_int, [vp, i64, i32, vp, i64, i32, vp, i32, vp, i32, vp, i32, vp, i32, f32,
                               vp, i64, i64, vp]),
    "desco_shmp_layer_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp]),
    "desco_linear64_bf16x6_f32": (c_int, [vp, i64, vp, i32, vp, i32, f32, vp, i64, i64, vp]),
    "desco_shmp_layer_bf16x6_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp]),
    "desco_shmp_layer_pool_bf16x6_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp]),
    "desco_pool_reduce_f32": (c_int, [vp, vp, vp, vp, i64, vp, i64, vp, i64, i32, vp]),
    "desco_pool_post_bf16x6_f32": (c_int, [vp, i64, i32, vp, i32, vp, i32, f32, vp, i64, i64, vp, vp, vp, POINTER(vp), vp, i32, vp]),
    "desco_post_mp_tail_f16x3_f32": (c_int, [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "desco_count_head_emb_f16x3_f32": (c_int, [vp, i64, i64, vp, vp, vp, i64, i32, vp, f32, vp, f32, i32, vp, i64, i32, vp]),
    "desco_gemm_bf16x6_desc_f32": (c_int, [POINTER(GemmDesc), vp, i32, vp]),
    "desco_split_bf16x3_t_f32": (c_int, [vp, i32, i32, i64, vp, vp]),
    "desco_gemm_bf16x6_multi_f32": (c_int, [i32, POINTER(GemmDesc), POINTER(vp), vp]),
    "desco_split_bf16x3_batch_f32": (c_int, [vp, i64, i32, i32, i32, i32, vp, vp]),
    "desco_gemm_bf16_multi_f32": (c_int, [i32, POINTER(GemmDesc), POINTER(vp), vp]),
    "desco_pool_reduce_multi_f32": (c_int, [i32, POINTER(vp), vp, vp, vp, i64, POINTER(vp), i64, POINTER(vp), i64, i32, vp]),
    "desco_shmp_layer_f16x3_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, i64, i64, vp, i64, vp, i64, vp, vp, i64, vp]),
    "desco_shmp_layer_pool_f16x3_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp]),
    "desco_shmp_layer_pool_table_f16x3_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, vp, vp]),
    "desco_shmp_pool_tile_rows": (c_int, []),
    "desco_degree_affine_f32": (c_int, [vp, i64, i64, i32, vp, i32, f32, vp, i64, vp, i64, vp, vp]),
    "desco_degree_affine_pool_f32": (c_int, [vp, i64, i32, vp, i32, f32, vp, i64, vp, vp, vp, vp]),
    "desco_gemm_bf16x6_f32": (c_int, [vp, i64, i32, vp, i64, i32, vp, i32, vp, i32, vp, i32, vp, i32, f32,
                                      vp, i64, i64, vp]),
    "desco_gemm_bf16_f32": (c_int, [vp, i64, i32, vp, i64, i32, vp, i32, vp, i32, vp, i32, vp, i32, f32,
                                      vp, i64, i64, vp]),
    "desco_round_bf16_f32": (c_int, [vp, i64, vp, vp]),
    "desco_gemm_f16x3_f32": (c_int, [vp, i64, i32, vp, i64, i32, vp, vp, i32, vp, i32, vp, i32, vp, i32, f32,
                                     vp, i64, i64, vp, vp]),
    "desco_row_absmax_f32": (c_int, [vp, i64, i32, vp, i64, i32, i64, vp, vp]),
    "desco_split_f16x2_f32": (c_int, [vp, i64, vp, vp, vp]),
    "desco_segment_sum_f32": (c_int, [vp, i64, i32, vp, i64, vp, i64, vp, i64, vp]),
    "desco_segment_sum_layers_f32": (c_int, [vp, i64, i64, i32, vp, i64, vp, i64, vp, i64, vp]),
    "desco_count_head_f32": (c_int, [vp, i64, vp, i64, i32, vp, f32, vp, f32, i32, vp, i64, i64, i32, vp]),
    "desco_scatter_rows_f32": (c_int, [vp, i64, vp, i64, i32, vp, i64, vp]),
    "desco_gossip_layer0_f32": (c_int, [vp, i64, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "desco_gossip_gather_f32": (c_int, [vp, vp, vp, i64, i32, vp, vp, vp]),
    "desco_gossip_scalars_f32": (c_int, [vp, i64, vp, vp, i64, i32, vp, vp, vp, vp]),
    "desco_split_bf16x3_f32": (c_int, [vp, i64, vp, vp]),
    "desco_partition_dev_count": (c_int, [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp]),
    "desco_partition_dev_scan": (c_int, [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]),
    "desco_partition_dev_fill": (c_int, [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp,
                                         i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]),
    "desco_vcsr_transpose_sym": (c_int, [vp, vp, i64, i32, i64, vp, vp, vp]),
    "desco_segment_ids": (c_int, [vp, i64, vp, vp]),
    "desco_gossip_fused_f32": (c_int, [vp, vp, vp, i64, i32] + [vp] * 16 + [f32, vp, vp, vp]),
    "desco_gossip_tile_order": (c_int, [vp, i64, vp, vp]),
    "desco_gossip_f16_stream": (c_int, [vp, vp, vp, vp, vp, vp]),
    "desco_gossip_fused_f16x3_f32": (c_int, [vp, vp, vp, i64, i32] + [vp] * 14 + [f32, vp, vp, vp, vp]),
    "desco_csr_gather_sum_add_f32": (c_int, [vp, i64, vp, vp, i64, vp, i64, vp, i64, vp]),
    "desco_shmp_bwd_dx_f32": (c_int, [vp, i64, vp, vp, i64, i64, i32, i32, vp, i64, vp, vp, i64, vp, f32, vp, vp]),
    "desco_add_rows_f32": (c_int, [vp, i64, vp, i64, i64, i32, vp]),
    "desco_gemm_tn_workspace": (ctypes.c_size_t, [i64, i32, i32, POINTER(c_int)]),
    "desco_gemm_tn_f32": (c_int, [vp, i64, vp, i64, i64, i32, i32, vp, i64, i32, vp, vp]),
    "desco_linear_bwd_w_workspace": (ctypes.c_size_t, [i64, i32, i32]),
    "desco_linear_bwd_w_f32": (c_int, [vp, i64, i32, vp, i64, i32, vp, i64, i64, i32, vp, i64, vp, vp, vp]),
    "desco_colsum_f32": (c_int, [vp, i64, i64, i32, vp, i32, vp, vp]),
    "desco_gemm_f32_multi": (c_int, [i32, POINTER(GemmDesc), vp]),
    "desco_linear_bwd_w_multi_workspace": (ctypes.c_size_t, [i32, POINTER(BwdWDesc)]),
    "desco_linear_bwd_w_multi_f32": (c_int, [i32, POINTER(BwdWDesc), vp, vp]),
    "desco_shmp_trunk_small_max_rows": (c_int, []),
    "desco_shmp_trunk_small_fwd_f32": (c_int, [vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i64, vp]),
    "desco_shmp_trunk_small_bwd_f32": (c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i64, vp, vp, vp, vp]),
    "desco_shmp_trunk_graphs_max_rows": (c_int, []),
    "desco_shmp_trunk_graphs_fwd_f32": (c_int, [vp, vp, vp, i64, i32, vp, vp, vp, i32, POINTER(Dropout), vp, vp, i64, vp]),
    "desco_shmp_trunk_graphs_bwd_f32": (c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp, vp, i64, f32, vp, vp, vp, vp, vp]),
    "desco_linear_smallk_bwd_f32": (c_int, [vp, i64, i32, vp, i64, i64, vp, vp, vp]),
    "desco_rowdot_bwd_f32": (c_int, [vp, i64, i32, vp, vp, i64, vp, i64, vp, vp, vp]),
    "desco_adam_step_f32": (c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64, f64, vp]),
    "desco_act_grad_f32": (c_int, [vp, vp, i32, f32, vp, i64, vp]),
    "desco_count_head_bwd_workspace": (ctypes.c_size_t, [i64, i32, i32]),
    "desco_count_head_bwd_f32": (c_int, [vp, i64, vp, i64, i32, vp, f32, vp, i64, i64, i32, vp, i64,
                                         vp, vp, vp, vp]),
    "desco_affine_rows_f32": (c_int, [vp, vp, i32, vp, i32, i32, f32, vp, i64, vp]),
    "desco_affine_rows_bwd_f32": (c_int, [vp, i32, vp, i32, i64, vp, vp, vp]),
    "desco_rng_next": (c_int, [vp, vp, vp]),
    "desco_dropout_mask_f32": (c_int, [POINTER(Dropout), i64, i32, vp, i64, vp]),
    "desco_affine_rows_dropout_f32": (c_int, [vp, vp, i32, vp, i32, i32, f32, POINTER(Dropout), vp, i64, vp]),
    "desco_act_grad_dropout_f32": (c_int, [vp, vp, i32, f32, POINTER(Dropout), vp, i64, i32, vp]),
    "desco_rowdot2_f32": (c_int, [vp, vp, i32, vp, i64, vp]),
    "desco_rowdot_add_f32": (c_int, [vp, i64, i32, vp, f32, vp, vp, i64, vp]),
    "desco_copy2d_multi_f32": (c_int, [i32, POINTER(Copy2dDesc), vp]),
    "desco_fold_shmp_fwd_f32": (c_int, [vp, i32, i32, i32, vp, vp, vp]),
    "desco_fold_shmp_bwd_f32": (c_int, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "desco_loss_f32": (c_int, [vp, vp, i64, i32, vp, vp, vp, vp]),
    "desco_affine_scalar_f32": (c_int, [vp, vp, vp, vp, vp, i64, vp]),
    "desco_gossip_fold_fwd_f32": (c_int, [POINTER(GossipFoldParams), POINTER(GossipFoldOut), vp]),
    "desco_gossip_fold_bwd_f32": (c_int, [POINTER(GossipFoldParams), POINTER(GossipFoldOut), POINTER(GossipFoldGrads), vp]),
    "desco_fill_f32": (c_int, [vp, f32, i64, vp]),
}


class DescoLibraryError(ImportError):
    pass


def lib():
    """Load (once) and return the shared library; raises DescoLibraryError when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DescoLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. desco_amd has no CPU or "
                "PyTorch fallback; run `make -C desco_amd/csrc` (or __graft_entry__.build()).")
        # PyTorch's wheel bundles its own HIP / HSA runtime.  It must be in the process BEFORE this
        # library is opened, so that the library's libamdhip64 dependency resolves (by SONAME) to the
        # runtime torch's tensors live in; opened first, it would pull in /opt/rocm's copy and the
        # process would hold two runtimes ("no ROCm-capable device" from the second one).
        import torch  # noqa: F401
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise DescoLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        ver = handle.desco_abi_version()
        if ver == ABI_VERSION + 1000 and os.environ.get("DESCO_ALLOW_DEBUG_LIB") == "1":
            pass        # a timing-only ablation / probe build (wrong results by design): developer tools only
        elif ver != ABI_VERSION:
            raise DescoLibraryError(
                f"{LIB_PATH}: ABI version {ver}, expected {ABI_VERSION}" +
                (" (a debug / ablation build; set DESCO_ALLOW_DEBUG_LIB=1 for tools/debug runs)" if ver >= 1000 else "; rebuild it"))
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().desco_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdesco_hip {what} failed (code {rc}): {msg}")
