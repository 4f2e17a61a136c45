"""ctypes binding of libglass_hip.so (the C ABI declared in include/glass_hip.h).

The library is built in-tree by `make -C glass_amd/csrc` (or `__graft_entry__.build()`); the
`.so` sits next to this file so it travels to the GPU box with the repo snapshot.  There is no
fallback: if the library is missing, or a call fails, a GlassHipError is raised.
"""
import ctypes
import os

# torch MUST be imported before libglass_hip.so is opened: torch bundles its own HIP runtime with
# the same SONAME (libamdhip64.so.7) as /opt/rocm's.  Loaded first, it is the one runtime both torch
# and this library use (one set of streams and device pointers); loaded second, the process would
# hold two runtimes and every launch here would fail with "no ROCm-capable device".
import torch  # noqa: F401
from ctypes import c_void_p, c_int, c_int64, c_uint64, c_float, c_double, c_char_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLASS_HIP_LIB: another build of the same library (laboratory A/B of compile-time variants); the product default is the
# in-tree build next to this file
LIB_PATH = os.environ.get("GLASS_HIP_LIB") or os.path.join(_HERE, "libglass_hip.so")

POOL_MODES = {"sum": 0, "mean": 1, "max": 2, "size": 3}
AGGR_MODES = {"mean": 0, "sum": 1, "gcn": 2}
ACT_NONE, ACT_ELU, ACT_RELU = 0, 1, 2
PLAN_HEADER_WORDS = 16
EMBED_NORM_MAX_ROWS = 8192  # GLASS_EMBED_NORM_MAX_ROWS
ABI_VERSION = 6


class GlassHipError(RuntimeError):
    pass


_P = c_void_p
_I = c_int64


class GnBwdSrc(ctypes.Structure):
    """glass_gn_bwd_src (include/glass_hip.h): the comb pair's gradient operand derived on load from a GraphNorm's output
    gradient (its backward apply fused into glass_comb_eff_bwd_f32)."""
    _fields_ = [("acc", c_void_p), ("n_rep", c_int64), ("dy", c_void_p), ("lddy", c_int64), ("x", c_void_p), ("ldx", c_int64),
                ("addend", c_void_p), ("ldadd", c_int64), ("saved", c_void_p), ("gamma", c_void_p), ("alpha", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("dalpha", c_void_p), ("accumulate", c_int), ("act", c_int),
                ("p_drop", c_float), ("call_id", c_uint64)]

    @property
    def ptr(self):
        return ctypes.addressof(self)


class GnSrc(ctypes.Structure):
    """glass_gn_src (include/glass_hip.h): a GraphNorm whose forward sums are still in exact accumulators."""
    _fields_ = [("acc", c_void_p), ("n_src", c_int64), ("n_rep", c_int64), ("gamma", c_void_p), ("beta", c_void_p),
                ("alpha", c_void_p), ("eps", c_float)]

    @classmethod
    def of(cls, acc, n_src, n_rep, gn):
        """acc: int64 tensor (n_src consecutive accumulator blocks, n_rep replicas used); gn: the GraphNorm module."""
        return cls(acc.data_ptr(), n_src, n_rep, gn.weight.data_ptr(), gn.bias.data_ptr(), gn.mean_scale.data_ptr(),
                   float(gn.eps))

    @property
    def ptr(self):
        return ctypes.addressof(self)
# name -> (restype, argtypes); mirrors include/glass_hip.h one to one
SIGNATURES = {
    "glass_version": (c_int, []),
    "glass_last_error_string": (c_char_p, []),
    "glass_spmm_plan_build": (c_int, [_P, _I, _P, POINTER(c_int64)]),
    "glass_spmm_ws_bytes": (c_int64, [_P, _I]),
    "glass_spmm_csr_f32": (c_int, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P]),
    "glass_adj_values_f32": (c_int, [_P, _P, _P, _I, c_int, _P, _P, _P]),
    "glass_maxzoz_i64": (c_int, [_P, _I, _P, _I, _P]),
    "glass_embed_label_f32": (c_int, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P]),
    "glass_mix_fwd_f32": (c_int, [_P, _I, _P, c_double, c_int, _P, _I, _I, _I, _P]),
    "glass_mix_bwd_f32": (c_int, [_P, _I, _P, _I, _P, c_double, c_int, _P, _I, _I, _I, _P]),
    "glass_graphnorm_ws_bytes": (c_int64, [_I, _I]),
    "glass_graphnorm_fwd_f32": (c_int, [_P, _I, _P, _I, _I, _I, _P, _P, _P, c_float, _P, c_int, c_float, _P, c_uint64,
                                        _P, _P]),
    "glass_graphnorm_bwd_f32": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float,
                                        _P, c_uint64, _P, _P]),
    "glass_rng_advance": (c_int, [_P, _P]),
    "glass_empty_launch": (c_int, [_I, _I, _I, _I, _I, _P]),
    "glass_dropout_scales_f32": (c_int, [_P, c_uint64, c_float, _I, _I, _P, _P]),
    "glass_dual_linear_stat_rows": (c_int64, [_I]),
    "glass_copy_pair": (c_int, [_P, _P, _I, _P, _P, _I, _P]),
    "glass_graphnorm_stats_f32": (c_int, [_P, _I, _I, _I, _P, _P, _P, c_float, _P, _P, _P]),
    "glass_readout_supported": (c_int, [_I, _I, c_int]),
    "glass_readout_ws_bytes": (c_int64, [_I, _I, _I]),
    "glass_readout_train_f32": (c_int, [_P, _I, _P, _P, _P, _P, _I, _I, c_int, _P, _P, _P, c_int, _I, _P, _P, _P, _P, _P, _I,
                                        _P, _P, c_int, _P, _P, _P, c_int, _P, _I, _I, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "glass_readout_scatter_ws_bytes": (c_int64, [_I, _I, _I]),
    "glass_linear_wgrad_reduce_batch_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "glass_wgrad_reduce_spmm_f32": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P,
                                            _P, _P]),
    "glass_spmm_reduce_rows_f32": (c_int, [_P, _P, _I, _I, _P, _I, _P]),
    "glass_segment_pool_f32": (c_int, [_P, _I, _P, _I, _I, c_int, _P, _I, _P, _I, _I, _P]),
    "glass_segment_pool_bwd_f32": (c_int, [_P, _I, _P, _I, _I, c_int, _P, _P, _I, _I, _I, _P]),
    "glass_segment_pool_bwd_atomic_f32": (c_int, [_P, _I, _P, _I, _I, c_int, _P, _P, _I, _I, _I, _P]),
    "glass_segment_pool_max_bwd_exact_f32": (c_int, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P, _P]),
    "glass_pair_pool_ws_bytes": (c_int64, [_I, _I]),
    "glass_segment_pool_bwd_exact_ws_bytes": (c_int64, [_I, _I, _I]),
    "glass_segment_pool_bwd_exact_f32": (c_int, [_P, _I, _P, _I, _I, c_int, _P, _I, _I, _I, _P, _P]),
    "glass_pair_pool_f32": (c_int, [_P, _I, _P, _I, c_int, _P, _I, _I, _I, _P]),
    "glass_pair_pool_bwd_f32": (c_int, [_P, _I, _P, _I, c_int, _P, _I, _I, _I, _P, _P]),
    "glass_dense_caps_query": (c_int, [_I, _P]),
    "glass_pair_head_supported": (c_int, [_I]),
    "glass_pair_head_ws_bytes": (c_int64, [_I, _I]),
    "glass_pair_head_fwd_f32": (c_int, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, c_float, _P, c_uint64, _P, _P, _P, _P, _P, _P]),
    "glass_pair_head_bwd_f32": (c_int, [_P, _I, _I, _P, _I, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, _P, _P, _I, _P, _P]),
    "glass_linear_wgrad_ws_bytes": (c_int64, [_I, _I, _I]),
    "glass_linear_wgrad_f32": (c_int, [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, c_int, _P, _P]),
    "glass_dual_linear_supported": (c_int, [_I]),
    "glass_dual_linear_layout": (c_int, [_I]),
    "glass_dual_linear_fwd_f32": (c_int, [_P, _I, _P, _I, _P, _P, _P, c_double, c_int, _P, _I, _P, _I, _I, _I, _P, c_int, _P, _P,
                                          c_int, c_float, _P, c_uint64, _P, _I, _P, _I, _P]),
    "glass_dual_linear_fwd_gather_supported": (c_int, [_I]),
    "glass_step_prologue_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, c_float, _P, _P, _I, _P, _I, _P]),
    "glass_peer_allreduce_adam_f32": (c_int, [_P, _I, _P, _P, _P, _P, c_double, c_double, c_double, c_double, _P, _P, _P, _I, _P, _P]),
    "glass_peer_alloc": (c_int, [_I, _P]),
    "glass_peer_free": (c_int, [_P]),
    "glass_peer_export": (c_int, [_P, _P]),
    "glass_peer_import": (c_int, [_P, _P]),
    "glass_peer_close": (c_int, [_P]),
    "glass_step_head_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, c_float, _P, _P, _I, _P, _I,
                                    _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "glass_graphnorm_finalize_f32": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P, c_float, _P, _P]),
    "glass_graphnorm_apply_f32": (c_int, [_P, _I, _P, _I, _I, _I, _P, c_int, c_float, _P, c_uint64, _P]),
    "glass_dual_linear_dgrad_f32": (c_int, [_P, _I, _P, _I, _P, c_double, c_int, _P, _I, _P, _I, c_float, _P, c_uint64, _P, _I,
                                            _I, _I, _P, _P, _I, _P, _P, c_int, c_float, c_uint64, c_int, _P]),
    "glass_dual_linear_bwd_f32": (c_int, [_P, _I, _P, _I, _P, c_double, c_int, _P, _I, _P, _I, c_float, _P, c_uint64, _P, _I,
                                          _I, _I, _P, _P, _I, _P, _P, c_int, c_float, c_uint64, c_int, _P, _I, _P, _I, _P, _P]),
    "glass_gn_exact_supported": (c_int, [_I]),
    "glass_gn_exact_fwd_supported": (c_int, [_I]),
    "glass_gn_exact_words": (c_int64, [_I]),
    "glass_graphnorm_stats_exact_f32": (c_int, [_P, _I, _I, _I, _P, c_int, _P]),
    "glass_graphnorm_bwd_from_stats_f32": (c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P,
                                                   c_int, c_int, c_float, _P, c_uint64, _P, _P]),
    "glass_embed_norm_fwd_f32": (c_int, [_P, _P, _I, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _I, c_float, _P, c_uint64, _P,
                                         _I, _P, _I, _I, _P]),
    "glass_embed_norm_bwd_f32": (c_int, [_P, _P, _I, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, _I, _P]),
    "glass_embed_norm_bwd_adam_f32": (c_int, [_P, _P, _I, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, _I, _P, _P, _I, _P, _P, _P,
                                              _P, _I, _P, c_double, c_double, c_double, c_double, _P, _I, _I, _I, _I, _P]),
    "glass_dual_linear_wgrad_f32": (c_int, [_P, _I, _P, _I, _P, c_double, c_int, _P, _I, _P, _I, _I, _I, _P, _I, _P,
                                            c_int, _P, _P]),
    "glass_dense_pack_batch_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "glass_dense_image_floats": (_I, [_I, _I, c_int]),
    "glass_dual_linear_dgrad_layout": (c_int, [_I, _I]),
    "glass_dual_linear_fwd_layout": (c_int, [_I, _I]),
    "glass_head_loss_fwd_f32": (c_int, [_P, _I, _P, _P, _P, c_int, _I, _I, _I, _P, _P, _P, _P]),
    "glass_head_linear_f32": (c_int, [_P, _I, _P, _P, _I, _I, _I, _P, _I, _P]),
    "glass_head_loss_bwd_f32": (c_int, [_P, _I, _P, _P, _P, c_int, _P, _I, _I, _I, _P, _I, _P, _P, c_int, _P]),
    "glass_batch_labels_ws_bytes": (c_int64, [_I]),
    "glass_batch_labels": (c_int, [_P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, c_int, _P]),
    "glass_batch_labels_gather": (c_int, [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, c_int, _P]),
    "glass_comb_eff_supported": (c_int, [_I]),
    "glass_comb_eff_blocks": (c_int64, [_I, _I, _I]),
    "glass_comb_eff_fwd_blocks": (c_int64, [_I, _I, _I]),
    "glass_comb_eff_ws_bytes": (c_int64, [_I, _I, _I]),
    "glass_comb_eff_fwd_layout": (c_int, [_I]),
    "glass_comb_eff_fwd_supported": (c_int, [_I]),
    "glass_comb_eff_max_rows": (c_int64, [_I]),
    "glass_comb_eff_dgrad_layout2": (c_int, [_I]),
    "glass_comb_eff_fwd_f32": (c_int, [_P, _I, _P, _I, _P, _P, _P, c_double, _P, _I, _I, _I, _P, c_int, _P, _P, c_int, c_float, _P,
                                       c_uint64, _P, _I, _P, _P, _I, _P]),
    "glass_comb_eff_bwd_f32": (c_int, [_P, _I, _P, c_double, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, c_int, c_float, _P,
                                       c_uint64, c_int, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P]),
    "glass_comb_eff_bwd_gn_src_supported": (c_int, [_I, _I]),
    "glass_adam_step_f32": (c_int, [_P, _P, _P, _P, _I, _P, c_double, c_double, c_double, c_double, _P, _P]),
}

_lib = None


class DenseCaps(ctypes.Structure):
    """include/glass_hip.h: glass_dense_caps — one capability record per hidden width."""
    _fields_ = [(n, ctypes.c_int32) for n in ("family", "weight_layout", "fwd_layout_trans", "fwd_layout_comb", "dgrad_layout_trans",
                                              "dgrad_layout_comb", "stat_rows", "fwd_gather", "gn_exact", "gn_exact_fwd", "comb_eff",
                                              "comb_eff_fwd", "comb_eff_fwd_layout", "comb_eff_dgrad_layout2", "pair_head", "act_codes",
                                              "product_form", "serve_width")]


_caps = {}


def dense_caps(H):
    """The library's capability record for hidden width H (cached)."""
    c = _caps.get(int(H))
    if c is None:
        c = DenseCaps()
        check(load().glass_dense_caps_query(int(H), ctypes.addressof(c)), "glass_dense_caps_query")
        _caps[int(H)] = c
    return c


# options of a dense call: bits above the activation code in the `act` word of glass_dual_linear_{fwd,bwd,dgrad,wgrad}_f32
ACT_MASK = 0xff
DENSE_F32_PRODUCTS = 0x100


def load():
    """Load the shared library once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlassHipError(f"{LIB_PATH} not found: build it with `make -C glass_amd/csrc` "
                            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.glass_version()
    if v != ABI_VERSION:
        raise GlassHipError(f"libglass_hip ABI version {v}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().glass_last_error_string().decode(errors="replace")
        raise GlassHipError(f"{what} failed (rc={rc}): {msg}")
