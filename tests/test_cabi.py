"""CPU-side checks of the C ABI: the library loads, exports every symbol include/glass_hip.h
declares, the ctypes table mirrors the header, the host-side plan builder is correct, and argument
validation returns error codes (no exceptions across the ABI).  No GPU compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "glass_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glass_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from glass_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in glass_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.glass_version() == _lib.ABI_VERSION == 6


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "glass_hip.h")).read()
    for cite in ("impl/models.py:164", "impl/models.py:83-111", "impl/utils.py:32-45", "impl/models.py:346-350"):
        assert cite in text


def _plan(rowptr):
    from glass_amd import _lib
    lib = _lib.load()
    rp = np.asarray(rowptr, dtype=np.int32)
    n = rp.shape[0] - 1
    words = ctypes.c_int64(0)
    assert lib.glass_spmm_plan_build(rp.ctypes.data, n, None, ctypes.byref(words)) == 0
    plan = np.zeros(words.value, dtype=np.int32)
    assert lib.glass_spmm_plan_build(rp.ctypes.data, n, plan.ctypes.data, ctypes.byref(words)) == 0
    return plan


def _check_plan(rowptr):
    rp = np.asarray(rowptr, dtype=np.int64)
    n = rp.shape[0] - 1
    plan = _plan(rowptr)
    hdr = plan[:16]
    assert hdr[0] == 0x474C5350 and hdr[2] == n and hdr[3] == rp[-1]
    n_sweep, n_long, n_red, n_slots, thr, chunk = hdr[4:10]
    sweep = plan[hdr[10]:hdr[10] + 4 * n_sweep].reshape(-1, 4)  # items (r0, r1, e0, e1)
    deg = rp[1:] - rp[:-1]
    longs = plan[hdr[11]:hdr[11] + 4 * n_long].reshape(-1, 4)
    reds = plan[hdr[12]:hdr[12] + 3 * n_red].reshape(-1, 3)
    # sweep items: consecutive runs of short rows, in order, covering every short row exactly once, each within the
    # kernel's staging caps (64 rows, 256 edges), carrying their own edge range
    all_long = 0 < n <= 1024 and rp[-1] >= 64 * n  # few rows, many entries: no sweep, every row is workgroup items
    assert (thr == 0 and n_sweep == 0) if all_long else thr > 0
    covered = []
    for r0, r1, e0, e1 in sweep.tolist():
        assert 0 <= r0 < r1 <= n and r1 - r0 <= 64 and e0 == rp[r0] and e1 == rp[r1] and e1 - e0 <= 256
        covered += list(range(r0, r1))
    assert covered == np.nonzero(deg < thr)[0].tolist()
    # header word 14: the smallest mean degree (rounded up) of any sweep item — the launch picks the kernel without the
    # flat-mode branch when no item can qualify at its feature width
    # header word 15: seven nibbles = fifteenths of the sweep cost (edges + 4 per row) lying in items that qualify for flat
    # mode at G = 1, 2, 4, .. 64 lane groups (edges <= 4 * G * rows) — the launch picks the kernel from it
    if n_sweep:
        items = sweep.tolist()
        assert hdr[14] == min(-(-(e1 - e0) // (r1 - r0)) for r0, r1, e0, e1 in items)
        cost = lambda it: (it[3] - it[2]) + 4 * (it[1] - it[0])
        total = sum(cost(it) for it in items)
        for k in range(7):
            flat = sum(cost(it) for it in items if it[3] - it[2] <= 4 * (1 << k) * (it[1] - it[0]))
            assert (int(hdr[15]) >> (4 * k)) & 15 == 15 * flat // total
    # every long row (deg >= thr) is covered exactly by its chunks, in order, whole 64-edge batches
    long_rows = np.nonzero(deg >= thr)[0]
    assert sorted(set(longs[:, 0].tolist())) == long_rows.tolist()
    slots = []
    for r in long_rows:
        it = longs[longs[:, 0] == r]
        assert it[0, 1] == rp[r] and it[-1, 2] == rp[r + 1]
        assert np.all(it[1:, 1] == it[:-1, 2])
        assert np.all((it[:, 2] - it[:, 1]) <= chunk) and np.all((it[:-1, 2] - it[:-1, 1]) % 64 == 0)
        if len(it) == 1:
            assert it[0, 3] == -1
        else:
            slots += it[:, 3].tolist()
            (rr, ) = reds[reds[:, 0] == r]
            assert rr[1] == it[0, 3] and rr[2] == len(it)
    assert slots == list(range(n_slots))
    return hdr


def test_plan_builder_shapes():
    rng = np.random.default_rng(0)
    _check_plan([0])  # empty matrix
    _check_plan([0, 0, 0, 0])  # only empty rows
    _check_plan(np.concatenate([[0], np.cumsum(rng.integers(0, 60, 5000))]))
    hdr = _check_plan(np.concatenate([[0], np.cumsum(rng.choice([0, 3, 255, 256, 257, 2048, 2049, 50000], 300))]))
    assert hdr[5] > 0 and hdr[6] > 0 and hdr[7] > 0
    hdr = _check_plan(np.concatenate([[0], np.cumsum(np.full(20000, 37))]))
    assert hdr[5] == 0 and hdr[4] == 20000  # ppi_bp-like: one row per wave
    hdr = _check_plan(np.concatenate([[0], np.cumsum(np.full(5000, 12))]))
    assert hdr[4] == 5000  # density-like: the budget floor (16 cost units) still gives every degree-12 row its own wave
    hdr = _check_plan(np.concatenate([[0], np.cumsum([10] * 3000 + [100, 300, 63])]))
    assert hdr[8] == 64 and hdr[9] == 512 and hdr[5] == 2  # latency-bound product (edges + 4 rows <= 2 Mi): rows of >= 64 edges go to workgroups
    hdr = _check_plan(np.concatenate([[0], np.cumsum([40] * 60000 + [100, 300, 63])]))
    assert hdr[8] == 256 and hdr[9] == 2048 and hdr[5] == 1  # throughput-bound: long rows start at 256 edges, 2048-edge chunks
    hdr = _check_plan(np.concatenate([[0], np.cumsum(rng.choice([0, 0, 40, 300, 5000], 60))]))  # selection-matrix-like
    assert hdr[4] == 0 and hdr[5] >= 60 and hdr[8] == 0


def test_plan_builder_rejects_bad_rowptr():
    from glass_amd import _lib
    lib = _lib.load()
    words = ctypes.c_int64(0)
    bad = np.array([0, 5, 3], dtype=np.int32)
    assert lib.glass_spmm_plan_build(bad.ctypes.data, 2, None, ctypes.byref(words)) == -1
    assert b"monotone" in lib.glass_last_error_string()
    assert lib.glass_spmm_plan_build(None, 2, None, ctypes.byref(words)) == -1


def test_argument_validation_returns_codes_not_exceptions():
    """Validation runs before any HIP call, so these work without a GPU."""
    from glass_amd import _lib
    lib = _lib.load()
    assert lib.glass_spmm_csr_f32(None, None, None, None, 0, None, 0, 4, 8, None, None, None, None) == -1
    plan = _plan([0, 1, 2])
    x = np.zeros(16, dtype=np.float32)
    # plan built for 2 rows used with n_rows=3 -> GLASS_E_PLAN
    rc = lib.glass_spmm_csr_f32(plan.ctypes.data, plan.ctypes.data, x.ctypes.data, x.ctypes.data, 8, x.ctypes.data, 8,
                                3, 8, plan.ctypes.data, plan.ctypes.data, None, None)
    assert rc == -2 and b"plan does not match" in lib.glass_last_error_string()
    assert lib.glass_adj_values_f32(plan.ctypes.data, None, None, 2, 7, x.ctypes.data, None, None) == -3  # bad aggr
    assert lib.glass_segment_pool_f32(x.ctypes.data, 4, x.ctypes.data, 1, 1, 9, x.ctypes.data, 4, None, 4, 4,
                                      None) == -3  # bad pool mode
    assert lib.glass_graphnorm_fwd_f32(None, 0, None, 0, 0, 0, None, None, None, 1e-5, None, 0, 0.0, None, 0, None,
                                       None) == -1
    assert lib.glass_linear_wgrad_f32(x.ctypes.data, 6, x.ctypes.data, 4, 2, 6, 4, x.ctypes.data, 4, None, 0,
                                      x.ctypes.data, None) == -3  # O % 4 != 0 -> caller uses a library GEMM
    assert lib.glass_mix_fwd_f32(x.ctypes.data, 4, x.ctypes.data, 0.5, 0, x.ctypes.data, 4, 2, 4, None) == -1  # ldt<2H
    assert lib.glass_spmm_ws_bytes(x.ctypes.data, 64) == -2  # not a plan header


def test_step_program_entry_points_validate_on_the_host():
    """The entry points of the step program (table embedding, GraphNorm in pieces, fused readout, deferred weight
    gradient reduction, batch copy): host-side answers and argument checks, no GPU."""
    from glass_amd import _lib
    lib = _lib.load()
    x = np.zeros(64, dtype=np.float32)
    p = x.ctypes.data
    # readout support matrix: C multiple of 4 and <= 1024, K <= 256, pooling sum(0) | mean(1) | size(3)
    assert lib.glass_readout_supported(128, 6, 0) == 1 and lib.glass_readout_supported(128, 6, 3) == 1
    assert lib.glass_readout_supported(128, 6, 2) == 0 and lib.glass_readout_supported(130, 6, 0) == 1  # any C (scalar form)
    assert lib.glass_readout_supported(2048, 6, 0) == 0 and lib.glass_readout_supported(128, 300, 0) == 0
    assert lib.glass_readout_ws_bytes(80, 128, 6) >= 8 * 2 * 80 * 128 + 4 * (4 * 128 + 80 * 128 + 80 * 6 + 80)
    assert lib.glass_readout_ws_bytes(0, 128, 6) == -1
    assert lib.glass_readout_scatter_ws_bytes(1000, 80, 10) == 0 and lib.glass_readout_scatter_ws_bytes(50000, 200, 155) > 4 * 2 * 200 * 155
    # node-bucketed exact backward of the pools (bucket.h): offsets + rank + list (+ subgraph scales unless node pairs)
    assert lib.glass_pair_pool_ws_bytes(17080, 131072) == 4 * (17084 + 4 * 131072)
    assert lib.glass_segment_pool_bwd_exact_ws_bytes(3000, 500, 37) == 4 * (3004 + 2 * 500 * 37 + 500)
    assert lib.glass_segment_pool_bwd_exact_ws_bytes(3000, 500, 2) == lib.glass_pair_pool_ws_bytes(3000, 500)
    assert lib.glass_pair_pool_ws_bytes(0, 5) == 0
    # exact GraphNorm accumulators: everything at hidden 64, the forward sums alone also at hidden 128
    assert lib.glass_gn_exact_supported(64) == 1 and lib.glass_gn_exact_supported(128) == 0
    assert lib.glass_gn_exact_fwd_supported(64) == 1 and lib.glass_gn_exact_fwd_supported(128) == 1 and lib.glass_gn_exact_fwd_supported(256) == 0
    # layer 0's trans kernel gathers from the embedding table at every dense width
    assert all(lib.glass_dual_linear_fwd_gather_supported(h) == 1 for h in (8, 17, 64, 128, 256, 512))
    args = [p, 128, p, p, p, p, 80, 10, 2, p, p, p, 0, 6, p, p, p, p, p, 128, p, p, 1, p, p, p, 1, p, 1000, 128, None, None, None, None,
            None, 0, None, None, None]
    assert lib.glass_readout_train_f32(*args) == -3  # max pooling is not fusable
    # table path: more rows than GLASS_EMBED_NORM_MAX_ROWS
    assert lib.glass_embed_norm_fwd_f32(p, p, 10000, p, p, p, p, 1e-5, p, p, None, None, 0, 0.0, None, 1, p, 64, p, 10, 64,
                                        None) == -1
    assert b"8192" in lib.glass_last_error_string()
    assert lib.glass_embed_norm_bwd_f32(None, p, 4, p, p, p, p, p, 1, p, p, p, 1, 64, None) == -1
    # GraphNorm pieces
    ptrs = np.array([p] * 9, dtype=np.uint64)
    assert lib.glass_graphnorm_finalize_f32(ptrs.ctypes.data, 9, 4, 64, 100, p, p, p, 1e-5, p, None) == -1  # > 8 sources
    assert lib.glass_graphnorm_apply_f32(p, 2, p, 4, 4, 4, p, 0, 0.0, None, 0, None) == -1                # ldx < C
    assert lib.glass_graphnorm_apply_f32(p, 4, p, 4, 4, 4, p, 0, 0.5, None, 0, None) == -1                # dropout w/o state
    assert lib.glass_graphnorm_stats_f32(None, 4, 4, 4, p, p, p, 1e-5, p, p, None) == -1
    assert lib.glass_graphnorm_bwd_from_stats_f32(p, 4, p, 4, p, 4, None, 0, 4, 4, p, p, p, None, 3, p, p, p, 1, 0, 0.0, None,
                                                  0, p, None) == -1
    # deferred reduction and batch copy
    assert lib.glass_linear_wgrad_reduce_batch_f32(0, None, None, None, None, None, None, None, None, None, None) == 0
    assert lib.glass_linear_wgrad_reduce_batch_f32(1, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.glass_copy_pair(p, p, 6, p, p, 4, None) == -1 and lib.glass_copy_pair(None, p, 4, p, p, 4, None) == -1
    # fused dense forward: GraphNorm prologue without a side output is rejected before any launch
    assert lib.glass_dual_linear_fwd_f32(p, 64, None, 0, p, p, p, 0.9, 1, p, 128, p, 64, 16, 64, None, 0, p, None, 0, 0.0, None, 0,
                                         None, 0, None, 0, None) == -1
    # exact GraphNorm accumulators: hidden 64 only; a glass_gn_src with a missing pointer is rejected before any launch
    assert lib.glass_gn_exact_supported(64) == 1 and lib.glass_gn_exact_supported(128) == 0 and lib.glass_gn_exact_supported(16) == 0
    assert lib.glass_gn_exact_words(64) > 0 and lib.glass_gn_exact_words(64) % (2 * 64 * 2) == 0
    bad = _lib.GnSrc(p, 1, 4, p, None, p, 1e-5)
    assert lib.glass_dual_linear_fwd_f32(p, 64, None, 0, p, p, p, 0.9, 1, p, 128, p, 64, 16, 64, None, 0, p, bad.ptr, 0, 0.0, None,
                                         0, p, 64, None, 0, None) == -1
    assert lib.glass_graphnorm_stats_exact_f32(p, 64, 16, 64, None, 16, None) == -1


def test_round6_entry_points_validate_on_the_host():
    """glass_step_head_f32 (prologue || labels over a device cursor) and glass_peer_allreduce_adam_f32 (one-shot exchange + Adam):
    argument checks answer with codes before any launch — no GPU needed."""
    import ctypes
    from glass_amd import _lib, peer
    lib = _lib.load()
    x = np.zeros(64, dtype=np.float32)
    p = x.ctypes.data
    pro = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0, 0]   # an empty prologue (no pack job, no table, no zero-fill)
    # no cursor / no label buffers
    assert lib.glass_step_head_f32(*pro, None, 8, 7, 8, p, p, p, p, p, p, 100, None) == -1
    assert lib.glass_step_head_f32(*pro, p, 8, 7, 8, None, p, p, p, p, p, 100, None) == -1
    # a batch of zero rows, a target row size that is not a multiple of 4 bytes, a misaligned cursor
    assert lib.glass_step_head_f32(*pro, p, 0, 7, 8, p, p, p, p, p, p, 100, None) == -1
    assert lib.glass_step_head_f32(*pro, p, 8, 7, 6, p, p, p, p, p, p, 100, None) == -1
    assert lib.glass_step_head_f32(*pro, p + 4, 8, 7, 8, p, p, p, p, p, p, 100, None) == -1
    assert b"step_head" in lib.glass_last_error_string()
    # one-shot exchange: world size out of range, a rank without a mapping, a zero spin limit
    grp = peer._PeerGroupStruct()
    grp.world, grp.rank = 9, 0
    args = (100, p, p, p, p, 0.9, 0.999, 1e-8, 0.0, p, p, p)
    assert lib.glass_peer_allreduce_adam_f32(ctypes.byref(grp), *args, 1000, None, None) == -1
    grp.world, grp.rank = 2, 0
    grp.grad[0], grp.flags[0] = p, p          # rank 1 has no arena / flag block
    assert lib.glass_peer_allreduce_adam_f32(ctypes.byref(grp), *args, 1000, None, None) == -1
    assert b"rank 1" in lib.glass_last_error_string()
    grp.grad[1], grp.flags[1] = p, p
    assert lib.glass_peer_allreduce_adam_f32(ctypes.byref(grp), *args, 0, None, None) == -1
    assert lib.glass_peer_alloc(0, None) == -1 and lib.glass_peer_export(None, None) == -1 and lib.glass_peer_import(None, None) == -1
    assert lib.glass_peer_free(None) == 0 and lib.glass_peer_close(None) == 0


def test_repeatable_entry_points_refuse_instead_of_falling_back_to_float_atomics():
    """an entry point that documents bitwise repeatability never reaches a float atomicAdd — beyond the
    ordered scatter's LDS staging (and for max pooling) it returns GLASS_E_WS and names the workspace form; the one
    float-atomic scatter is exported under its own name.  The refusals happen before any HIP call (no GPU needed)."""
    from glass_amd import _lib
    lib = _lib.load()
    x = np.zeros(64, dtype=np.float32)
    p = x.ctypes.data
    E_WS = -4
    # sum pooling, 200 x 155 padded entries > 12 288: refused, message names the exact form
    assert lib.glass_segment_pool_bwd_f32(p, 64, p, 200, 155, 0, None, p, 64, 50000, 64, None) == E_WS
    msg = lib.glass_last_error_string()
    assert b"glass_segment_pool_bwd_exact_f32" in msg and b"glass_segment_pool_bwd_atomic_f32" in msg
    # max pooling at any size: refused (its exact form is glass_segment_pool_max_bwd_exact_f32)
    assert lib.glass_segment_pool_bwd_f32(p, 64, p, 8, 10, 2, p, p, 64, 1000, 64, None) == E_WS
    assert b"glass_segment_pool_max_bwd_exact_f32" in lib.glass_last_error_string()
    # the atomic form validates like the others (max pooling without argmax -> GLASS_E_ARG)
    assert lib.glass_segment_pool_bwd_atomic_f32(p, 64, p, 8, 10, 2, None, p, 64, 1000, 64, None) == -1
    # fused readout beyond 16 384 entries without scatter_ws: refused
    args = [p, 128, p, p, p, p, 200, 155, 0, p, p, p, 0, 6, p, p, p, p, p, 128, p, p, 1, p, p, p, 1, p, 50000, 128, None, None, None,
            None, None, 0, None, None, None]
    assert lib.glass_readout_scatter_ws_bytes(50000, 200, 155) > 0
    assert lib.glass_readout_train_f32(*args) == E_WS and b"scatter_ws" in lib.glass_last_error_string()
    # no float atomicAdd left in the readout at all, and exactly one kernel with one in the pool file
    src = open(os.path.join(ROOT, "glass_amd", "csrc", "readout.hip")).read()
    assert not re.search(r"atomicAdd\(\s*(dst|dx|djk|demb)", src)
    pool = open(os.path.join(ROOT, "glass_amd", "csrc", "pool.hip")).read()
    assert pool.count("void pool_bwd_kernel(") == 1 and "pool_bwd_launch_atomic(" in pool


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from glass_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libglass_hip.so"))
    with pytest.raises(_lib.GlassHipError, match="no CPU fallback"):
        _lib.load()


def test_plan_builder_property_based():
    """hypothesis: for arbitrary degree sequences the plan tiles every row exactly once (sweep ranges in order,
    long rows covered by contiguous whole-batch chunks, partial slots numbered consecutively)."""
    from hypothesis import given, settings, strategies as st

    degrees = st.lists(st.one_of(st.integers(0, 40), st.integers(0, 600), st.sampled_from([255, 256, 2047, 2048, 2049, 9000])),
                       min_size=0, max_size=300)

    @settings(max_examples=60, deadline=None)
    @given(degrees)
    def check(degs):
        _check_plan(np.concatenate([[0], np.cumsum(np.asarray(degs, dtype=np.int64))]))

    check()


def test_dense_capability_record():
    """glass_dense_caps_query: ONE record per hidden width (the per-feature queries collapsed) — host
    arithmetic, agrees with the thin per-feature wrappers that remain, and tells which widths fall back to library GEMMs."""
    from glass_amd import _lib
    lib = _lib.load()
    fam = {h: _lib.dense_caps(h).family for h in (8, 17, 20, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024)}
    assert fam == {8: 1, 17: 1, 20: 1, 32: 1, 48: 0, 64: 2, 96: 0, 128: 3, 192: 0, 256: 3, 384: 0, 512: 3, 1024: 0}
    for h in (8, 20, 64, 128, 256, 512):
        c = _lib.dense_caps(h)
        assert c.weight_layout == lib.glass_dual_linear_layout(h) and c.stat_rows == lib.glass_dual_linear_stat_rows(h)
        assert c.fwd_layout_comb == lib.glass_dual_linear_fwd_layout(h, 2 * h) and c.dgrad_layout_trans == lib.glass_dual_linear_dgrad_layout(h, h)
        assert c.fwd_gather == lib.glass_dual_linear_fwd_gather_supported(h)
        assert c.gn_exact == lib.glass_gn_exact_supported(h) and c.comb_eff == lib.glass_comb_eff_supported(h)
        assert c.act_codes == (1 << 1) | (1 << 2)   # ELU and ReLU fused at every served width
    assert [_lib.dense_caps(h).serve_width for h in (8, 20, 33, 48, 64, 96, 128, 200, 256, 300, 512, 600)] == \
        [8, 20, 64, 64, 64, 128, 128, 256, 256, 512, 512, 0]
    c64, c128 = _lib.dense_caps(64), _lib.dense_caps(128)
    assert c64.pair_head == 1 and c128.pair_head == 0 and c64.comb_eff == 1 and c128.comb_eff == 0 and c128.comb_eff_fwd == 1
    assert lib.glass_dense_caps_query(0, None) == -1
    # the product form of the LDS-tiled family: the record names the family's DEFAULT; a call opts out in its own act word
    assert _lib.dense_caps(256).product_form == 1 and _lib.dense_caps(128).product_form == 1
    assert _lib.dense_caps(64).product_form == 1   # round 6: the staged hidden-64 kernels form split products by default too
    assert _lib.dense_caps(20).product_form == 0   # plain fmaf on the narrow family


def test_library_keeps_no_mutable_state():
    """"no global state -> re-entrant and thread-safe per stream".  The library reads no environment variable
    (laboratory knobs are constants in the product build), exports no setter, and the one behavioural choice a caller has —
    the product form of the tiled dense family — is an option bit of each call's `act` word (the GPU side of this contract:
    tests/test_gpu_hardening.py::test_two_threads_two_streams_two_product_forms)."""
    from glass_amd import _lib
    lib = _lib.load()
    csrc = os.path.join(ROOT, "glass_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            text = open(os.path.join(csrc, name)).read()
            assert "getenv" not in text, f"{name} reads the environment"
            assert not re.search(r"^static\s+std::atomic|^std::atomic", text, re.M), f"{name} has a process-global atomic"
    header = open(os.path.join(ROOT, "include", "glass_hip.h")).read()
    assert "GLASS_DENSE_F32_PRODUCTS" in header and "GLASS_ACT_MASK" in header
    for gone in ("glass_dense_product_form", "glass_dense_product_form_set"):
        assert not hasattr(lib, gone) and (gone + "(") not in header
    assert _lib.DENSE_F32_PRODUCTS == 0x100 and _lib.ACT_MASK == 0xff
    # an act word with an unknown activation code under the mask is still refused; option bits alone do not make it valid
    x = np.zeros(64, dtype=np.float32)
    p = x.ctypes.data
    assert lib.glass_dual_linear_fwd_f32(p, 64, None, 0, p, p, p, 0.9, 7 | 0x100, p, 128, p, 64, 16, 64, None, 0, None, None, 0, 0.0,
                                         None, 0, None, 0, None, 0, None) != 0
