// K2 (adjacency normalisation), K4 (max-zero-one label), K3 (label + embedding gather) and the
// label-conditioned mix of GLASSConv.  All HBM-bound streaming / gather kernels: 16-B accesses,
// TC lanes per row, 4 rows in flight per thread.
#include "common.h"

namespace glass {

constexpr int kUnrollE = 4;

struct RowTiling {
    int vw, cw, tc, tc_log2, rpb, ctiles;
};

static RowTiling row_tiling(int64_t C, bool vec_ok) {
    RowTiling t;
    t.vw = vec_ok ? 4 : 1;
    t.cw = (int)ceil_div(C, t.vw);
    t.tc = pow2_ceil_cap(t.cw, kBlock);
    t.tc_log2 = 0;
    while ((1 << t.tc_log2) < t.tc) ++t.tc_log2;
    t.rpb = kBlock / t.tc;
    t.ctiles = (int)ceil_div(t.cw, t.tc);
    return t;
}

static unsigned row_blocks(int64_t n_rows, const RowTiling& t, int unroll) {
    int64_t b = ceil_div(n_rows, (int64_t)t.rpb * unroll);
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (unsigned)b;
}

template <int VW> struct Vf;
template <> struct Vf<4> {
    float a[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 v = *reinterpret_cast<const float4*>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    }
};
template <> struct Vf<1> {
    float a[1];
    __device__ __forceinline__ void load(const float* p) { a[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = a[0]; }
};

// ---- K2: buildAdj values (reference impl/models.py:83-111) ------------------------------------
// one wave per row; lanes stride the row's edges, fixed-order shuffle reduction.
__global__ __launch_bounds__(kBlock) void adj_degree_kernel(const int32_t* __restrict__ rowptr,
                                                            const float* __restrict__ w, int n_rows,
                                                            float* __restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float s = 0.f;
    for (int e = e0 + lane; e < e1; e += kWave) s += w[e];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k);
    if (lane == 0) deg[row] = s < 0.5f ? s + 1.0f : s;  // models.py:93-94
}

__global__ __launch_bounds__(kBlock) void adj_values_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ w, int n_rows, int aggr,
                                                            const float* __restrict__ deg, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    const float d = deg[row];
    const float inv = 1.0f / d;                 // mean: (1/deg)[row] * w      (models.py:95-99)
    const float rs = 1.0f / sqrtf(d);           // gcn : deg^-1/2              (models.py:104)
    for (int e = e0 + lane; e < e1; e += kWave) {
        float v = w[e];
        if (aggr == 0) v = inv * v;
        else if (aggr == 2) v = rs * v * (1.0f / sqrtf(deg[col[e]]));
        val[e] = v;
    }
}

// ---- K4: MaxZOZ (reference impl/utils.py:32-45) ------------------------------------------------
__global__ void maxzoz_scatter_kernel(const int64_t* __restrict__ pos, int64_t n_pos, int64_t* __restrict__ z,
                                      int64_t n_nodes) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = pos[i];
        if (p >= 0 && p < n_nodes) z[p] = 1;  // all writers store 1: idempotent, race-free result
    }
}

// ---- K3+K4: out[n,:] = W[x[n],:], mask[n] = label ----------------------------------------------
template <int VW>
__global__ __launch_bounds__(kBlock) void embed_label_kernel(const int64_t* __restrict__ x,
                                                             const float* __restrict__ W, int64_t V,
                                                             const int64_t* __restrict__ z,
                                                             const int64_t* __restrict__ pos, int64_t n_pos,
                                                             float* __restrict__ out, int64_t ldo,
                                                             uint8_t* __restrict__ mask, int64_t N, int H,
                                                             int tc_log2) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    if (blockIdx.y == 0) {
        if (z) {
            for (int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x; n < N; n += (int64_t)gridDim.x * kBlock)
                mask[n] = z[n] > 0 ? 1 : 0;
        } else if (pos) {  // mask was zero-filled by the launch function
            for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pos; i += (int64_t)gridDim.x * kBlock) {
                const int64_t p = pos[i];
                if (p >= 0 && p < N) mask[p] = 1;
            }
        }
    }
    if (c0 >= H) return;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnrollE) {
        Vf<VW> v[kUnrollE];
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
#pragma unroll
            for (int k = 0; k < VW; ++k) v[u].a[k] = 0.f;
            if (rr < N) {
                const int64_t idx = x[rr];
                if (idx >= 0 && idx < V) v[u].load(W + idx * H + c0);  // out-of-range -> zeros (validated by caller)
            }
        }
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) v[u].store(out + rr * ldo + c0);
        }
    }
}

// ---- mix: out = mask ? zr*a1 + (1-zr)*a0 : zr*a0 + (1-zr)*a1,  a = act(T) ----------------------
// reference impl/models.py:158-162 (act = ELU on trans_fns) and 169-173 (no act on comb_fns)
template <int VW>
__global__ __launch_bounds__(kBlock) void mix_fwd_kernel(const float* __restrict__ T, int64_t ldt,
                                                         const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                         float* __restrict__ out, int64_t ldo, int64_t N, int H,
                                                         int tc_log2) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    if (c0 >= H) return;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnrollE) {
        Vf<VW> t1[kUnrollE], t0[kUnrollE];
        float w1[kUnrollE];
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) {
                t1[u].load(T + rr * ldt + c0);
                t0[u].load(T + rr * ldt + H + c0);
                w1[u] = mask[rr] ? zr : omz;
            }
        }
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            const float w0 = (w1[u] == zr) ? omz : zr;
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                float a1 = t1[u].a[k], a0 = t0[u].a[k];
                a1 = act_exact(act, a1), a0 = act_exact(act, a0);
                t1[u].a[k] = w1[u] * a1 + w0 * a0;
            }
            t1[u].store(out + rr * ldo + c0);
        }
    }
}

template <int VW>
__global__ __launch_bounds__(kBlock) void mix_bwd_kernel(const float* __restrict__ dout, int64_t ldd,
                                                         const float* __restrict__ T, int64_t ldt,
                                                         const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                         float* __restrict__ dT, int64_t lddt, int64_t N, int H,
                                                         int tc_log2) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    if (c0 >= H) return;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnrollE) {
        Vf<VW> g[kUnrollE], t1[kUnrollE], t0[kUnrollE];
        float w1[kUnrollE];
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) {
                g[u].load(dout + rr * ldd + c0);
                if (act != GLASS_ACT_NONE) {
                    t1[u].load(T + rr * ldt + c0);
                    t0[u].load(T + rr * ldt + H + c0);
                }
                w1[u] = mask[rr] ? zr : omz;
            }
        }
#pragma unroll
        for (int u = 0; u < kUnrollE; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            const float w0 = (w1[u] == zr) ? omz : zr;
            Vf<VW> d1, d0;
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                float g1 = g[u].a[k] * w1[u], g0 = g[u].a[k] * w0;
                g1 *= act_grad(act, t1[u].a[k]), g0 *= act_grad(act, t0[u].a[k]);
                d1.a[k] = g1;
                d0.a[k] = g0;
            }
            d1.store(dT + rr * lddt + c0);
            d0.store(dT + rr * lddt + H + c0);
        }
    }
}

// Plain fill kernels instead of hipMemsetAsync: inside a captured hipGraph a memset NODE followed by the scatter
// kernel was observed to leave stale labels behind after other work had run between replays (z of the same pos
// differed between replays; found by hashing every intermediate of a replayed step) — kernels order reliably.
__global__ __launch_bounds__(kBlock) void fill_i64_kernel(int64_t* __restrict__ p, int64_t n, int64_t v) {
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) p[k] = v;
}

__global__ __launch_bounds__(kBlock) void fill_u8_kernel(uint8_t* __restrict__ p, int64_t n, uint8_t v) {
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) p[k] = v;
}

// two small device-to-device copies in one launch (the batch's pos and target into the captured step's buffers)
__global__ __launch_bounds__(kBlock) void copy_pair_kernel(uint32_t* __restrict__ d0, const uint32_t* __restrict__ s0,
                                                           int64_t w0, uint32_t* __restrict__ d1,
                                                           const uint32_t* __restrict__ s1, int64_t w1) {
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < w0 + w1; k += (int64_t)gridDim.x * kBlock) {
        if (k < w0) d0[k] = s0[k];
        else d1[k - w0] = s1[k - w0];
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_adj_values_f32(const int32_t* rowptr, const int32_t* col, const float* w, int64_t n_rows,
                                    int aggr, float* deg_ws, float* val, void* stream) {
    GLASS_REQUIRE(rowptr && deg_ws && n_rows >= 0, "adj_values: null pointer");
    if (aggr < 0 || aggr > 2) {
        set_error("adj_values: unknown aggr %d (0=mean,1=sum,2=gcn)", aggr);
        return GLASS_E_UNSUPPORTED;  // reference raises NotImplementedError (models.py:110-111)
    }
    if (n_rows == 0) return 0;
    // col / w / val may be null for a graph without edges (every row empty): the kernels never touch them
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div(n_rows, kBlock / kWave);
    hipLaunchKernelGGL(adj_degree_kernel, dim3(grid), dim3(kBlock), 0, st, rowptr, w, (int)n_rows, deg_ws);
    hipLaunchKernelGGL(adj_values_kernel, dim3(grid), dim3(kBlock), 0, st, rowptr, col, w, (int)n_rows, aggr, deg_ws,
                       val);
    return launch_status("glass_adj_values_f32");
}

extern "C" int glass_maxzoz_i64(const int64_t* pos, int64_t n_pos, int64_t* z, int64_t n_nodes, void* stream) {
    GLASS_REQUIRE(z && n_nodes >= 0 && n_pos >= 0 && (pos || n_pos == 0), "maxzoz: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (n_nodes == 0) return 0;
    int64_t fb = ceil_div(n_nodes, kBlock);
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st, z, n_nodes, (int64_t)0);
    if (n_pos > 0) {
        int64_t b = ceil_div(n_pos, 256);
        if (b > 1024) b = 1024;
        hipLaunchKernelGGL(maxzoz_scatter_kernel, dim3((unsigned)b), dim3(256), 0, st, pos, n_pos, z, n_nodes);
    }
    return launch_status("glass_maxzoz_i64");
}

extern "C" int glass_embed_label_f32(const int64_t* x, const float* W, int64_t V, const int64_t* z,
                                     const int64_t* pos, int64_t n_pos, float* out, int64_t ldo, uint8_t* mask,
                                     int64_t n_nodes, int64_t H, void* stream) {
    GLASS_REQUIRE(x && W && out && mask, "embed_label: null pointer");
    GLASS_REQUIRE(n_nodes > 0 && H > 0 && V > 0 && ldo >= H, "embed_label: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (!z && !(pos == nullptr && n_pos < 0)) {  // no z, no pos: all labeled (n_pos < 0: mask is an input, left alone)
        int64_t fb = ceil_div(n_nodes, kBlock);
        if (fb > 2048) fb = 2048;
        hipLaunchKernelGGL(fill_u8_kernel, dim3((unsigned)fb), dim3(kBlock), 0, st, mask, n_nodes, (uint8_t)(pos ? 0 : 1));
    }
    const bool vec = H % 4 == 0 && ldo % 4 == 0 && aligned16(W) && aligned16(out);
    const RowTiling t = row_tiling(H, vec);
    dim3 grid(row_blocks(n_nodes, t, kUnrollE), t.ctiles);
    if (vec)
        hipLaunchKernelGGL(embed_label_kernel<4>, grid, dim3(kBlock), 0, st, x, W, V, z, pos, n_pos, out, ldo, mask,
                           n_nodes, (int)H, t.tc_log2);
    else
        hipLaunchKernelGGL(embed_label_kernel<1>, grid, dim3(kBlock), 0, st, x, W, V, z, pos, n_pos, out, ldo, mask,
                           n_nodes, (int)H, t.tc_log2);
    return launch_status("glass_embed_label_f32");
}

extern "C" int glass_mix_fwd_f32(const float* T, int64_t ldt, const uint8_t* mask, double z_ratio, int act, float* out,
                                 int64_t ldo, int64_t n_nodes, int64_t H, void* stream) {
    GLASS_REQUIRE(T && mask && out, "mix_fwd: null pointer");
    GLASS_REQUIRE(n_nodes > 0 && H > 0 && ldt >= 2 * H && ldo >= H, "mix_fwd: bad sizes");
    const bool vec = H % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0 && aligned16(T) && aligned16(out);
    const RowTiling t = row_tiling(H, vec);
    dim3 grid(row_blocks(n_nodes, t, kUnrollE), t.ctiles);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(mix_fwd_kernel<4>, grid, dim3(kBlock), 0, st, T, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio), act, out, ldo, n_nodes,
                           (int)H, t.tc_log2);
    else
        hipLaunchKernelGGL(mix_fwd_kernel<1>, grid, dim3(kBlock), 0, st, T, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio), act, out, ldo, n_nodes,
                           (int)H, t.tc_log2);
    return launch_status("glass_mix_fwd_f32");
}

extern "C" int glass_mix_bwd_f32(const float* dout, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask,
                                 double z_ratio, int act, float* dT, int64_t lddt, int64_t n_nodes, int64_t H,
                                 void* stream) {
    GLASS_REQUIRE(dout && mask && dT && (T || act == GLASS_ACT_NONE), "mix_bwd: null pointer");
    GLASS_REQUIRE(n_nodes > 0 && H > 0 && ldd >= H && lddt >= 2 * H && (act == GLASS_ACT_NONE || ldt >= 2 * H),
                  "mix_bwd: bad sizes");
    const bool vec = H % 4 == 0 && ldd % 4 == 0 && lddt % 4 == 0 && aligned16(dout) && aligned16(dT) &&
                     (act == GLASS_ACT_NONE || (ldt % 4 == 0 && aligned16(T)));
    const RowTiling t = row_tiling(H, vec);
    dim3 grid(row_blocks(n_nodes, t, kUnrollE), t.ctiles);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(mix_bwd_kernel<4>, grid, dim3(kBlock), 0, st, dout, ldd, T, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio), act, dT,
                           lddt, n_nodes, (int)H, t.tc_log2);
    else
        hipLaunchKernelGGL(mix_bwd_kernel<1>, grid, dim3(kBlock), 0, st, dout, ldd, T, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio), act, dT,
                           lddt, n_nodes, (int)H, t.tc_log2);
    return launch_status("glass_mix_bwd_f32");
}

extern "C" int glass_copy_pair(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1,
                               int64_t bytes1, void* stream) {
    GLASS_REQUIRE(dst0 && src0 && dst1 && src1 && bytes0 > 0 && bytes1 > 0, "copy_pair: null pointer / empty copy");
    GLASS_REQUIRE(bytes0 % 4 == 0 && bytes1 % 4 == 0 && ((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) % 4 == 0,
                  "copy_pair: 4-byte granularity");
    const int64_t words = (bytes0 + bytes1) / 4;
    int64_t blocks = ceil_div(words, kBlock);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_pair_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (uint32_t*)dst0,
                       (const uint32_t*)src0, bytes0 / 4, (uint32_t*)dst1, (const uint32_t*)src1, bytes1 / 4);
    return launch_status("glass_copy_pair");
}

// ---- measurement aid: a kernel that does nothing, launched with a given geometry --------------------------------------
// bench.py replays the training step's chain of launches with these (same grids, block sizes and dynamic LDS) to measure
// the latency floor of that chain on the box it runs on (`step_floor`): what the step would cost if every kernel were free.
namespace glass {
__global__ void empty_kernel() {}
}  // namespace glass

extern "C" int glass_empty_launch(int64_t grid_x, int64_t grid_y, int64_t grid_z, int64_t block, int64_t lds_bytes, void* stream) {
    GLASS_REQUIRE(grid_x > 0 && grid_y > 0 && grid_z > 0 && grid_x < (1ll << 31) && grid_y < 65536 && grid_z < 65536 && block > 0 &&
                      block <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024,
                  "empty_launch: bad geometry");
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)glass::empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(glass::empty_kernel, dim3((unsigned)grid_x, (unsigned)grid_y, (unsigned)grid_z), dim3((unsigned)block),
                       (size_t)lds_bytes, (hipStream_t)stream);
    return glass::launch_status("glass_empty_launch");
}
