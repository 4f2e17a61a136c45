// Embedding lookup + emb_gn (+ dropout) through the embedding TABLE instead of the node matrix.
// reference impl/models.py:246-251:  x = input_emb(x); x = emb_gn(x); x = dropout(x)
//
// h0 = W[x] holds only V distinct rows (V = max_deg + 1; ~60 for the degree feature of ppi_bp), so the
// whole-graph GraphNorm statistics are count-weighted sums over the TABLE:
//     sum_n h0[n,c] = sum_v cnt[v] * W[v,c],     sum_n h0[n,c]^2 = sum_v cnt[v] * W[v,c]^2
// (cnt[v] = nodes using row v: static per dataset, the row lengths of graph.Selection's CSR).  Forward:
// one small kernel turns W into the normalised table Wn = W*scale + shift (same fma as the GraphNorm
// apply kernel) and the gather kernel reads Wn — instead of gather + statistics + finalize + apply over
// [N,H].  Backward: with G[v] = sum_{n: x[n]=v} g[n] (selection-matrix product on K1, g already masked
// by the dropout), the GraphNorm backward collapses to the table as well:
//     S1 = sum_v G[v],  S2 = sum_v G[v]*xhat[v],  dW[v] = A*G[v] + cnt[v]*(Bx*W[v] + K)
// — instead of backward statistics + finalize + apply over [N,H] and an extra add into dW.
#include "common.h"
#include "gn_math.h"

namespace glass {

constexpr int kTabCols = 16;                    // columns per workgroup (one per lane of a 16-lane group)
constexpr int kTabSlots = kBlock / kTabCols;    // row slots per workgroup (16): V ~ 60 rows -> 4 sequential loads each

// Sum the two fp64 accumulators of this thread's column over the row slots (fixed order); result valid in slot 0.
__device__ __forceinline__ void slot_reduce(double& a, double& b, double* lds, int tc, int tr) {
    lds[threadIdx.x * 2] = a;
    lds[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    if (tr == 0)
        for (int r = 1; r < kTabSlots; ++r) {
            a += lds[(r * kTabCols + tc) * 2];
            b += lds[(r * kTabCols + tc) * 2 + 1];
        }
}

__global__ __launch_bounds__(kBlock) void emb_table_fwd_kernel(const float* __restrict__ W, int V, int H,
                                                               const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ alpha, float eps,
                                                               float* __restrict__ saved, float* __restrict__ table,
                                                               uint8_t* __restrict__ mask_fill, int fill_value,
                                                               int64_t n_nodes) {
    __shared__ double lds[kBlock * 2];
    __shared__ float coef[2 * kTabCols];
    // label bytes initialised here (0 before the gather kernel scatters pos, 1 = everything labeled) instead of by a
    // memset launch (two fill launches for a byte count that is not a multiple of 4)
    if (mask_fill)
        for (int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x; n < n_nodes; n += (int64_t)gridDim.x * kBlock)
            mask_fill[n] = (uint8_t)fill_value;
    if ((int)blockIdx.x >= (H + kTabCols - 1) / kTabCols) return;  // extra workgroups only help with the fill
    const int tc = threadIdx.x & (kTabCols - 1), tr = threadIdx.x / kTabCols;
    const int c = blockIdx.x * kTabCols + tc;
    const bool ok = c < H;
    double s = 0.0, q = 0.0;
    if (ok)
        for (int v = tr; v < V; v += kTabSlots) {
            const double cn = (double)(rowptr[v + 1] - rowptr[v]);
            const double w = (double)W[(int64_t)v * H + c];
            s += cn * w;
            q += cn * w * w;
        }
    slot_reduce(s, q, lds, tc, tr);
    if (tr == 0 && ok) {
        float mu, rstd, scale, shift;
        gn_fwd_coeffs(s, q, (double)rowptr[V], gamma[c], beta[c], alpha[c], eps, mu, rstd, scale, shift);
        saved[c] = mu;
        saved[H + c] = rstd;
        saved[2 * H + c] = scale;
        saved[3 * H + c] = shift;
        coef[tc] = scale;
        coef[kTabCols + tc] = shift;
    }
    __syncthreads();
    if (!ok) return;
    const float scale = coef[tc], shift = coef[kTabCols + tc];
    for (int v = tr; v < V; v += kTabSlots) table[(int64_t)v * H + c] = fmaf(W[(int64_t)v * H + c], scale, shift);
}

__global__ __launch_bounds__(kBlock) void emb_table_bwd_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                               int V, int H, const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ saved, float* __restrict__ dW,
                                                               int accumulate_w, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                               int accumulate) {
    __shared__ double lds[kBlock * 2];
    __shared__ float coef[3 * kTabCols];
    const int tc = threadIdx.x & (kTabCols - 1), tr = threadIdx.x / kTabCols;
    const int c = blockIdx.x * kTabCols + tc;
    const bool ok = c < H;
    double s1 = 0.0, s2 = 0.0;
    if (ok) {
        const float mu = saved[c], rstd = saved[H + c], al = alpha[c];
        for (int v = tr; v < V; v += kTabSlots) {
            const float g = G[(int64_t)v * H + c];
            const float xhat = (W[(int64_t)v * H + c] - al * mu) * rstd;
            s1 += (double)g;
            s2 += (double)g * (double)xhat;
        }
    }
    slot_reduce(s1, s2, lds, tc, tr);
    if (tr == 0 && ok) {
        float A, Bx, K, da;
        gn_bwd_coeffs(s1, s2, (double)rowptr[V], gamma[c], alpha[c], saved[c], saved[H + c], A, Bx, K, da);
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
        if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + da;
        coef[tc] = A;
        coef[kTabCols + tc] = Bx;
        coef[2 * kTabCols + tc] = K;
    }
    __syncthreads();
    if (!ok) return;
    const float A = coef[tc], Bx = coef[kTabCols + tc], K = coef[2 * kTabCols + tc];
    for (int v = tr; v < V; v += kTabSlots) {
        const int64_t o = (int64_t)v * H + c;
        const float cn = (float)(rowptr[v + 1] - rowptr[v]);
        const float d = fmaf(A, G[o], cn * fmaf(Bx, W[o], K));
        dW[o] = accumulate_w ? dW[o] + d : d;
    }
}

// out[n,:] = dropout(table[x[n],:]), mask[n] = label byte (from z, or scattered from pos, or all ones)
template <int VW> struct Vg;
template <> struct Vg<4> {
    float a[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 v = *reinterpret_cast<const float4*>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    }
};
template <> struct Vg<1> {
    float a[1];
    __device__ __forceinline__ void load(const float* p) { a[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = a[0]; }
};

constexpr int kUnrollG = 4;

template <int VW>
__global__ __launch_bounds__(kBlock) void embed_gather_kernel(const int64_t* __restrict__ x,
                                                              const float* __restrict__ table, int64_t V,
                                                              const int64_t* __restrict__ z,
                                                              const int64_t* __restrict__ pos, int64_t n_pos,
                                                              float* __restrict__ out, int64_t ldo,
                                                              uint8_t* __restrict__ mask, int64_t N, int H,
                                                              int tc_log2, Drop drop,
                                                              const uint64_t* __restrict__ rng_state) {
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
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnrollG) {
        Vg<VW> v[kUnrollG];
#pragma unroll
        for (int u = 0; u < kUnrollG; ++u) {
            const int64_t rr = r + u * stride;
#pragma unroll
            for (int k = 0; k < VW; ++k) v[u].a[k] = 0.f;
            if (rr < N) {
                const int64_t idx = x[rr];
                if (idx >= 0 && idx < V) v[u].load(table + idx * H + c0);  // out-of-range -> zeros (validated by caller)
            }
        }
#pragma unroll
        for (int u = 0; u < kUnrollG; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            if (drop.p > 0.f) {
                float ds[VW];
                drop_scales<VW>(drop, rr, c0, ds);
#pragma unroll
                for (int k = 0; k < VW; ++k) v[u].a[k] *= ds[k];
            }
            v[u].store(out + rr * ldo + c0);
        }
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_embed_norm_fwd_f32(const int64_t* x, const float* W, int64_t V, const int32_t* class_rowptr,
                                        const float* gamma, const float* beta, const float* alpha, float eps,
                                        float* saved, float* table, const int64_t* z, const int64_t* pos,
                                        int64_t n_pos, float p_drop, const uint64_t* rng_state, uint64_t call_id,
                                        float* out, int64_t ldo, uint8_t* mask, int64_t n_nodes, int64_t H,
                                        void* stream) {
    GLASS_REQUIRE(x && W && class_rowptr && gamma && beta && alpha && saved && table && out && mask,
                  "embed_norm_fwd: null pointer");
    GLASS_REQUIRE(n_nodes > 0 && H > 0 && V > 0 && V <= GLASS_EMBED_NORM_MAX_ROWS && ldo >= H,
                  "embed_norm_fwd: bad sizes (V=%lld, at most %d table rows)", (long long)V, GLASS_EMBED_NORM_MAX_ROWS);
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "embed_norm_fwd: bad dropout args");
    hipStream_t st = (hipStream_t)stream;
    // no z: the table kernel fills the label bytes (0 when pos will be scattered by the gather kernel, else all 1)
    int64_t tab_blocks = ceil_div(H, kTabCols);
    const bool mask_given = !z && !pos && n_pos < 0;  // the label bytes are an INPUT (glass_batch_labels wrote them)
    if (!z && !mask_given) {  // a few more workgroups share the byte fill
        int64_t fill_blocks = ceil_div(n_nodes, (int64_t)kBlock * 16);
        if (fill_blocks > 64) fill_blocks = 64;
        if (fill_blocks > tab_blocks) tab_blocks = fill_blocks;
    }
    hipLaunchKernelGGL(emb_table_fwd_kernel, dim3((unsigned)tab_blocks), dim3(kBlock), 0, st, W, (int)V, (int)H,
                       class_rowptr, gamma, beta, alpha, eps, saved, table, (z || mask_given) ? nullptr : mask, pos ? 0 : 1, n_nodes);
    const bool vec = H % 4 == 0 && ldo % 4 == 0 && aligned16(table) && aligned16(out);
    const int vw = vec ? 4 : 1;
    const int cw = (int)ceil_div(H, vw);
    const int tc = pow2_ceil_cap(cw, kBlock);
    int tc_log2 = 0;
    while ((1 << tc_log2) < tc) ++tc_log2;
    int64_t blocks = ceil_div(n_nodes, (int64_t)(kBlock / tc) * kUnrollG);
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div(cw, tc));
    const Drop drop = make_drop(p_drop, call_id, H);
    if (vec)
        hipLaunchKernelGGL(embed_gather_kernel<4>, grid, dim3(kBlock), 0, st, x, table, V, z, pos, n_pos, out, ldo, mask,
                           n_nodes, (int)H, tc_log2, drop, rng_state);
    else
        hipLaunchKernelGGL(embed_gather_kernel<1>, grid, dim3(kBlock), 0, st, x, table, V, z, pos, n_pos, out, ldo, mask,
                           n_nodes, (int)H, tc_log2, drop, rng_state);
    return launch_status("glass_embed_norm_fwd_f32");
}

extern "C" int glass_embed_norm_bwd_f32(const float* G, const float* W, int64_t V, const int32_t* class_rowptr,
                                        const float* gamma, const float* alpha, const float* saved, float* dW,
                                        int accumulate_w, float* dgamma, float* dbeta, float* dalpha, int accumulate,
                                        int64_t H, void* stream) {
    GLASS_REQUIRE(G && W && class_rowptr && gamma && alpha && saved && dW, "embed_norm_bwd: null pointer");
    GLASS_REQUIRE(H > 0 && V > 0 && V <= GLASS_EMBED_NORM_MAX_ROWS, "embed_norm_bwd: bad sizes");
    hipLaunchKernelGGL(emb_table_bwd_kernel, dim3((unsigned)ceil_div(H, kTabCols)), dim3(kBlock), 0, (hipStream_t)stream, G,
                       W, (int)V, (int)H, class_rowptr, gamma, alpha, saved, dW, accumulate_w, dgamma, dbeta, dalpha,
                       accumulate);
    return launch_status("glass_embed_norm_bwd_f32");
}
