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
#include "emb_table.h"
#include "gn_math.h"

namespace glass {

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
    emb_table_fwd_block(blockIdx.x, W, V, H, rowptr, gamma, beta, alpha, eps, saved, table, lds, coef);
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

// The step's LAST launch on the table path: what is left of the selection product on K1 (rows cut into several chunks:
// their partial rows summed in slot order), the table backward above, and Adam over the whole parameter arena.
//   workgroups [0, n_tab): 16 table columns each — slot sums -> G, column sums, dW; then Adam on exactly the elements this
//     workgroup produced (table columns, emb_gn's three vectors), by the threads that wrote their gradients;
//   workgroups [n_tab, grid): Adam over the rest of the arena (every other gradient was final before this launch).
// Three dependent launches (spmm_reduce, emb_table_bwd, adam) become one.
struct TailAdam {
    float* p;        // nullptr: no optimizer step here
    float* g;
    float *m, *v;
    int64_t n;       // arena elements
    const float* lr;
    float beta1, beta2, eps, weight_decay;
    int64_t* step;
    int64_t off_W, off_gamma, off_beta, off_alpha;  // arena offsets of the table [V*H] and emb_gn's vectors [H]
};

// (W, gamma, alpha alias the parameter arena ad.p, which this kernel updates: no __restrict__ on them)
__global__ __launch_bounds__(kBlock) void emb_tail_kernel(float* __restrict__ G, const float* W, int V, int H,
                                                          const int32_t* __restrict__ rowptr,
                                                          const float* gamma, const float* alpha,
                                                          const float* __restrict__ saved, float* dW,
                                                          int accumulate_w, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                          int accumulate, const float* __restrict__ partials,
                                                          const int32_t* __restrict__ rrows, int n_reduce, TailAdam ad,
                                                          int n_tab) {
    __shared__ double lds[kBlock * 2];
    __shared__ float coef[3 * kTabCols];
    int64_t step_now = 0;
    AdamCoef ac{};
    if (ad.p) {
        step_now = ad.step[0] + 1;
        ac = adam_coef(step_now, ad.lr[0], ad.beta1, ad.beta2, ad.eps, ad.weight_decay);
    }
    if ((int)blockIdx.x >= n_tab) {  // Adam over everything this launch did not produce itself
        const int64_t nb = (int64_t)gridDim.x - n_tab;
        const int64_t tab_lo = ad.off_W, tab_hi = ad.off_W + (int64_t)V * H;
        for (int64_t k = ((int64_t)blockIdx.x - n_tab) * kBlock + threadIdx.x; k < ad.n; k += nb * kBlock) {
            if ((k >= tab_lo && k < tab_hi) || (k >= ad.off_gamma && k < ad.off_gamma + H) ||
                (k >= ad.off_beta && k < ad.off_beta + H) || (k >= ad.off_alpha && k < ad.off_alpha + H))
                continue;
            adam_update(ac, ad.p, ad.g[k], ad.m, ad.v, k);
        }
        adam_ticket(ad.step, step_now);
        return;
    }
    const int tc = threadIdx.x & (kTabCols - 1), tr = threadIdx.x / kTabCols;
    const int c = blockIdx.x * kTabCols + tc;
    const bool ok = c < H;
    // Small tables (the degree feature: ~60 rows): a thread owns <= 4 table elements of its column — everything it will
    // need of them (weight = parameter, row count, Adam moments) is requested BEFORE the slot sums, so the launch is not a
    // chain of six dependent round trips.
    constexpr int kOwn = 4;
    const bool small = V <= kOwn * kTabSlots;
    float w_own[kOwn], cn_own[kOwn], m_own[kOwn], v_own[kOwn];
    if (small) {
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int v = tr + kTabSlots * k;
            const bool live = ok && v < V;
            const int64_t o = (int64_t)(live ? v : 0) * H + (ok ? c : 0);
            w_own[k] = live ? W[o] : 0.f;
            cn_own[k] = live ? (float)(rowptr[v + 1] - rowptr[v]) : 0.f;
            m_own[k] = (live && ad.p) ? ad.m[ad.off_W + o] : 0.f;
            v_own[k] = (live && ad.p) ? ad.v[ad.off_W + o] : 0.f;
        }
    }
    // rows of the selection product that were cut into several chunks: partial rows summed in slot order
    if (ok)
        for (int rr = tr; rr < n_reduce; rr += kTabSlots) {
            const int row = rrows[3 * rr], first = rrows[3 * rr + 1], n = rrows[3 * rr + 2];
            float sum = 0.f;
            for (int k = 0; k < n; ++k) sum += partials[(int64_t)(first + k) * H + c];
            G[(int64_t)row * H + c] = sum;
        }
    if (n_reduce > 0) __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    float g_own[kOwn];
    if (ok && small) {
        const float mu = saved[c], rstd = saved[H + c], al = alpha[c];
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int v = tr + kTabSlots * k;
            g_own[k] = v < V ? G[(int64_t)v * H + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {  // (same order of the fp64 sums as the general loop below)
            if (tr + kTabSlots * k >= V) continue;
            const float xhat = (w_own[k] - al * mu) * rstd;
            s1 += (double)g_own[k];
            s2 += (double)g_own[k] * (double)xhat;
        }
    } else if (ok) {
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
        const float g_gamma = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        const float g_beta = (accumulate ? dbeta[c] : 0.f) + (float)s1;
        const float g_alpha = (accumulate ? dalpha[c] : 0.f) + da;
        dgamma[c] = g_gamma;
        dbeta[c] = g_beta;
        dalpha[c] = g_alpha;
        coef[tc] = A;
        coef[kTabCols + tc] = Bx;
        coef[2 * kTabCols + tc] = K;
        if (ad.p) {
            adam_update(ac, ad.p, g_gamma, ad.m, ad.v, ad.off_gamma + c);
            adam_update(ac, ad.p, g_beta, ad.m, ad.v, ad.off_beta + c);
            adam_update(ac, ad.p, g_alpha, ad.m, ad.v, ad.off_alpha + c);
        }
    }
    __syncthreads();
    if (ok && small) {
        const float A = coef[tc], Bx = coef[kTabCols + tc], K = coef[2 * kTabCols + tc];
#pragma unroll
        for (int k = 0; k < kOwn; ++k) {
            const int v = tr + kTabSlots * k;
            if (v >= V) continue;
            const int64_t o = (int64_t)v * H + c;
            float d = fmaf(A, g_own[k], cn_own[k] * fmaf(Bx, w_own[k], K));
            if (accumulate_w) d += dW[o];
            dW[o] = d;
            if (ad.p) {  // (the weight IS the parameter: W aliases ad.p + off_W)
                float pk = w_own[k];
                adam_element(ac, pk, d, m_own[k], v_own[k]);
                ad.m[ad.off_W + o] = m_own[k];
                ad.v[ad.off_W + o] = v_own[k];
                ad.p[ad.off_W + o] = pk;
            }
        }
    } else if (ok) {
        const float A = coef[tc], Bx = coef[kTabCols + tc], K = coef[2 * kTabCols + tc];
        for (int v = tr; v < V; v += kTabSlots) {
            const int64_t o = (int64_t)v * H + c;
            const float cn = (float)(rowptr[v + 1] - rowptr[v]);
            float d = fmaf(A, G[o], cn * fmaf(Bx, W[o], K));
            if (accumulate_w) d += dW[o];
            dW[o] = d;
            if (ad.p) adam_update(ac, ad.p, d, ad.m, ad.v, ad.off_W + o);  // (W aliases ad.p + off_W: read above, updated here)
        }
    }
    if (ad.p) adam_ticket(ad.step, step_now);
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

extern "C" int glass_embed_norm_bwd_adam_f32(float* G, const float* W, int64_t V, const int32_t* class_rowptr,
                                             const float* gamma, const float* alpha, const float* saved, float* dW,
                                             int accumulate_w, float* dgamma, float* dbeta, float* dalpha, int accumulate,
                                             int64_t H, const float* partials, const int32_t* reduce_rows, int64_t n_reduce,
                                             float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_param,
                                             const float* lr_dev, double beta1, double beta2, double eps,
                                             double weight_decay, int64_t* step_dev, int64_t off_W, int64_t off_gamma,
                                             int64_t off_beta, int64_t off_alpha, void* stream) {
    GLASS_REQUIRE(G && W && class_rowptr && gamma && alpha && saved && dW && dgamma && dbeta && dalpha,
                  "embed_norm_bwd_adam: null pointer");
    GLASS_REQUIRE(H > 0 && V > 0 && V <= GLASS_EMBED_NORM_MAX_ROWS && n_reduce >= 0 && (n_reduce == 0 || (partials && reduce_rows)),
                  "embed_norm_bwd_adam: bad sizes");
    TailAdam ad{};
    const int n_tab = (int)ceil_div(H, kTabCols);
    int64_t blocks = n_tab;
    if (param) {
        GLASS_REQUIRE(grad && exp_avg && exp_avg_sq && lr_dev && step_dev && n_param > 0 && off_W >= 0 &&
                          off_W + V * H <= n_param && off_gamma >= 0 && off_gamma + H <= n_param && off_beta >= 0 &&
                          off_beta + H <= n_param && off_alpha >= 0 && off_alpha + H <= n_param && dW == grad + off_W &&
                          W == param + off_W && dgamma == grad + off_gamma && dbeta == grad + off_beta &&
                          dalpha == grad + off_alpha,
                      "embed_norm_bwd_adam: the table and emb_gn's vectors must be views of the arena at the given offsets");
        ad = TailAdam{param, grad, exp_avg, exp_avg_sq, n_param, lr_dev, (float)beta1, (float)beta2, (float)eps,
                      (float)weight_decay, step_dev, off_W, off_gamma, off_beta, off_alpha};
        int64_t rest = ceil_div(n_param, kBlock);
        if (rest > 2048) rest = 2048;
        blocks += rest;
    }
    hipLaunchKernelGGL(emb_tail_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, G, W, (int)V, (int)H,
                       class_rowptr, gamma, alpha, saved, dW, accumulate_w, dgamma, dbeta, dalpha, accumulate, partials,
                       reduce_rows, (int)n_reduce, ad, n_tab);
    return launch_status("glass_embed_norm_bwd_adam_f32");
}
