// K6: whole-graph GraphNorm (+ optional ELU + inverted dropout) forward and backward.
// Replaces PyG GraphNorm(batch=None) as called at reference impl/models.py:165,249,257,266,271
// and the F.dropout / activation that follow it (impl/models.py:166,251,258-259).
//
// HBM-bound column reductions + elementwise passes over a row-major [N,C] matrix:
//   forward : 1 read (fp64 sum / sum-of-squares per column) + 1 read + 1 write
//   backward: 2 reads (dy, x) for the two column sums + 2 reads + 1 write
// Thread layout: TC lanes x 16 B cover a row (coalesced), 256/TC rows per workgroup step,
// 4 rows in flight per thread.  Column sums are fp64, combined across row slots through LDS and
// across workgroups by a second tiny kernel in fixed order -> bitwise repeatable.
#include "common.h"
#include "gn_math.h"
#include "gn_acc.h"

namespace glass {

constexpr int kUnroll = 4;
constexpr int kMaxStatBlocks = 1024;

struct Tiling {
    int vw;       // floats per access (4 if everything is 16-B aligned, else 1)
    int cw;       // column words = ceil(C / vw)
    int tc;       // lanes across columns (power of two <= 256)
    int tc_log2;
    int rpb;      // rows per workgroup step = 256 / tc
    int ctiles;   // grid.y
};

static Tiling make_tiling(int64_t C, bool vec_ok) {
    Tiling t;
    t.vw = vec_ok ? 4 : 1;
    t.cw = (int)ceil_div(C, t.vw);
    t.tc = pow2_ceil_cap(t.cw, kBlock);
    t.tc_log2 = 0;
    while ((1 << t.tc_log2) < t.tc) ++t.tc_log2;
    t.rpb = kBlock / t.tc;
    t.ctiles = (int)ceil_div(t.cw, t.tc);
    return t;
}

static int stat_blocks(int64_t n_rows, const Tiling& t) {
    int64_t b = ceil_div(n_rows, (int64_t)t.rpb * kUnroll * 2);
    if (b < 1) b = 1;
    if (b > kMaxStatBlocks) b = kMaxStatBlocks;
    return (int)b;
}

// scratch: [kMaxStatBlocks][2][C] doubles (column partials) + [4][C] floats (backward coefficients)
static int64_t ws_partials_bytes(int64_t C) { return (int64_t)kMaxStatBlocks * 2 * C * (int64_t)sizeof(double); }

template <int VW> struct F;
template <> struct F<4> {
    float a[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 v = *reinterpret_cast<const float4*>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    }
};
template <> struct F<1> {
    float a[1];
    __device__ __forceinline__ void load(const float* p) { a[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = a[0]; }
};

// Load per-column parameters for this thread's column word (zeros past C).
template <int VW>
__device__ __forceinline__ void load_cols(float (&dst)[VW], const float* src, int c0, int C) {
#pragma unroll
    for (int k = 0; k < VW; ++k) dst[k] = (c0 + k < C) ? src[c0 + k] : 0.f;
}

// Reduce this thread's 2*VW fp64 accumulators over the row slots of the workgroup (fixed order)
// and let row slot 0 write them to partial[blk][which][c].
template <int VW>
__device__ __forceinline__ void block_reduce_store(double (&s0)[VW], double (&s1)[VW], double* lds, int tc, int tr,
                                                   int TC, int rpb, double* partial, int c0, int C, int exact = 0) {
    double* mine = lds + (size_t)threadIdx.x * 2 * VW;
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        mine[k] = s0[k];
        mine[VW + k] = s1[k];
    }
    __syncthreads();
    if (tr == 0) {
        for (int r = 1; r < rpb; ++r) {
            const double* o = lds + (size_t)(r * TC + tc) * 2 * VW;
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                s0[k] += o[k];
                s1[k] += o[VW + k];
            }
        }
        if (exact) {  // `partial` = exact accumulators (gn_acc.h), forward scale
            long long* a = reinterpret_cast<long long*>(partial);
#pragma unroll
            for (int k = 0; k < VW; ++k)
                if (c0 + k < C) {
                    gn_acc_add(a, blockIdx.x % exact, 0, c0 + k, C, s0[k], kAccScaleFwd);
                    gn_acc_add(a, blockIdx.x % exact, 1, c0 + k, C, s1[k], kAccScaleFwd);
                }
            return;
        }
        double* p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < VW; ++k)
            if (c0 + k < C) {
                p[c0 + k] = s0[k];
                p[C + c0 + k] = s1[k];
            }
    }
}

// ---- forward statistics: per-column sum(x), sum(x^2) ------------------------------------------
template <int VW>
__global__ __launch_bounds__(kBlock) void gn_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int C,
                                                          int tc_log2, double* __restrict__ partial, int exact) {
    __shared__ double lds[kBlock * 2 * VW];
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    const bool ok = c0 < C;
    double s[VW], q[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) s[k] = q[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnroll) {
        F<VW> v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
#pragma unroll
            for (int k = 0; k < VW; ++k) v[u].a[k] = 0.f;
            if (ok && rr < N) v[u].load(x + rr * ldx + c0);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                const double d = (double)v[u].a[k];
                s[k] += d;
                q[k] += d * d;
            }
    }
    block_reduce_store<VW>(s, q, lds, tc, tr, TC, rpb, partial, c0, C, exact);
}

// The same sums at C = 64 straight into the exact accumulators, shaped for a launch that is all latency (round 6; the
// step's two statistics launches at ppi_bp-shape, 6.6 us each): ONE round of loads per thread where the graph allows it
// (row tiles of 16 * nst rows, one workgroup per CU: thread (rs, ga) takes rows rs + 16 st, columns 4 ga .. + 3, up to
// five rows in flight), the 16 row slots of a column folded by lane shuffles + one pass through LDS instead of a serial
// walk by 16 threads, and the 2 x 64 sums added by 128 threads (two atomics each) instead of 16 threads with 16 each.
__global__ __launch_bounds__(kBlock) void gn_stats64_exact_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int nst,
                                                                  long long* __restrict__ acc, int n_rep) {
    constexpr int C = 64, kRound = 5;
    __shared__ double red[4][2 * C];
    const int tid = threadIdx.x, rs = tid >> 4, ga = tid & 15, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 16 * nst;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int st0 = 0; st0 < nst; st0 += kRound) {
        float4 v[kRound];
#pragma unroll
        for (int u = 0; u < kRound; ++u) {
            const int64_t r = r0 + 16 * (st0 + u) + rs;
            v[u] = (st0 + u < nst && r < N) ? *reinterpret_cast<const float4*>(x + r * ldx + 4 * ga) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kRound; ++u) {
            const double a[4] = {(double)v[u].x, (double)v[u].y, (double)v[u].z, (double)v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += a[k];
                q[k] += a[k] * a[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // the wave's four row slots of column 4 ga + k (lanes ga, ga + 16, ga + 32, ga + 48)
        s[k] += __shfl_xor(s[k], 16);
        q[k] += __shfl_xor(q[k], 16);
        s[k] += __shfl_xor(s[k], 32);
        q[k] += __shfl_xor(q[k], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[w][4 * ga + k] = s[k];
            red[w][C + 4 * ga + k] = q[k];
        }
    }
    __syncthreads();
    if (tid < 2 * C) {  // fixed order over the four waves; the integer adds commute: bitwise repeatable
        const double t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        gn_acc_add(acc, blockIdx.x % n_rep, tid / C, tid % C, C, t, kAccScaleFwd);
    }
}

// ---- finalize: sum the workgroup partials in fixed order; derive mean / rstd / scale / shift ----
// Up to 8 partial buffers [nblk][2][C_each], source k covering columns k*C_each .. (k+1)*C_each - 1 (one source for
// the statistics kernel above; several when the statistics come from the epilogues of the kernels that wrote the
// column blocks of a jumping-knowledge buffer).
constexpr int kMaxStatSrc = 8;
struct StatSrc {
    const double* p[kMaxStatSrc];
};

__global__ __launch_bounds__(kBlock) void gn_finalize_src_kernel(StatSrc src, int nblk, int C_each, int C, int64_t N,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ alpha, float eps,
                                                                 float* __restrict__ saved) {
    __shared__ double lds[kBlock * 2];
    const int tc = threadIdx.x & (kFinCols - 1), tr = threadIdx.x / kFinCols;
    const int c = blockIdx.x * kFinCols + tc;
    const bool ok = c < C;
    double ss, qq;
    gn_sum_partials(ok ? src.p[c / C_each] : nullptr, nblk, C_each, ok ? c % C_each : 0, ok, tc, tr, lds, ss, qq);
    if (tr == 0 && ok)
        gn_fwd_coeffs(ss, qq, (double)N, gamma[c], beta[c], alpha[c], eps, saved[c], saved[C + c], saved[2 * C + c],
                      saved[3 * C + c]);
}

// ---- forward apply: y = dropout(act(x*scale + shift)) ------------------------------------------
template <int VW>
__global__ __launch_bounds__(kBlock) void gn_apply_kernel(const float* __restrict__ x, int64_t ldx,
                                                          float* __restrict__ y, int64_t ldy, int64_t N, int C,
                                                          int tc_log2, const float* __restrict__ saved, int act,
                                                          Drop drop, const uint64_t* __restrict__ rng_state) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    if (c0 >= C) return;
    float scale[VW], shift[VW];
    load_cols<VW>(scale, saved + 2 * C, c0, C);
    load_cols<VW>(shift, saved + 3 * C, c0, C);
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnroll) {
        F<VW> v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) v[u].load(x + rr * ldx + c0);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            float ds[VW];
#pragma unroll
            for (int k = 0; k < VW; ++k) ds[k] = 1.f;
            if (drop.p > 0.f) drop_scales<VW>(drop, rr, c0, ds);
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                float h = fmaf(v[u].a[k], scale[k], shift[k]);
                h = act_exact(act, h);
                v[u].a[k] = h * ds[k];
            }
            v[u].store(y + rr * ldy + c0);
        }
    }
}

// ---- backward statistics: S1 = sum(g), S2 = sum(g * xhat), g = dy * dropmask * act'(h) ---------
template <int VW>
__device__ __forceinline__ void bwd_g(float (&g)[VW], const float (&xv)[VW], const float (&scale)[VW],
                                      const float (&shift)[VW], int act, const Drop& drop, int64_t row, int c0) {
    if (drop.p > 0.f) {
        float ds[VW];
        drop_scales<VW>(drop, row, c0, ds);
#pragma unroll
        for (int k = 0; k < VW; ++k) g[k] *= ds[k];
    }
    if (act != GLASS_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < VW; ++k) g[k] *= act_grad(act, fmaf(xv[k], scale[k], shift[k]));
    }
}

template <int VW>
__global__ __launch_bounds__(kBlock) void gn_bwd_stats_kernel(const float* __restrict__ dy, int64_t lddy,
                                                              const float* __restrict__ x, int64_t ldx, int64_t N,
                                                              int C, int tc_log2, const float* __restrict__ saved,
                                                              const float* __restrict__ alpha, int act, Drop drop,
                                                              const uint64_t* __restrict__ rng_state,
                                                              double* __restrict__ partial) {
    __shared__ double lds[kBlock * 2 * VW];
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    const bool ok = c0 < C;
    float mu[VW], rstd[VW], scale[VW], shift[VW], al[VW];
    load_cols<VW>(mu, saved, c0, C);
    load_cols<VW>(rstd, saved + C, c0, C);
    load_cols<VW>(scale, saved + 2 * C, c0, C);
    load_cols<VW>(shift, saved + 3 * C, c0, C);
    load_cols<VW>(al, alpha, c0, C);
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    double s1[VW], s2[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) s1[k] = s2[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnroll) {
        F<VW> g[kUnroll], xv[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
#pragma unroll
            for (int k = 0; k < VW; ++k) g[u].a[k] = xv[u].a[k] = 0.f;
            if (ok && rr < N) {
                g[u].load(dy + rr * lddy + c0);
                xv[u].load(x + rr * ldx + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
            if (!ok || rr >= N) continue;
            bwd_g<VW>(g[u].a, xv[u].a, scale, shift, act, drop, rr, c0);
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                const float xhat = (xv[u].a[k] - al[k] * mu[k]) * rstd[k];
                s1[k] += (double)g[u].a[k];
                s2[k] += (double)g[u].a[k] * (double)xhat;
            }
        }
    }
    block_reduce_store<VW>(s1, s2, lds, tc, tr, TC, rpb, partial, c0, C);
}

// finalize backward: parameter grads + coefficients of dx = A*g + Bx*x + K
__global__ __launch_bounds__(kBlock) void gn_finalize_bwd_kernel(const double* __restrict__ partial, int nblk, int C,
                                                                 int64_t N, const float* __restrict__ gamma,
                                                                 const float* __restrict__ alpha,
                                                                 const float* __restrict__ saved,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ dalpha, int accumulate,
                                                                 float* __restrict__ coef) {
    __shared__ double lds[kBlock * 2];
    gn_finalize_bwd_block(blockIdx.x, partial, nblk, C, N, gamma, alpha, saved, dgamma, dbeta, dalpha, accumulate, coef,
                          lds);
}

template <int VW>
__global__ __launch_bounds__(kBlock) void gn_bwd_apply_kernel(const float* __restrict__ dy, int64_t lddy,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              float* __restrict__ dx, int64_t lddx,
                                                              const float* __restrict__ addend, int64_t ldadd,
                                                              int64_t N, int C, int tc_log2,
                                                              const float* __restrict__ saved,
                                                              const float* __restrict__ coef, int act, Drop drop,
                                                              const uint64_t* __restrict__ rng_state) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    if (c0 >= C) return;
    float scale[VW], shift[VW], A[VW], Bx[VW], K[VW];
    load_cols<VW>(scale, saved + 2 * C, c0, C);
    load_cols<VW>(shift, saved + 3 * C, c0, C);
    load_cols<VW>(A, coef, c0, C);
    load_cols<VW>(Bx, coef + C, c0, C);
    load_cols<VW>(K, coef + 2 * C, c0, C);
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * kUnroll) {
        F<VW> g[kUnroll], xv[kUnroll], ad[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) {
                g[u].load(dy + rr * lddy + c0);
                xv[u].load(x + rr * ldx + c0);
                if (addend) ad[u].load(addend + rr * ldadd + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            bwd_g<VW>(g[u].a, xv[u].a, scale, shift, act, drop, rr, c0);
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                g[u].a[k] = fmaf(A[k], g[u].a[k], fmaf(Bx[k], xv[u].a[k], K[k]));
                if (addend) g[u].a[k] += ad[u].a[k];
            }
            g[u].store(dx + rr * lddx + c0);
        }
    }
}

// Backward apply with the finalize folded in: the two column sums come from the exact accumulators the data-gradient
// epilogues added to (gn_acc.h); every workgroup folds the replicas and derives the coefficients of ALL columns in its
// prologue (C <= 128), workgroup 0 also writes the parameter gradients — no finalize launch.  The thread's first batch of
// rows is requested BEFORE the fold so that both share one memory round trip (a barrier drains outstanding loads);
// CC = 64: the barrier-free fold by four lanes per column, CC = 0: any C <= 128 through LDS.
template <int VW, int CC, int UN = kUnroll>
__global__ __launch_bounds__(kBlock) void gn_bwd_apply_acc_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                  const float* __restrict__ x, int64_t ldx,
                                                                  float* __restrict__ dx, int64_t lddx,
                                                                  const float* __restrict__ addend, int64_t ldadd,
                                                                  int64_t N, int C, int tc_log2,
                                                                  const float* __restrict__ saved,
                                                                  const long long* __restrict__ acc, int n_rep,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ alpha,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dalpha, int accumulate, int act,
                                                                  Drop drop, const uint64_t* __restrict__ rng_state) {
    __shared__ double sums[kBlock];
    __shared__ float coef_s[3 * (kBlock / 2)];
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = tc * VW;
    const bool col_ok = c0 < C;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    const int64_t r_first = (int64_t)blockIdx.x * rpb + tr;
    F<VW> g[UN], xv[UN], ad[UN];
    auto load_batch = [&](int64_t r) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t rr = r + u * stride;
            if (col_ok && rr < N) {
                g[u].load(dy + rr * lddy + c0);
                xv[u].load(x + rr * ldx + c0);
                if (addend) ad[u].load(addend + rr * ldadd + c0);
            }
        }
    };
    load_batch(r_first);
    auto finish_col = [&](int c, double s1, double s2) __attribute__((always_inline)) {
        float A, Bx, K, da;
        gn_bwd_coeffs(s1, s2, (double)N, gamma[c], alpha[c], saved[c], saved[C + c], A, Bx, K, da);
        coef_s[c] = A;
        coef_s[C + c] = Bx;
        coef_s[2 * C + c] = K;
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
            if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + da;
        }
    };
    if (CC > 0) {
        double s1, s2;
        int col;
        if (gn_acc_col_sums<(CC > 0 ? CC : 64), kBlock>(acc, n_rep, kAccScaleBwd, col, s1, s2)) finish_col(col, s1, s2);
    } else {
        gn_acc_fold(acc, C, 1, sums, kAccScaleBwd, n_rep);
        if ((int)threadIdx.x < C) finish_col(threadIdx.x, sums[threadIdx.x], sums[C + threadIdx.x]);
    }
    __syncthreads();
    if (!col_ok) return;
    float scale[VW], shift[VW], A[VW], Bx[VW], K[VW];
    load_cols<VW>(scale, saved + 2 * C, c0, C);
    load_cols<VW>(shift, saved + 3 * C, c0, C);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        A[k] = coef_s[c0 + k];
        Bx[k] = coef_s[C + c0 + k];
        K[k] = coef_s[2 * C + c0 + k];
    }
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    for (int64_t r = r_first; r < N; r += stride * UN) {
        if (r != r_first) load_batch(r);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            bwd_g<VW>(g[u].a, xv[u].a, scale, shift, act, drop, rr, c0);
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                g[u].a[k] = fmaf(A[k], g[u].a[k], fmaf(Bx[k], xv[u].a[k], K[k]));
                if (addend) g[u].a[k] += ad[u].a[k];
            }
            g[u].store(dx + rr * lddx + c0);
        }
    }
}

__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1; }

// The keep-scales (0 or 1 / (1 - p)) a dropout with this call id draws under the CURRENT (seed, step) words for an
// [N, C] tensor — written out, so that a checker can hand the very same masks to a reference implementation.
__global__ __launch_bounds__(kBlock) void dropout_scales_kernel(Drop drop, const uint64_t* __restrict__ rng_state, int64_t N,
                                                                int C, float* __restrict__ out) {
    drop.seed = rng_state[0];
    drop.step = rng_state[1];
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < N * C; k += (int64_t)gridDim.x * kBlock) {
        float ds[1];
        drop_scales<1>(drop, k / C, (int)(k % C), ds);
        out[k] = ds[0];
    }
}

static unsigned apply_blocks(int64_t n_rows, const Tiling& t) {
    int64_t b = ceil_div(n_rows, (int64_t)t.rpb * kUnroll);
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (unsigned)b;
}

}  // namespace glass

using namespace glass;

extern "C" int64_t glass_graphnorm_ws_bytes(int64_t n_rows, int64_t C) {
    (void)n_rows;
    return ws_partials_bytes(C) + 4 * C * (int64_t)sizeof(float);
}

extern "C" int glass_graphnorm_fwd_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int64_t C,
                                       const float* gamma, const float* beta, const float* alpha, float eps,
                                       float* saved, int act, float p_drop, const uint64_t* rng_state,
                                       uint64_t call_id, void* ws, void* stream) {
    GLASS_REQUIRE(x && y && gamma && beta && alpha && saved && ws, "graphnorm_fwd: null pointer");
    GLASS_REQUIRE(n_rows > 0 && C > 0 && ldx >= C && ldy >= C, "graphnorm_fwd: bad sizes N=%lld C=%lld",
                  (long long)n_rows, (long long)C);
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "graphnorm_fwd: bad dropout args");
    GLASS_REQUIRE(act_code_ok(act), "graphnorm_fwd: bad act %d", act);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y);
    const Tiling t = make_tiling(C, vec);
    const int nblk = stat_blocks(n_rows, t);
    double* partial = (double*)ws;
    const Drop drop = make_drop(p_drop, call_id, C);
    dim3 gs(nblk, t.ctiles), ga(apply_blocks(n_rows, t), t.ctiles);
    if (vec) {
        hipLaunchKernelGGL(gn_stats_kernel<4>, gs, dim3(kBlock), 0, st, x, ldx, n_rows, (int)C, t.tc_log2, partial, 0);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<1>, gs, dim3(kBlock), 0, st, x, ldx, n_rows, (int)C, t.tc_log2, partial, 0);
    }
    hipLaunchKernelGGL(gn_finalize_src_kernel, dim3((unsigned)ceil_div(C, kFinCols)), dim3(kBlock), 0, st,
                       StatSrc{{partial}}, nblk, (int)C, (int)C, n_rows, gamma, beta, alpha, eps, saved);
    if (vec) {
        hipLaunchKernelGGL(gn_apply_kernel<4>, ga, dim3(kBlock), 0, st, x, ldx, y, ldy, n_rows, (int)C, t.tc_log2,
                           saved, act, drop, rng_state);
    } else {
        hipLaunchKernelGGL(gn_apply_kernel<1>, ga, dim3(kBlock), 0, st, x, ldx, y, ldy, n_rows, (int)C, t.tc_log2,
                           saved, act, drop, rng_state);
    }
    return launch_status("glass_graphnorm_fwd_f32");
}

extern "C" int glass_graphnorm_stats_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t C, const float* gamma,
                                         const float* beta, const float* alpha, float eps, float* saved, void* ws,
                                         void* stream) {
    GLASS_REQUIRE(x && gamma && beta && alpha && saved && ws, "graphnorm_stats: null pointer");
    GLASS_REQUIRE(n_rows > 0 && C > 0 && ldx >= C, "graphnorm_stats: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && aligned16(x);
    const Tiling t = make_tiling(C, vec);
    const int nblk = stat_blocks(n_rows, t);
    double* partial = (double*)ws;
    dim3 gs(nblk, t.ctiles);
    if (vec) {
        hipLaunchKernelGGL(gn_stats_kernel<4>, gs, dim3(kBlock), 0, st, x, ldx, n_rows, (int)C, t.tc_log2, partial, 0);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<1>, gs, dim3(kBlock), 0, st, x, ldx, n_rows, (int)C, t.tc_log2, partial, 0);
    }
    hipLaunchKernelGGL(gn_finalize_src_kernel, dim3((unsigned)ceil_div(C, kFinCols)), dim3(kBlock), 0, st,
                       StatSrc{{partial}}, nblk, (int)C, (int)C, n_rows, gamma, beta, alpha, eps, saved);
    return launch_status("glass_graphnorm_stats_f32");
}

// The statistics pass alone, into exact accumulators (gn_acc.h; zeroed by glass_step_prologue_f32): the consumer kernel
// derives the coefficients (glass_gn_src), no finalize launch.
extern "C" int glass_graphnorm_stats_exact_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t C, int64_t* acc,
                                               int n_rep, void* stream) {
    GLASS_REQUIRE(x && acc && n_rows > 0 && C > 0 && ldx >= C && n_rep >= 1 && n_rep <= kAccRep, "graphnorm_stats_exact: bad arguments");
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && aligned16(x);
    if (vec && C == 64) {
        // one round of <= 5 rows per thread while the graph fits 256 workgroups of 80 rows; taller tiles beyond
        int nst = (int)ceil_div(ceil_div(n_rows, (int64_t)256), (int64_t)16);
        if (nst < 5) nst = n_rows >= 80 * 64 ? 5 : (nst < 1 ? 1 : nst);
        hipLaunchKernelGGL(gn_stats64_exact_kernel, dim3((unsigned)ceil_div(n_rows, (int64_t)16 * nst)), dim3(kBlock), 0, (hipStream_t)stream,
                           x, ldx, n_rows, nst, reinterpret_cast<long long*>(acc), n_rep);
        return launch_status("glass_graphnorm_stats_exact_f32");
    }
    const Tiling t = make_tiling(C, vec);
    // (fewer, longer workgroups to thin out the adds per replica were slower: 8.3 / 12.7 us with half / a quarter of them)
    dim3 gs(stat_blocks(n_rows, t), t.ctiles);
    if (vec) {
        hipLaunchKernelGGL(gn_stats_kernel<4>, gs, dim3(kBlock), 0, (hipStream_t)stream, x, ldx, n_rows, (int)C, t.tc_log2,
                           reinterpret_cast<double*>(acc), n_rep);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<1>, gs, dim3(kBlock), 0, (hipStream_t)stream, x, ldx, n_rows, (int)C, t.tc_log2,
                           reinterpret_cast<double*>(acc), n_rep);
    }
    return launch_status("glass_graphnorm_stats_exact_f32");
}

extern "C" int glass_graphnorm_finalize_f32(const double* const* partials, int64_t n_src, int64_t nblk, int64_t C_each,
                                            int64_t n_rows, const float* gamma, const float* beta, const float* alpha,
                                            float eps, float* saved, void* stream) {
    GLASS_REQUIRE(partials && gamma && beta && alpha && saved, "graphnorm_finalize: null pointer");
    GLASS_REQUIRE(n_src > 0 && n_src <= kMaxStatSrc && nblk > 0 && nblk < (1ll << 31) && C_each > 0 && n_rows > 0,
                  "graphnorm_finalize: bad sizes (at most %d sources)", kMaxStatSrc);
    StatSrc src;
    for (int k = 0; k < kMaxStatSrc; ++k) src.p[k] = k < n_src ? partials[k] : nullptr;
    for (int k = 0; k < n_src; ++k) GLASS_REQUIRE(src.p[k], "graphnorm_finalize: null source %d", k);
    const int64_t C = n_src * C_each;
    hipLaunchKernelGGL(gn_finalize_src_kernel, dim3((unsigned)ceil_div(C, kFinCols)), dim3(kBlock), 0, (hipStream_t)stream, src,
                       (int)nblk, (int)C_each, (int)C, n_rows, gamma, beta, alpha, eps, saved);
    return launch_status("glass_graphnorm_finalize_f32");
}

extern "C" int glass_graphnorm_apply_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int64_t C,
                                         const float* saved, int act, float p_drop, const uint64_t* rng_state,
                                         uint64_t call_id, void* stream) {
    GLASS_REQUIRE(x && y && saved, "graphnorm_apply: null pointer");
    GLASS_REQUIRE(n_rows > 0 && C > 0 && ldx >= C && ldy >= C, "graphnorm_apply: bad sizes");
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "graphnorm_apply: bad dropout args");
    GLASS_REQUIRE(act_code_ok(act), "graphnorm_apply: bad act %d", act);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y);
    const Tiling t = make_tiling(C, vec);
    const Drop drop = make_drop(p_drop, call_id, C);
    dim3 ga(apply_blocks(n_rows, t), t.ctiles);
    if (vec) {
        hipLaunchKernelGGL(gn_apply_kernel<4>, ga, dim3(kBlock), 0, st, x, ldx, y, ldy, n_rows, (int)C, t.tc_log2, saved,
                           act, drop, rng_state);
    } else {
        hipLaunchKernelGGL(gn_apply_kernel<1>, ga, dim3(kBlock), 0, st, x, ldx, y, ldy, n_rows, (int)C, t.tc_log2, saved,
                           act, drop, rng_state);
    }
    return launch_status("glass_graphnorm_apply_f32");
}

extern "C" int glass_graphnorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx,
                                       int64_t lddx, const float* addend, int64_t ldadd, int64_t n_rows, int64_t C,
                                       const float* gamma,
                                       const float* alpha, const float* saved, float* dgamma, float* dbeta,
                                       float* dalpha, int accumulate, int act, float p_drop,
                                       const uint64_t* rng_state, uint64_t call_id, void* ws, void* stream) {
    GLASS_REQUIRE(dy && x && dx && gamma && alpha && saved && ws, "graphnorm_bwd: null pointer");
    GLASS_REQUIRE(n_rows > 0 && C > 0 && lddy >= C && ldx >= C && lddx >= C && (!addend || ldadd >= C),
                  "graphnorm_bwd: bad sizes");
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "graphnorm_bwd: bad dropout args");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && aligned16(dy) && aligned16(x) &&
                     aligned16(dx) && (!addend || (ldadd % 4 == 0 && aligned16(addend)));
    const Tiling t = make_tiling(C, vec);
    const int nblk = stat_blocks(n_rows, t);
    double* partial = (double*)ws;
    float* coef = (float*)((char*)ws + ws_partials_bytes(C));
    const Drop drop = make_drop(p_drop, call_id, C);
    dim3 gs(nblk, t.ctiles), ga(apply_blocks(n_rows, t), t.ctiles);
    if (vec) {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<4>, gs, dim3(kBlock), 0, st, dy, lddy, x, ldx, n_rows, (int)C,
                           t.tc_log2, saved, alpha, act, drop, rng_state, partial);
    } else {
        hipLaunchKernelGGL(gn_bwd_stats_kernel<1>, gs, dim3(kBlock), 0, st, dy, lddy, x, ldx, n_rows, (int)C,
                           t.tc_log2, saved, alpha, act, drop, rng_state, partial);
    }
    hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3((unsigned)ceil_div(C, kFinCols)), dim3(kBlock), 0, st, partial, nblk,
                       (int)C, n_rows, gamma, alpha, saved, dgamma, dbeta, dalpha, accumulate, coef);
    if (vec) {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<4>, ga, dim3(kBlock), 0, st, dy, lddy, x, ldx, dx, lddx, addend, ldadd,
                           n_rows, (int)C, t.tc_log2, saved, coef, act, drop, rng_state);
    } else {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, ga, dim3(kBlock), 0, st, dy, lddy, x, ldx, dx, lddx, addend, ldadd,
                           n_rows, (int)C, t.tc_log2, saved, coef, act, drop, rng_state);
    }
    return launch_status("glass_graphnorm_bwd_f32");
}

extern "C" int glass_graphnorm_bwd_from_stats_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx,
                                                  int64_t lddx, const float* addend, int64_t ldadd, int64_t n_rows,
                                                  int64_t C, const float* gamma, const float* alpha, const float* saved,
                                                  const double* partial, int64_t nblk, float* dgamma, float* dbeta,
                                                  float* dalpha, int accumulate, int act, float p_drop,
                                                  const uint64_t* rng_state, uint64_t call_id, void* ws, void* stream) {
    GLASS_REQUIRE(dy && x && dx && gamma && alpha && saved && partial && ws, "graphnorm_bwd_from_stats: null pointer");
    GLASS_REQUIRE(n_rows > 0 && C > 0 && nblk != 0 && nblk < (1ll << 31) && lddy >= C && ldx >= C && lddx >= C &&
                      (!addend || ldadd >= C),
                  "graphnorm_bwd_from_stats: bad sizes");
    if (nblk < 0) {
        // `partial` = the exact accumulators of gn_acc.h (what the data-gradient epilogues added to with gn_exact): finalize
        // and apply as ONE launch
        const bool vec4 = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && aligned16(dy) && aligned16(x) &&
                          aligned16(dx) && (!addend || (ldadd % 4 == 0 && aligned16(addend)));
        GLASS_REQUIRE(vec4 && (C == 16 || C == 32 || C == 64 || C == 128) && aligned16(partial),
                      "graphnorm_bwd_from_stats: the exact form serves C = 16 .. 128 (powers of two), 16-B aligned operands");
        GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "graphnorm_bwd_from_stats: bad dropout args");
        const Tiling t4 = make_tiling(C, true);
        const int n_rep = (int)-nblk;  // replicas the producers used
        GLASS_REQUIRE(n_rep <= kAccRep && (n_rep == 1 || n_rep % 2 == 0), "graphnorm_bwd_from_stats: nblk = -(replicas), 2 .. 16");
        if (C == 64 && n_rep >= 2)
            // 80-row tiles (five rows per thread in one round of loads): about one workgroup per CU at ppi_bp-shape
            hipLaunchKernelGGL((gn_bwd_apply_acc_kernel<4, 64, 5>), dim3((unsigned)ceil_div(n_rows, (int64_t)t4.rpb * 5)), dim3(kBlock), 0, (hipStream_t)stream,
                               dy, lddy, x, ldx, dx, lddx, addend, ldadd, n_rows, (int)C, t4.tc_log2, saved, (const long long*)partial,
                               n_rep, gamma, alpha, dgamma, dbeta, dalpha, accumulate, act, make_drop(p_drop, call_id, C), rng_state);
        else
            hipLaunchKernelGGL((gn_bwd_apply_acc_kernel<4, 0>), dim3(apply_blocks(n_rows, t4)), dim3(kBlock), 0, (hipStream_t)stream,
                               dy, lddy, x, ldx, dx, lddx, addend, ldadd, n_rows, (int)C, t4.tc_log2, saved, (const long long*)partial,
                               n_rep, gamma, alpha, dgamma, dbeta, dalpha, accumulate, act, make_drop(p_drop, call_id, C), rng_state);
        return launch_status("glass_graphnorm_bwd_from_stats_f32 (exact)");
    }
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "graphnorm_bwd_from_stats: bad dropout args");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && aligned16(dy) && aligned16(x) &&
                     aligned16(dx) && (!addend || (ldadd % 4 == 0 && aligned16(addend)));
    const Tiling t = make_tiling(C, vec);
    float* coef = (float*)((char*)ws + ws_partials_bytes(C));
    const Drop drop = make_drop(p_drop, call_id, C);
    dim3 ga(apply_blocks(n_rows, t), t.ctiles);
    hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3((unsigned)ceil_div(C, kFinCols)), dim3(kBlock), 0, st, partial, (int)nblk,
                       (int)C, n_rows, gamma, alpha, saved, dgamma, dbeta, dalpha, accumulate, coef);
    if (vec) {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<4>, ga, dim3(kBlock), 0, st, dy, lddy, x, ldx, dx, lddx, addend, ldadd,
                           n_rows, (int)C, t.tc_log2, saved, coef, act, drop, rng_state);
    } else {
        hipLaunchKernelGGL(gn_bwd_apply_kernel<1>, ga, dim3(kBlock), 0, st, dy, lddy, x, ldx, dx, lddx, addend, ldadd,
                           n_rows, (int)C, t.tc_log2, saved, coef, act, drop, rng_state);
    }
    return launch_status("glass_graphnorm_bwd_from_stats_f32");
}

extern "C" int glass_rng_advance(uint64_t* rng_state, void* stream) {
    GLASS_REQUIRE(rng_state, "rng_advance: null pointer");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    return launch_status("glass_rng_advance");
}

extern "C" int glass_dropout_scales_f32(const uint64_t* rng_state, uint64_t call_id, float p_drop, int64_t n_rows, int64_t C,
                                        float* out, void* stream) {
    GLASS_REQUIRE(rng_state && out && n_rows > 0 && C > 0 && p_drop > 0.f && p_drop < 1.f, "dropout_scales: bad arguments");
    int64_t blocks = ceil_div(n_rows * C, kBlock);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_scales_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream,
                       make_drop(p_drop, call_id, C), rng_state, n_rows, (int)C, out);
    return launch_status("glass_dropout_scales_f32");
}
