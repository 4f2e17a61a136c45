// K5n: the dense half of GLASSConv for NARROW layers (hidden <= 32): the widths the reference's own YAMLs use for the
// shipped synthetic sets — 8 (config/density.yml, cut_ratio.yml), 17 (component.yml), 20 (coreness.yml).
// reference impl/models.py:158-162 (trans_fns + ELU + mix), 167-173 (cat + comb_fns + mix) and their backward.
//
// At these widths a whole weight pair is a few KiB and a row a few dozen bytes: nothing here is a matrix-core problem
// (a 16-row MFMA tile would be mostly padding), and the graphs that use them are small (5-17 k nodes), so a launch is
// its latency chain.  One THREAD owns one row: the pair's weights sit in LDS (every lane reads the same word: broadcast),
// the row lives in registers, plain fp32 fma chains in k order.  Same entry points and the same fusions as the MFMA
// families (dense.hip / dense_tiled.hip): GraphNorm apply (+ ELU + dropout) while loading with the normalised operand
// written to a side output, the embedding lookup inside the load (xa_index), the GraphNorm statistics of the output in
// the epilogue, dZ synthesised on the fly, backward GraphNorm column sums / addend / dropout mask in the data-gradient
// epilogue — so the step program (glass_amd/stack.py) runs unchanged and no library GEMM is left on the path.
// Weights are read as they are (row-major [2H][K]): no packed images (glass_dual_linear_layout(H) == 2).
#include "dense_common.h"
#include "wgrad_common.h"

namespace glass {

constexpr int kNarrowMaxH = 32;
constexpr int kNarrowRows = 256;  // rows per workgroup = rows per statistics partial

bool narrow_shape_ok(int64_t H) { return H >= 1 && H <= kNarrowMaxH; }
int narrow_rows() { return kNarrowRows; }

// Sum (a, b) of column c over the 256 rows of the workgroup: waves by shuffles, the four waves through LDS in fixed order.
// red: [4][2 * HP] doubles.  Call for every column, then narrow_cols_store after a barrier.
template <int HP>
__device__ __forceinline__ void narrow_col_partial(double a, double b, int c, double* red) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        a += __shfl_xor(a, s);
        b += __shfl_xor(b, s);
    }
    if ((threadIdx.x & 63) == 0) {
        red[((threadIdx.x >> 6) * HP + c) * 2] = a;
        red[((threadIdx.x >> 6) * HP + c) * 2 + 1] = b;
    }
}

template <int HP>
__device__ __forceinline__ void narrow_cols_store(const double* red, int H, double* __restrict__ dst /* [2][H] of this block */) {
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += kNarrowRows) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            a += red[(w * HP + c) * 2];
            b += red[(w * HP + c) * 2 + 1];
        }
        dst[c] = a;
        dst[H + c] = b;
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------
template <int HP, bool COMB>
__global__ __launch_bounds__(kNarrowRows) void narrow_fwd_kernel(const float* __restrict__ xa, int64_t lda,
                                                                 const float* __restrict__ xb, int64_t ldb,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const uint8_t* __restrict__ mask, float zr, float omz,
                                                                 int act, float* __restrict__ T, int64_t ldt,
                                                                 float* __restrict__ out, int64_t ldo, int64_t N, int H,
                                                                 double* __restrict__ stats, GnPrologue pro,
                                                                 const int64_t* __restrict__ xa_index, int xa_rows) {
    constexpr int KP = COMB ? 2 * HP : HP;
    __shared__ float w_s[2 * HP * KP];
    __shared__ float b_s[2 * HP];
    __shared__ double red[4 * 2 * HP];
    const int K = COMB ? 2 * H : H;
    for (int idx = threadIdx.x; idx < 2 * H * K; idx += kNarrowRows) w_s[idx] = W[idx];
    for (int idx = threadIdx.x; idx < 2 * H; idx += kNarrowRows) b_s[idx] = bias[idx];
    const int64_t row = (int64_t)blockIdx.x * kNarrowRows + threadIdx.x;
    const bool ok = row < N;
    float x[KP];
    int64_t src = ok ? row : 0;
    if (!COMB && xa_index) {
        src = ok ? xa_index[row] : 0;
        src = src < 0 ? 0 : (src >= xa_rows ? xa_rows - 1 : src);
    }
#pragma unroll
    for (int k = 0; k < HP; ++k) x[k] = (ok && k < H) ? xa[src * lda + k] : 0.f;
    if (pro.saved) {  // xa is the input of a GraphNorm with final statistics: normalise (+ ELU + dropout) while loading
        Drop drop = pro.drop;
        if (drop.p > 0.f) {
            drop.seed = pro.rng_state[0];
            drop.step = pro.rng_state[1];
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            if (k >= H) continue;
            float h = fmaf(x[k], pro.saved[2 * pro.C + k], pro.saved[3 * pro.C + k]);
            h = act_exact(pro.act, h);
            if (drop.p > 0.f) {
                float ds[1];
                drop_scales<1>(drop, row, k, ds);
                h *= ds[0];
            }
            x[k] = ok ? h : 0.f;
            if (pro.side && ok) pro.side[row * pro.lds + k] = h;
        }
    }
    if (COMB) {
#pragma unroll
        for (int k = 0; k < HP; ++k) x[HP + k] = (ok && k < H) ? xb[row * ldb + k] : 0.f;
    }
    __syncthreads();
    const bool lab = ok && mask[row] != 0;
    const float w1 = lab ? zr : omz, w0 = lab ? omz : zr;
#pragma unroll
    for (int o = 0; o < HP; ++o) {
        if (o >= H) continue;
        float z1 = b_s[o], z0 = b_s[H + o];
        const float* r1 = w_s + o * K;
        const float* r0 = w_s + (H + o) * K;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            if (k >= H) continue;
            z1 = fmaf(x[k], r1[k], z1);
            z0 = fmaf(x[k], r0[k], z0);
        }
        if (COMB) {
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                if (k >= H) continue;
                z1 = fmaf(x[HP + k], r1[H + k], z1);
                z0 = fmaf(x[HP + k], r0[H + k], z0);
            }
        }
        if (T && ok) {
            T[row * ldt + o] = z1;
            T[row * ldt + H + o] = z0;
        }
        float a1 = z1, a0 = z0;
        a1 = act_fast(act, a1), a0 = act_fast(act, a0);
        const float v = ok ? w1 * a1 + w0 * a0 : 0.f;
        if (ok) out[row * ldo + o] = v;
        if (stats) narrow_col_partial<HP>((double)v, (double)v * (double)v, o, red);
    }
    if (stats) narrow_cols_store<HP>(red, H, stats + (size_t)blockIdx.x * 2 * H);
}

// ---- backward data gradient ---------------------------------------------------------------------------------------
// out[N, n_out] = dZ[N, 2H] @ W[2H, n_out] (+ addend) (* dropout mask); dZ[n, o] = coef(n, o < H) * dsrc[n, o mod H] * act'(T[n, o])
template <int HP>
__global__ __launch_bounds__(kNarrowRows) void narrow_dgrad_kernel(const float* __restrict__ dsrc, int64_t ldd,
                                                                   const float* __restrict__ T, int64_t ldt,
                                                                   const uint8_t* __restrict__ mask, float zr, float omz,
                                                                   int act, const float* __restrict__ W, int n_out,
                                                                   const float* __restrict__ addend, int64_t ldadd, Drop drop,
                                                                   const uint64_t* __restrict__ rng_state,
                                                                   float* __restrict__ out, int64_t ldo, int64_t N, int H,
                                                                   GnBwdStats gs) {
    __shared__ float w_s[2 * HP * 2 * HP];
    __shared__ double red[4 * 2 * HP];
    for (int idx = threadIdx.x; idx < 2 * H * n_out; idx += kNarrowRows) w_s[idx] = W[idx];
    const int64_t row = (int64_t)blockIdx.x * kNarrowRows + threadIdx.x;
    const bool ok = row < N;
    const bool lab = ok && mask[row] != 0;
    const float c1 = lab ? zr : omz, c0 = lab ? omz : zr;
    float dz[2 * HP];
#pragma unroll
    for (int o = 0; o < HP; ++o) {
        float d = (ok && o < H) ? dsrc[row * ldd + o] : 0.f;
        float g1 = d * c1, g0 = d * c0;
        if (act != GLASS_ACT_NONE && ok && o < H) {
            g1 *= act_grad(act, T[row * ldt + o]);
            g0 *= act_grad(act, T[row * ldt + H + o]);
        }
        dz[o] = g1;
        dz[HP + o] = g0;
    }
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    if (gs.partial && gs.drop.p > 0.f) {
        gs.drop.seed = rng_state[0];
        gs.drop.step = rng_state[1];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2 * HP; ++j) {
        if (j >= n_out) continue;
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < HP; ++o) {
            if (o >= H) continue;
            acc = fmaf(dz[o], w_s[o * n_out + j], acc);
        }
#pragma unroll
        for (int o = 0; o < HP; ++o) {
            if (o >= H) continue;
            acc = fmaf(dz[HP + o], w_s[(H + o) * n_out + j], acc);
        }
        if (addend && ok) acc += addend[row * ldadd + j];
        if (drop.p > 0.f) {  // gradient w.r.t. the pre-dropout tensor: same mask as the forward drew
            float ds[1];
            drop_scales<1>(drop, row, j, ds);
            acc *= ds[0];
        }
        if (ok) out[row * ldo + j] = acc;
        if (gs.partial && j < H) {  // the first H columns are the gradient of a GraphNorm OUTPUT: its backward column sums
            float gp = 0.f, xhat = 0.f;
            if (ok) {
                const float xv = gs.x[row * gs.ldx + j];
                gp = acc;
                if (gs.drop.p > 0.f) {
                    float ds[1];
                    drop_scales<1>(gs.drop, row, j, ds);
                    gp *= ds[0];
                }
                if (gs.act != GLASS_ACT_NONE) gp *= act_grad(gs.act, fmaf(xv, gs.saved[2 * H + j], gs.saved[3 * H + j]));
                xhat = (xv - gs.alpha[j] * gs.saved[j]) * gs.saved[H + j];
            }
            narrow_col_partial<HP>((double)gp, (double)gp * (double)xhat, j, red);
        }
    }
    if (gs.partial) narrow_cols_store<HP>(red, H, gs.partial + (size_t)blockIdx.x * 2 * H);
}

// ---- weight gradient: per-slab partial sums ---------------------------------------------------------------------------
// dW[o, i] = sum_n dZ[n, o] * X[n, i], db[o] = sum_n dZ[n, o], X = [X | X2]; O = 2H <= 64, I = H or 2H <= 64.
// One workgroup per slab of 64 rows (the graphs are small: the slabs are what fills the chip): the slab's dZ (synthesised
// while staging) and X rows go to LDS once, every thread owns up to 16 of the O*I outputs and walks the 64 staged rows in
// order; slab partials [n_slabs][stride] (stride = O*I + O rounded up to 4) are summed in slab order by the batched reduce
// (linear.hip) — deterministic.
constexpr int kNarrowSlab = 64;

__global__ __launch_bounds__(kNarrowRows) void narrow_wgrad_kernel(WgradSynth sy, const float* __restrict__ X, int64_t ldx,
                                                                   int64_t N, int O, int I, int stride,
                                                                   float* __restrict__ part) {
    __shared__ float dz_s[kNarrowSlab][2 * kNarrowMaxH];
    __shared__ float x_s[kNarrowSlab][2 * kNarrowMaxH];
    const int H = sy.H;
    const int64_t r0 = (int64_t)blockIdx.x * kNarrowSlab;
    const int nrow = (int)min((int64_t)kNarrowSlab, N - r0);
    for (int idx = threadIdx.x; idx < kNarrowSlab * O; idx += kNarrowRows) {
        const int r = idx / O, o = idx % O;
        float g = 0.f;
        if (r < nrow) {
            const int64_t n = r0 + r;
            const bool first = o < H;
            const float cf = ((sy.mask[n] != 0) == first) ? sy.zr : sy.omz;
            g = sy.dsrc[n * sy.ldd + (first ? o : o - H)] * cf;
            if (sy.act != GLASS_ACT_NONE) g *= act_grad(sy.act, sy.T[n * sy.ldt + o]);
        }
        dz_s[r][o] = g;
    }
    for (int idx = threadIdx.x; idx < kNarrowSlab * I; idx += kNarrowRows) {
        const int r = idx / I, i = idx % I;
        float v = 0.f;
        if (r < nrow) {
            const int64_t n = r0 + r;
            v = (i < H || sy.X2 == nullptr) ? X[n * ldx + i] : sy.X2[n * sy.ldx2 + (i - H)];
        }
        x_s[r][i] = v;
    }
    __syncthreads();
    float* p = part + (int64_t)blockIdx.x * stride;
    for (int idx = threadIdx.x; idx < O * I; idx += kNarrowRows) {
        const int o = idx / I, i = idx % I;
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < kNarrowSlab; ++r) acc = fmaf(dz_s[r][o], x_s[r][i], acc);
        p[idx] = acc;
    }
    if ((int)threadIdx.x < O) {
        float b = 0.f;
        for (int r = 0; r < kNarrowSlab; ++r) b += dz_s[r][threadIdx.x];
        p[O * I + threadIdx.x] = b;
    }
}

// partial layout of the narrow weight gradient (also read by linear.hip's batched reduce)
void narrow_wgrad_geom(int64_t N, int64_t O, int64_t I, int* n_slabs, int* stride) {
    *n_slabs = (int)ceil_div(N, kNarrowSlab);
    *stride = (int)(ceil_div(O * I + O, 4) * 4);
}

#define GLASS_NARROW_DISPATCH(H, CALL)      \
    if (H <= 8) { CALL(8) }                 \
    else if (H <= 12) { CALL(12) }          \
    else if (H <= 16) { CALL(16) }          \
    else if (H <= 20) { CALL(20) }          \
    else if (H <= 24) { CALL(24) }          \
    else if (H <= 28) { CALL(28) }          \
    else { CALL(32) }

int launch_narrow_fwd(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* W, const float* bias,
                      const uint8_t* mask, float zr, float omz, int act, float* T, int64_t ldt, float* out, int64_t ldo,
                      int64_t N, int64_t H, double* stats, const GnPrologue& pro, const int64_t* xa_index, int64_t xa_rows,
                      hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(N, kNarrowRows));
#define CALL(HP)                                                                                                              \
    if (xb)                                                                                                                   \
        hipLaunchKernelGGL((narrow_fwd_kernel<HP, true>), grid, dim3(kNarrowRows), 0, st, xa, lda, xb, ldb, W, bias, mask, zr,  \
                           omz, act, T, ldt, out, ldo, N, (int)H, stats, pro, xa_index, (int)xa_rows);                        \
    else                                                                                                                      \
        hipLaunchKernelGGL((narrow_fwd_kernel<HP, false>), grid, dim3(kNarrowRows), 0, st, xa, lda, xb, ldb, W, bias, mask, zr, \
                           omz, act, T, ldt, out, ldo, N, (int)H, stats, pro, xa_index, (int)xa_rows);
    GLASS_NARROW_DISPATCH(H, CALL)
#undef CALL
    return launch_status("glass_dual_linear_fwd_f32 (narrow)");
}

int launch_narrow_dgrad(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, float zr, float omz,
                        int act, const float* W, int64_t n_out, const float* addend, int64_t ldadd, const Drop& drop,
                        const uint64_t* rng_state, float* out, int64_t ldo, int64_t N, int64_t H, const GnBwdStats& gs,
                        hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(N, kNarrowRows));
#define CALL(HP)                                                                                                          \
    hipLaunchKernelGGL((narrow_dgrad_kernel<HP>), grid, dim3(kNarrowRows), 0, st, dsrc, ldd, T, ldt, mask, zr, omz, act, W, \
                       (int)n_out, addend, ldadd, drop, rng_state, out, ldo, N, (int)H, gs);
    GLASS_NARROW_DISPATCH(H, CALL)
#undef CALL
    return launch_status("glass_dual_linear_dgrad_f32 (narrow)");
}

int launch_narrow_wgrad(const WgradSynth& sy, const float* X, int64_t ldx, int64_t N, int64_t O, int64_t I, float* part,
                        hipStream_t st) {
    int n_slabs, stride;
    narrow_wgrad_geom(N, O, I, &n_slabs, &stride);
    hipLaunchKernelGGL(narrow_wgrad_kernel, dim3((unsigned)n_slabs), dim3(kNarrowRows), 0, st, sy, X, ldx, N, (int)O, (int)I,
                       stride, part);
    return launch_status("glass_dual_linear_wgrad_f32 (narrow)");
}

}  // namespace glass
