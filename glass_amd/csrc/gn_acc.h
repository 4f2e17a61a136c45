// Exact cross-workgroup column sums without a reduction launch.
//
// GraphNorm over the whole graph (PyG GraphNorm, batch = None; reference impl/models.py:165,249,257,266,271) needs per-column
// sums over ALL rows between every pair of kernels of the step.  Per-workgroup fp64 partials + a tiny "finalize" launch
// (graphnorm.hip) cost a dependent launch of ~4.7 us each, seven times per step at ppi_bp-shape.  Here the producers add
// their per-workgroup partial sums into a few replicas of FIXED-POINT accumulators with 64-bit INTEGER atomics — integer
// addition is associative, so the result does not depend on the order the workgroups arrive in: bitwise repeatable, which
// float atomics are not — and every consumer workgroup folds the replicas itself in its prologue (one round trip of 16-32
// loads per thread).  Representation of a partial sum v (a double): v * 2^52 = hi * 2^40 + lo, hi = floor(v * 2^12) as int64,
// lo in [0, 2^40): absolute resolution 2^-52 (2.2e-16) per workgroup partial, range |sum| < 2^50 — far inside what fp32
// activations and their squares can reach over 10^6 rows.  A partial that is NaN, infinite or out of that range does not
// saturate into a finite wrong sum: it sets a STICKY POISON BIT — bit 63 of the lo limb, by an integer atomicOr; the lo limbs
// of up to 2^22 in-range adds stay below 2^62, so no carry ever reaches or clears it, and OR commutes with itself like the
// adds do — and every fold turns a poisoned column into NaN, which the consumers' arithmetic then propagates exactly as the
// reference's float sums do (a diverging run stays visible; tests/test_gpu_model.py::test_nonfinite_values_reach_the_outputs).
// Resolution caveat: contributions below 2^-52 (forward) / 2^-64 (backward) of a PARTIAL are truncated — "exact" means the
// integer sum of the partials is exact and order-independent, not that each partial is represented exactly.  Layout: int64
// [kAccRep][2 quantities][C][2 limbs], zeroed once per step by the prologue launch (glass_step_prologue_f32).
#pragma once
#include "common.h"
#include "gn_math.h"

namespace glass {

#ifndef GLASS_ACC_REP
#define GLASS_ACC_REP 16
#endif
// replicas a block holds room for; a producer spreads its adds over the first n_rep of them (workgroup b -> replica
// b % n_rep) and the consumer folds the same n_rep — 16 behind the statistics kernel, whose ~134 workgroups all finish
// together, 4 behind the dense kernels' epilogues, which finish spread over the launch (every consumer workgroup reads
// n_rep * 2 KB at hidden 64: with 16 that was as much L2 traffic as the weight image).  Adds to ONE address queue at the memory-side
// atomic unit (~100 ns each, measured: the 134-workgroup statistics kernel 10.6 / 8.1 / 6.6 us with 4 / 8 / 16 replicas, 5.4 us
// when it writes plain partials); the consumers' fold loads are all in flight at once, so their cost barely moves.
constexpr int kAccRep = GLASS_ACC_REP;

// hi-limb scale of the two uses: forward sums (x, x^2: large, never tiny against eps) keep 2^-52 resolution and 2^50 range;
// backward sums (gradients: small) trade range for resolution — 2^-64 per workgroup partial, |sum| < 2^38
constexpr double kAccScaleFwd = 4096.0;      // 2^12
constexpr double kAccScaleBwd = 16777216.0;  // 2^24

constexpr unsigned long long kAccPoison = 1ull << 63;  // bit 63 of a lo limb: some partial of this column was not representable
// the value of a folded column: NaN when any replica carried the poison bit
__device__ __forceinline__ double gn_acc_value(long long hi, long long lo, bool poisoned, double inv_scale) {
    return poisoned ? __longlong_as_double(0x7ff8000000000000ll) : ((double)hi + (double)lo * (1.0 / 1099511627776.0)) * inv_scale;
}

__device__ __forceinline__ size_t gn_acc_index(int rep, int which, int c, int C) { return (((size_t)rep * 2 + which) * C + c) * 2; }
__host__ __device__ constexpr int64_t gn_acc_words(int64_t C) { return (int64_t)kAccRep * 2 * C * 2; }

__device__ __forceinline__ void gn_acc_add(long long* __restrict__ acc, int rep, int which, int c, int C, double v, double scale) {
    const double sv = v * scale;
    unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + gn_acc_index(rep, which, c, C));
    if (!(fabs(sv) <= 4.0e18)) {  // NaN, +-Inf or beyond the fixed-point range: poison the column (sticky, order-independent)
        atomicOr(p + 1, kAccPoison);
        return;
    }
    const double fl = floor(sv);
    const long long hi = (long long)fl;
    const long long lo = (long long)((sv - fl) * 1099511627776.0);  // 2^40
    atomicAdd(p, (unsigned long long)hi);
    atomicAdd(p + 1, (unsigned long long)lo);
}

// Every thread of the workgroup calls this; afterwards out[which * C + c] (LDS doubles, 2 * C of them, C = n_src * C_each)
// holds the two sums of every column.  acc: n_src consecutive accumulator blocks of C_each columns — block k covers columns
// k * C_each .. (the column blocks of a jumping-knowledge buffer, each summed by the kernel that wrote it).
__device__ __forceinline__ void gn_acc_fold(const long long* __restrict__ acc, int C_each, int n_src, double* out, double scale,
                                            int n_rep = kAccRep) {
    const int C = C_each * n_src;
    const double inv = 1.0 / scale;
    for (int item = threadIdx.x; item < 2 * C; item += blockDim.x) {
        const int which = item / C, c = item - which * C;
        const int k = c / C_each, cl = c - k * C_each;
        const long long* base = acc + (size_t)k * gn_acc_words(C_each);
        long long hi = 0, lo = 0;
        bool bad = false;
#pragma unroll
        for (int r = 0; r < kAccRep; ++r) {
            if (r >= n_rep) break;  // (the producers of this sum used the first n_rep replicas)
            const long long* p = base + gn_acc_index(r, which, cl, C_each);
            hi += p[0];
            lo += p[1];
            bad |= p[1] < 0;
        }
        out[item] = gn_acc_value(hi, lo, bad, inv);
    }
    __syncthreads();
}

// A GraphNorm whose forward sums are still in accumulators: the consumer derives the coefficients itself.
struct GnExactSrc {
    const long long* acc;  // nullptr: the statistics are final in `saved` already
    int n_src;             // accumulator blocks (of C / n_src columns each)
    int n_rep;             // replicas the producers spread their adds over (the first n_rep of a block's kAccRep)
    const float *gamma, *beta, *alpha;
    float eps;
    float* saved_w;        // [4C] mean, rstd, scale, shift: written by workgroup 0 for the backward
};

// Consumer prologue, every thread of the workgroup: scale | shift of all C columns into coef_s (LDS floats [2C]) — folded
// from the accumulators (sums: LDS doubles [2C]), or copied from the final `saved`.  mu_rstd_s (LDS [2C], optional): mean | rstd.
__device__ __forceinline__ void gn_fwd_coef_block(const GnExactSrc& src, const float* __restrict__ saved, int C, int64_t N,
                                                  double* sums, float* coef_s, float* mu_rstd_s) {
    if (src.acc) {
        gn_acc_fold(src.acc, C / src.n_src, src.n_src, sums, kAccScaleFwd, src.n_rep);
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float mu, rstd, scale, shift;
            gn_fwd_coeffs(sums[c], sums[C + c], (double)N, src.gamma[c], src.beta[c], src.alpha[c], src.eps, mu, rstd, scale, shift);
            coef_s[c] = scale;
            coef_s[C + c] = shift;
            if (mu_rstd_s) {
                mu_rstd_s[c] = mu;
                mu_rstd_s[C + c] = rstd;
            }
            if (blockIdx.x == 0 && src.saved_w) {
                src.saved_w[c] = mu;
                src.saved_w[C + c] = rstd;
                src.saved_w[2 * C + c] = scale;
                src.saved_w[3 * C + c] = shift;
            }
        }
    } else {
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
            coef_s[c] = saved[2 * C + c];
            if (mu_rstd_s) mu_rstd_s[c] = saved[c];
        }
    }
    __syncthreads();
}

// The same block in two steps for workgroups of >= 2C threads, so that the caller can put its own independent loads (and
// the barrier that drains them all) between the requests and their use: `issue` only starts the loads — thread t < 2C
// takes item t = (which, c) like gn_acc_fold, threads t < C also the column's parameters — `finish` adds, exchanges
// through LDS and derives the coefficients (two barriers, no memory round trip of its own).
struct GnCoefEarly {
    long long h[kAccRep], l[kAccRep];
    float p0, p1, p2;  // gamma, beta, alpha of column t (exact form) / saved[2C + t], saved[t] (final statistics)
};

__device__ __forceinline__ void gn_coef_early_issue(const GnExactSrc& src, const float* __restrict__ saved, int C, GnCoefEarly& E) {
    const int t = threadIdx.x;
    E.p0 = E.p1 = E.p2 = 0.f;
#pragma unroll
    for (int r = 0; r < kAccRep; ++r) E.h[r] = E.l[r] = 0;
    if (t >= 2 * C) return;
    if (!src.acc) {
        E.p0 = saved[2 * C + t];
        E.p1 = saved[t];
        return;
    }
    const int C_each = C / src.n_src;
    const int which = t / C, c = t - which * C;
    const int k = c / C_each, cl = c - k * C_each;
    const long long* base = src.acc + (size_t)k * gn_acc_words(C_each);
#pragma unroll
    for (int r = 0; r < kAccRep; ++r) {
        const int rr = r < src.n_rep ? r : 0;  // (clamped: a replica read twice is not added)
        const long long* p = base + gn_acc_index(rr, which, cl, C_each);
        E.h[r] = p[0];
        E.l[r] = p[1];
    }
    if (t < C) {
        E.p0 = src.gamma[t];
        E.p1 = src.beta[t];
        E.p2 = src.alpha[t];
    }
}

__device__ __forceinline__ void gn_coef_early_finish(const GnExactSrc& src, int C, int64_t N, const GnCoefEarly& E, double* sums,
                                                     float* coef_s, float* mu_rstd_s) {
    const int t = threadIdx.x;
    if (!src.acc) {
        if (t < 2 * C) {
            coef_s[t] = E.p0;
            if (mu_rstd_s) mu_rstd_s[t] = E.p1;
        }
        __syncthreads();
        return;
    }
    if (t < 2 * C) {
        long long hi = 0, lo = 0;
        bool bad = false;
#pragma unroll
        for (int r = 0; r < kAccRep; ++r) {
            hi += r < src.n_rep ? E.h[r] : 0;
            lo += r < src.n_rep ? E.l[r] : 0;
            bad |= r < src.n_rep && E.l[r] < 0;
        }
        sums[t] = gn_acc_value(hi, lo, bad, 1.0 / kAccScaleFwd);
    }
    __syncthreads();
    if (t < C) {
        float mu, rstd, scale, shift;
        gn_fwd_coeffs(sums[t], sums[C + t], (double)N, E.p0, E.p1, E.p2, src.eps, mu, rstd, scale, shift);
        coef_s[t] = scale;
        coef_s[C + t] = shift;
        if (mu_rstd_s) {
            mu_rstd_s[t] = mu;
            mu_rstd_s[C + t] = rstd;
        }
        if (blockIdx.x == 0 && src.saved_w) {
            src.saved_w[t] = mu;
            src.saved_w[C + t] = rstd;
            src.saved_w[2 * C + t] = scale;
            src.saved_w[3 * C + t] = shift;
        }
    }
    __syncthreads();
}

// The two sums of C columns held by ONE accumulator block, for a workgroup of T = 4C threads, WITHOUT a barrier or LDS:
// wave w, lane l: column c = 16w + (l & 15), sum `which` = (l >> 4) & 1, replica half = l >> 5 — every 16 lanes read 256
// contiguous bytes per load (with the four lanes of a column ADJACENT, each lane quad touched four cache lines per load and
// the fold cost 2.4 us more) — all loads in flight at once next to whatever the caller issued before; the four lanes of
// a column combine by lane shuffles (integer adds: any order gives the same bits).  Returns true in the lanes l < 16, which
// then hold v0 = sum 0 and v1 = sum 1 of their column `col`.
template <int C, int T>
__device__ __forceinline__ bool gn_acc_col_sums(const long long* __restrict__ acc, int n_rep, double scale, int& col, double& v0,
                                                double& v1) {
    static_assert(T == 4 * C && C % 16 == 0 && kAccRep % 2 == 0, "four lanes per column, sixteen columns per wave");
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = 16 * w + (l & 15), which = (l >> 4) & 1, half = l >> 5;
    col = c;
    long long vh[kAccRep / 2], vl[kAccRep / 2];
    const int per = n_rep >> 1;  // replicas per lane (n_rep is even)
#pragma unroll
    for (int r = 0; r < kAccRep / 2; ++r) {
        const int rr = r < per ? half * per + r : 0;  // (clamped: a replica read twice is not added)
        const long long* p = acc + gn_acc_index(rr, which, c, C);
        vh[r] = p[0];
        vl[r] = p[1];
    }
    long long hi = 0, lo = 0;
    int bad = 0;
#pragma unroll
    for (int r = 0; r < kAccRep / 2; ++r) {
        hi += r < per ? vh[r] : 0;
        lo += r < per ? vl[r] : 0;
        bad |= (r < per && vl[r] < 0) ? 1 : 0;
    }
    hi += __shfl_xor(hi, 32);
    lo += __shfl_xor(lo, 32);
    bad |= __shfl_xor(bad, 32);
    const double v = gn_acc_value(hi, lo, bad != 0, 1.0 / scale);
    const double other = __shfl_xor(v, 16);  // lanes with which == 0 receive sum 1
    v0 = v;
    v1 = other;
    return l < 16;
}

// The same fold in two steps, so that the caller can ISSUE these loads together with everything else its prologue reads and
// wait once (ISA of the one-step form inside the staged forward kernels: weights / bias / label byte, wait; operands and
// accumulators, wait; gamma / beta / alpha — sunk by the compiler into the l < 16 branch — wait: three dependent round
// trips of ~1 us each before the first stage).  issue: 8 x 16-B accumulator loads + 3 parameter loads per lane, no branch;
// finish: lane shuffles, ONE fp64 division (1 / N) and a reciprocal square root instead of two divisions + sqrt + division,
// every lane computing (no branch around the arithmetic either), lanes l < 16 writing.
struct GnCoefRegs {
    long long vh[kAccRep / 2], vl[kAccRep / 2];
    float gamma, beta, alpha;
};

// (issue leaves R untouched when the statistics are final already — no zero-fill on that path: values merged at a join
// point would make the compiler wait for the loads right there)
template <int C, int T>
__device__ __forceinline__ void gn_fwd_coef_issue(const GnExactSrc& src, GnCoefRegs& R) {
    static_assert(T == 4 * C && C % 16 == 0 && kAccRep % 2 == 0, "four lanes per column, sixteen columns per wave");
    const int t = threadIdx.x;
    const int l = t & 63, w = t >> 6;
    const int c = 16 * w + (l & 15), which = (l >> 4) & 1, half = l >> 5;
    const int per = src.n_rep >> 1;
#pragma unroll
    for (int r = 0; r < kAccRep / 2; ++r) {
        const int rr = r < per ? half * per + r : 0;  // (clamped: a replica read twice is not added)
        const long long* p = src.acc + gn_acc_index(rr, which, c, C);
        R.vh[r] = p[0];
        R.vl[r] = p[1];
    }
}

// gamma / beta / alpha of the lane's column: issue them LAST in the caller's prologue and pin them (glass_pin) behind every
// other issue — the compiler converts them to double right behind the load wherever that sits, i.e. it waits there for
// everything issued before (vmcnt counts in order)
template <int C>
__device__ __forceinline__ void gn_fwd_coef_issue_params(const GnExactSrc& src, GnCoefRegs& R) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = 16 * w + (l & 15);
    R.gamma = src.gamma[c];
    R.beta = src.beta[c];
    R.alpha = src.alpha[c];
}

template <int C, int T>
__device__ __forceinline__ void gn_fwd_coef_finish(const GnExactSrc& src, const float* __restrict__ saved, int64_t N,
                                                   const GnCoefRegs& R, float* coef_s) {
    const int t = threadIdx.x;
    const int l = t & 63, w = t >> 6;
    const int c = 16 * w + (l & 15);
    if (!src.acc) {  // final statistics: scale | shift are saved[2C .. 4C)   (wave-uniform branch)
        if (t < 2 * C) coef_s[t] = saved[2 * C + t];
        return;
    }
    const int per = src.n_rep >> 1;
    long long hi = 0, lo = 0;
    int bad = 0;
#pragma unroll
    for (int r = 0; r < kAccRep / 2; ++r) {
        hi += r < per ? R.vh[r] : 0;
        lo += r < per ? R.vl[r] : 0;
        bad |= (r < per && R.vl[r] < 0) ? 1 : 0;
    }
    hi += __shfl_xor(hi, 32);
    lo += __shfl_xor(lo, 32);
    bad |= __shfl_xor(bad, 32);
    const double v = gn_acc_value(hi, lo, bad != 0, 1.0 / kAccScaleFwd);
    const double other = __shfl_xor(v, 16);
    // lanes with which == 0 hold s = sum x in v and q = sum x^2 in `other`; the rest compute on swapped values and write nothing
    const double s = v, q = other;
    const double invN = 1.0 / (double)N;
    const double a = (double)R.alpha;
    const double mu = s * invN;
    double var = q * invN - mu * mu * (2.0 * a - a * a);
    if (var < 0.0) var = 0.0;
    const double rstd = rsqrt(var + (double)src.eps);
    const double scale = (double)R.gamma * rstd;
    const float shift = (float)((double)R.beta - scale * a * mu);
    if (l < 16) {
        coef_s[c] = (float)scale;
        coef_s[C + c] = shift;
        if (blockIdx.x == 0 && src.saved_w) {
            src.saved_w[c] = (float)mu;
            src.saved_w[C + c] = (float)rstd;
            src.saved_w[2 * C + c] = (float)scale;
            src.saved_w[3 * C + c] = shift;
        }
    }
}

// Forward coefficients of a GraphNorm (C columns, one accumulator block) into coef_s (LDS: scale[C] | shift[C]) WITHOUT a
// barrier: the caller's next barrier publishes them.  Workgroup 0 also writes saved_w.
template <int C, int T>
__device__ __forceinline__ void gn_fwd_coef_nobarrier(const GnExactSrc& src, const float* __restrict__ saved, int64_t N,
                                                      float* coef_s) {
    const int t = threadIdx.x;
    if (!src.acc) {  // final statistics: scale | shift are saved[2C .. 4C)
        if (t < 2 * C) coef_s[t] = saved[2 * C + t];
        return;
    }
    const int c = 16 * (t >> 6) + (t & 15);
    const float gamma = src.gamma[c], beta = src.beta[c], alpha = src.alpha[c];
    double s, q;
    int col;
    if (gn_acc_col_sums<C, T>(src.acc, src.n_rep, kAccScaleFwd, col, s, q)) {
        float mu, rstd, scale, shift;
        gn_fwd_coeffs(s, q, (double)N, gamma, beta, alpha, src.eps, mu, rstd, scale, shift);
        coef_s[c] = scale;
        coef_s[C + c] = shift;
        if (blockIdx.x == 0 && src.saved_w) {
            src.saved_w[c] = mu;
            src.saved_w[C + c] = rstd;
            src.saved_w[2 * C + c] = scale;
            src.saved_w[3 * C + c] = shift;
        }
    }
}

// Backward coefficients of a GraphNorm (64 columns, one accumulator block) into coef_s (LDS floats [5 * 64]: A | Bx | K |
// scale | shift) WITHOUT a barrier (the caller's next barrier publishes them); the workgroup with write_params also writes
// the parameter gradients (dgamma = s2, dbeta = s1, dalpha).  Every thread of the 256-thread workgroup calls it.
__device__ __forceinline__ void gn_bwd_coef_nobarrier(const long long* __restrict__ acc, int n_rep, int64_t N,
                                                      const float* __restrict__ saved, const float* __restrict__ gamma,
                                                      const float* __restrict__ alpha, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ dalpha, int accumulate,
                                                      bool write_params, float* coef_s) {
    constexpr int C = 64;
    const int t = threadIdx.x;
    const int c = 16 * (t >> 6) + (t & 15);
    const float ga = gamma[c], al = alpha[c], mu = saved[c], rstd = saved[C + c], sc = saved[2 * C + c], sh = saved[3 * C + c];
    double s1, s2;
    int col;
    if (gn_acc_col_sums<C, 256>(acc, n_rep, kAccScaleBwd, col, s1, s2)) {
        float A, Bx, K, da;
        gn_bwd_coeffs(s1, s2, (double)N, ga, al, mu, rstd, A, Bx, K, da);
        coef_s[c] = A;
        coef_s[C + c] = Bx;
        coef_s[2 * C + c] = K;
        coef_s[3 * C + c] = sc;
        coef_s[4 * C + c] = sh;
        if (write_params) {
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
            if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + da;
        }
    }
}

// dc (four columns of one row) from the GraphNorm's output gradient: the arithmetic of gn_bwd_apply_acc_kernel
__device__ __forceinline__ float4 gn_bwd_apply4(const float4& dy, const float4& x, const float4& ad, const float* coef_s, int c0,
                                                int act, const float (&ds)[4]) {
    constexpr int C = 64;
    const float4 A = *reinterpret_cast<const float4*>(coef_s + c0), Bx = *reinterpret_cast<const float4*>(coef_s + C + c0);
    const float4 K = *reinterpret_cast<const float4*>(coef_s + 2 * C + c0);
    const float4 sc = *reinterpret_cast<const float4*>(coef_s + 3 * C + c0), sh = *reinterpret_cast<const float4*>(coef_s + 4 * C + c0);
    const float dv[4] = {dy.x, dy.y, dy.z, dy.w}, xv[4] = {x.x, x.y, x.z, x.w}, av[4] = {ad.x, ad.y, ad.z, ad.w};
    const float Av[4] = {A.x, A.y, A.z, A.w}, Bv[4] = {Bx.x, Bx.y, Bx.z, Bx.w}, Kv[4] = {K.x, K.y, K.z, K.w};
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g = dv[k] * ds[k];
        g *= act_grad(act, fmaf(xv[k], scv[k], shv[k]));
        o[k] = fmaf(Av[k], g, fmaf(Bv[k], xv[k], Kv[k])) + av[k];
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace glass
