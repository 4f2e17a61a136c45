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
// activations and their squares can reach over 10^6 rows; out-of-range partials saturate.  Layout: int64
// [kAccRep][2 quantities][C][2 limbs], zeroed once per step by the prologue launch (glass_step_prologue_f32).
#pragma once
#include "common.h"
#include "gn_math.h"

namespace glass {

#ifndef GLASS_ACC_REP
#define GLASS_ACC_REP 16
#endif
// replicas a column's adds are spread over (workgroup b -> replica b % kAccRep).  Adds to ONE address queue at the memory-side
// atomic unit (~100 ns each, measured: the 134-workgroup statistics kernel 10.6 / 8.1 / 6.6 us with 4 / 8 / 16 replicas, 5.4 us
// when it writes plain partials); the consumers' fold loads are all in flight at once, so their cost barely moves.
constexpr int kAccRep = GLASS_ACC_REP;

// hi-limb scale of the two uses: forward sums (x, x^2: large, never tiny against eps) keep 2^-52 resolution and 2^50 range;
// backward sums (gradients: small) trade range for resolution — 2^-64 per workgroup partial, |sum| < 2^38
constexpr double kAccScaleFwd = 4096.0;      // 2^12
constexpr double kAccScaleBwd = 16777216.0;  // 2^24

__device__ __forceinline__ size_t gn_acc_index(int rep, int which, int c, int C) { return (((size_t)rep * 2 + which) * C + c) * 2; }
__host__ __device__ constexpr int64_t gn_acc_words(int64_t C) { return (int64_t)kAccRep * 2 * C * 2; }

__device__ __forceinline__ void gn_acc_add(long long* __restrict__ acc, int rep, int which, int c, int C, double v, double scale) {
    double sv = v * scale;
    sv = fmin(fmax(sv, -4.0e18), 4.0e18);
    const double fl = floor(sv);
    const long long hi = (long long)fl;
    const long long lo = (long long)((sv - fl) * 1099511627776.0);  // 2^40
    unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + gn_acc_index(rep, which, c, C));
    atomicAdd(p, (unsigned long long)hi);
    atomicAdd(p + 1, (unsigned long long)lo);
}

// Every thread of the workgroup calls this; afterwards out[which * C + c] (LDS doubles, 2 * C of them, C = n_src * C_each)
// holds the two sums of every column.  acc: n_src consecutive accumulator blocks of C_each columns — block k covers columns
// k * C_each .. (the column blocks of a jumping-knowledge buffer, each summed by the kernel that wrote it).
__device__ __forceinline__ void gn_acc_fold(const long long* __restrict__ acc, int C_each, int n_src, double* out, double scale) {
    const int C = C_each * n_src;
    const double inv = 1.0 / scale;
    for (int item = threadIdx.x; item < 2 * C; item += blockDim.x) {
        const int which = item / C, c = item - which * C;
        const int k = c / C_each, cl = c - k * C_each;
        const long long* base = acc + (size_t)k * gn_acc_words(C_each);
        long long hi = 0, lo = 0;
#pragma unroll
        for (int r = 0; r < kAccRep; ++r) {
            const long long* p = base + gn_acc_index(r, which, cl, C_each);
            hi += p[0];
            lo += p[1];
        }
        out[item] = ((double)hi + (double)lo * (1.0 / 1099511627776.0)) * inv;
    }
    __syncthreads();
}

// A GraphNorm whose forward sums are still in accumulators: the consumer derives the coefficients itself.
struct GnExactSrc {
    const long long* acc;  // nullptr: the statistics are final in `saved` already
    int n_src;             // accumulator blocks (of C / n_src columns each)
    const float *gamma, *beta, *alpha;
    float eps;
    float* saved_w;        // [4C] mean, rstd, scale, shift: written by workgroup 0 for the backward
};

// Consumer prologue, every thread of the workgroup: scale | shift of all C columns into coef_s (LDS floats [2C]) — folded
// from the accumulators (sums: LDS doubles [2C]), or copied from the final `saved`.  mu_rstd_s (LDS [2C], optional): mean | rstd.
__device__ __forceinline__ void gn_fwd_coef_block(const GnExactSrc& src, const float* __restrict__ saved, int C, int64_t N,
                                                  double* sums, float* coef_s, float* mu_rstd_s) {
    if (src.acc) {
        gn_acc_fold(src.acc, C / src.n_src, src.n_src, sums, kAccScaleFwd);
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

}  // namespace glass
