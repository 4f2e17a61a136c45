// GraphNorm per-column coefficient math (fp64), shared by the whole-graph kernels (graphnorm.hip) and the
// embedding-table shortcut (embnorm.hip).  PyG GraphNorm with batch=None:
//   out = x - a*mu;  var = mean(out^2);  y = gamma * out / sqrt(var + eps) + beta.
#pragma once
#include "common.h"

namespace glass {

// forward statistics of one column from s = sum(x), q = sum(x^2) over N rows -> (mu, rstd, scale, shift),
// y = x*scale + shift
__device__ __forceinline__ void gn_fwd_coeffs(double s, double q, double N, float gamma, float beta, float alpha,
                                              float eps, float& mu_f, float& rstd_f, float& scale_f, float& shift_f) {
    const double a = (double)alpha;
    const double mu = s / N;
    // mean((x - a*mu)^2) = E[x^2] - mu^2 * (2a - a^2)   (exact in fp64 for fp32 data)
    double var = q / N - mu * mu * (2.0 * a - a * a);
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double scale = (double)gamma * rstd;
    mu_f = (float)mu;
    rstd_f = (float)rstd;
    scale_f = (float)scale;
    shift_f = (float)((double)beta - scale * a * mu);
}

// backward from s1 = sum(g), s2 = sum(g * xhat), xhat = (x - a*mu)*rstd:  dx = A*g + Bx*x + K;
// dgamma = s2, dbeta = s1, dalpha returned.
__device__ __forceinline__ void gn_bwd_coeffs(double s1, double s2, double N, float gamma, float alpha, float mu_f,
                                              float rstd_f, float& A_f, float& Bx_f, float& K_f, float& dalpha_f) {
    const double g = (double)gamma, a = (double)alpha;
    const double mu = (double)mu_f, r = (double)rstd_f;
    const double m2 = s2 / N;
    const double sum_xhat = r * N * mu * (1.0 - a);          // sum_n (x_n - a*mu) * r
    const double sum_do = g * r * (s1 - sum_xhat * m2);       // sum_n d o_n
    // dx = do - a*mean(do),  do = g*r*(gr - xhat*m2)
    A_f = (float)(g * r);
    Bx_f = (float)(-g * r * r * m2);
    K_f = (float)(g * r * r * m2 * a * mu - a * (sum_do / N));
    dalpha_f = (float)(-mu * sum_do);
}

// Sum partial[b][2][C] (doubles, written by other workgroups — every load is an L2 miss) over b for 4 columns per
// workgroup: 64 partial slots x 8 partials per thread in flight, so up to 512 partials cost ONE memory round trip;
// slots folded through LDS in fixed order.  Result valid in the threads with slot 0 (tr == 0).  lds: kBlock*2 doubles.
constexpr int kFinCols = 4, kFinSlots = kBlock / kFinCols, kFinFly = 8;

__device__ __forceinline__ void gn_sum_partials(const double* __restrict__ part, int nblk, int C_part, int cl, bool ok,
                                                int tc, int tr, double* lds, double& ss, double& qq) {
    ss = 0.0, qq = 0.0;
    if (ok)
        for (int b = tr; b < nblk; b += kFinSlots * kFinFly) {
            double s[kFinFly], q[kFinFly];
#pragma unroll
            for (int u = 0; u < kFinFly; ++u) {
                const int bb = b + kFinSlots * u;
                s[u] = bb < nblk ? part[((size_t)bb * 2) * C_part + cl] : 0.0;
                q[u] = bb < nblk ? part[((size_t)bb * 2 + 1) * C_part + cl] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kFinFly; ++u) {
                ss += s[u];
                qq += q[u];
            }
        }
    lds[threadIdx.x * 2] = ss;
    lds[threadIdx.x * 2 + 1] = qq;
    __syncthreads();
    double a = 0.0, b2 = 0.0;
    if (tr < 8)
        for (int r = tr; r < kFinSlots; r += 8) {
            a += lds[(r * kFinCols + tc) * 2];
            b2 += lds[(r * kFinCols + tc) * 2 + 1];
        }
    __syncthreads();
    if (tr < 8) {
        lds[threadIdx.x * 2] = a;
        lds[threadIdx.x * 2 + 1] = b2;
    }
    __syncthreads();
    ss = 0.0, qq = 0.0;
    if (tr == 0)
        for (int r = 0; r < 8; ++r) {
            ss += lds[(r * kFinCols + tc) * 2];
            qq += lds[(r * kFinCols + tc) * 2 + 1];
        }
}

// One workgroup of the backward finalize (columns 4*blk .. 4*blk+3): coefficients coef[3C] = A, Bx, K and the
// parameter gradients from the partial sums.
__device__ __forceinline__ void gn_finalize_bwd_block(int blk, const double* __restrict__ partial, int nblk, int C,
                                                      int64_t N, const float* __restrict__ gamma,
                                                      const float* __restrict__ alpha, const float* __restrict__ saved,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* __restrict__ dalpha, int accumulate,
                                                      float* __restrict__ coef, double* lds) {
    const int tc = threadIdx.x & (kFinCols - 1), tr = threadIdx.x / kFinCols;
    const int c = blk * kFinCols + tc;
    double s1, s2;
    gn_sum_partials(partial, nblk, C, c, c < C, tc, tr, lds, s1, s2);
    if (tr == 0 && c < C) {
        float da;
        gn_bwd_coeffs(s1, s2, (double)N, gamma[c], alpha[c], saved[c], saved[C + c], coef[c], coef[C + c],
                      coef[2 * C + c], da);
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
        if (dalpha) dalpha[c] = (accumulate ? dalpha[c] : 0.f) + da;
    }
}

}  // namespace glass
