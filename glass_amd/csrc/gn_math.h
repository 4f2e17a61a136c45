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

}  // namespace glass
