// Forward half of the embedding-table shortcut (embnorm.hip) as a device function, so that it can also ride in the
// once-per-step prologue launch next to the weight packing (dense.hip, pack_batch_kernel).
#pragma once
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

// One workgroup (columns 16*blk .. 16*blk+15) of the table forward: emb_gn's statistics as count-weighted sums over the
// V table rows -> saved[4H] = mean, rstd, scale, shift; table (may be NULL) = W*scale + shift.
// lds: kBlock*2 doubles; coef: 2*kTabCols floats (shared).
__device__ __forceinline__ void emb_table_fwd_block(int blk, const float* __restrict__ W, int V, int H,
                                                    const int32_t* __restrict__ rowptr, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ alpha,
                                                    float eps, float* __restrict__ saved, float* __restrict__ table,
                                                    double* lds, float* coef) {
    const int tc = threadIdx.x & (kTabCols - 1), tr = threadIdx.x / kTabCols;
    const int c = blk * kTabCols + tc;
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
    if (table == nullptr) return;  // (workgroup-uniform)
    __syncthreads();
    if (!ok) return;
    const float scale = coef[tc], shift = coef[kTabCols + tc];
    for (int v = tr; v < V; v += kTabSlots) table[(int64_t)v * H + c] = fmaf(W[(int64_t)v * H + c], scale, shift);
}

}  // namespace glass
