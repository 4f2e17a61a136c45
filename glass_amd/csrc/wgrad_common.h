// Shared by the two weight-gradient kernel families: linear.hip (operands straight from global memory, 128 x 64 tile,
// small graphs / narrow layers) and wgrad_tiled.hip (LDS-tiled 128 x 256 tile: hidden >= 256 on large graphs).
#pragma once
#include "common.h"

namespace glass {

// SYNTH: G is not read but synthesised from the gradient of the mixed output (glass_dual_linear_wgrad_f32):
//   G[n,o] = coef(mask[n], o<H) * dsrc[n, o mod H] * (act ? ELU'(T[n,o]) : 1),  O = 2H,
// and X may be the virtual concatenation [X | X2] (each H wide) of the comb Linear's two inputs.
struct WgradSynth {
    const float* dsrc;   // [N,H]
    int64_t ldd;
    const float* T;      // [N,2H] pre-activations (act != 0)
    int64_t ldt;
    const uint8_t* mask;
    float zr, omz;
    int act, H;
    const float* X2;     // second input half (may be null)
    int64_t ldx2;
};

// wgrad_tiled.hip: used by glass_dual_linear_wgrad_f32 (and the deferred reduction of its partials) when
// wgrad_tiled_shape(N, O, I) — the partial kernel writes plain [slab][tile][128][256] partial
// sums (+ [slab][o-tile][128] bias partials), the reduce kernel sums the slabs in order into dW / db.
struct TiledWgradGeom {
    int n_slabs, rows_per_slab, ny, nz;      // ny = I / 256 input tiles, nz = O / 128 output tiles
    int64_t part_w_floats, part_b_floats;
};
bool wgrad_tiled_shape(int64_t N, int64_t O, int64_t I);
TiledWgradGeom wgrad_tiled_geom(int64_t N, int64_t O, int64_t I);
// the gradient operand is always synthesised from sy (glass_dual_linear_wgrad_f32); part_b == nullptr: no bias gradient
void launch_tiled_wgrad_partial(const float* X, int64_t ldx, int64_t N, int64_t O, int64_t I, const WgradSynth& sy,
                                float* part_w, float* part_b, hipStream_t st);
void launch_tiled_wgrad_reduce(const float* part_w, const float* part_b, int64_t N, int64_t O, int64_t I, float* dW,
                               int64_t lddw, float* db, int accumulate, hipStream_t st);

}  // namespace glass
