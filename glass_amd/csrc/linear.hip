// Weight/bias gradient of the stacked Linears of GLASSConv:  dW[o,i] (+)= sum_n G[n,o] * X[n,i],
// db[o] (+)= sum_n G[n,o]   (autograd backward of nn.Linear at reference impl/models.py:158-159,
// 169-170; G = gradient of the [N,2H] Linear output, X = its [N,H] or [N,2H] input).
//
// This is the one genuinely dense contraction on the path that the vendor GEMM serves badly: the
// output is tiny (128x64 .. 128x128) and the reduction dimension is the node count (17 080 .. 1 M),
// so hipBLASLt launches 8-28 workgroups and takes 80-95 us at ppi_bp-shape (profiles/r01_*).  Here
// the reduction is split over up to 256 workgroups ("split-K" over rows) on the fp32 matrix cores:
//   v_mfma_f32_32x32x2_f32, A = G^T tile (32 outputs x 2 rows), B = X tile (2 rows x 32 inputs).
// Loads are whole-row and vectorised: lane (c = l&31, h = l>>5) reads float4 G[n+h][4c..4c+3] and
// float2 X[n+h][2c..2c+1]; component t of the float4 feeds output tile t, whose 32 rows are the
// STRIDED set o = 4r+t (likewise inputs i = 2c+u) — the MFMA does not care which 32 outputs form a
// tile, so no shuffle or LDS transpose is needed.  One wave accumulates a full 128x64 block in 128
// accumulator registers; the 4 waves of a workgroup take interleaved row pairs and are combined
// through LDS; workgroup partials are summed in fixed order by a second kernel (deterministic).
// fp32 MFMA is an exact k-ordered fmaf chain, so numerics equal a plain fp32 reduction.
#include "common.h"
#include "wgrad_common.h"
#include "spmm_common.h"
#include "dense_common.h"

#include <stdlib.h>

namespace glass {

// The partial-sum kernel body lives in wgrad_common.h (wgrad_partial_body): it is also one branch of the fused
// backward launch of dense.hip.
template <bool SYNTH, bool EFF>
__global__ __launch_bounds__(kBlock, 1) void wgrad_partial_kernel(const float* __restrict__ G, int64_t ldg,
                                                                  const float* __restrict__ X, int64_t ldx,
                                                                  int64_t N, int O, int I, int rows_per_slab,
                                                                  float* __restrict__ part_w,
                                                                  float* __restrict__ part_b, float* __restrict__ header,
                                                                  WgradSynth sy) {
    __shared__ float lds[2 * kTile + 8 * kOT];  // 64 KiB: two wave-sized accumulator images; + bias partials [wave*2 + h][o]
    if (header && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        header[0] = EFF ? 1.f : 0.f;  // tells the (possibly deferred) reduce launch which form the partials have
        header[1] = sy.zr;
        header[2] = 0.f;  // (tiles in the permuted accumulator order)
    }
    wgrad_partial_body<SYNTH, 4, EFF>(G, ldg, X, ldx, N, O, I, rows_per_slab, part_w, part_b, sy, blockIdx.x, blockIdx.y,
                                      blockIdx.z, gridDim.x, gridDim.y, lds, lds + 2 * kTile, gridDim.z);
}

// ... with split products (wgrad_partial_split_body): the synthesised-gradient forms only (O = 2H, I = H or 2H: whole tiles)
template <bool ACT, bool EFF>
__global__ __launch_bounds__(kBlock, 1) void wgrad_partial_split_kernel(const float* __restrict__ X, int64_t ldx, int64_t N, int O,
                                                                        int I, int rows_per_slab, float* __restrict__ part_w,
                                                                        float* __restrict__ part_b, float* __restrict__ header,
                                                                        WgradSynth sy) {
    __shared__ float lds[2 * kTile + 8 * kOT];
    if (header && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        header[0] = EFF ? 1.f : 0.f;
        header[1] = sy.zr;
        header[2] = 0.f;
    }
    wgrad_partial_split_body<ACT, EFF>(X, ldx, N, O, I, rows_per_slab, part_w, part_b, sy, blockIdx.x, blockIdx.y, blockIdx.z,
                                       gridDim.x, gridDim.y, lds, lds + 2 * kTile, gridDim.z);
}

// Sum the slab partials and scatter to dW[o,i] / db[o].  A [n_slabs x 8192(+128)] column reduction:
// 16 lanes x float4 cover 64 columns, 16 row slots walk the slabs (4 loads in flight each) and are
// combined through LDS in slot order -> fixed summation order.
// header (behind the bias partials): [0] != 0: effective-weight form (wgrad_partial_body<.., EFF>) — output tile z of the f1
// half = (1-z) S + (2z-1) L, of the f0 half = z S - (2z-1) L with S / L the slab sums of tiles zz and nz/2 + zz.
// S / L form (wgrad_sl_body; comb pair at hidden 64): n_s slab tiles then n_l labeled-row tiles, one [64 x 128] tile each;
// dW[o][i] = (1-z) S + (2z-1) L, dW[64 + o][i] = z S - (2z-1) L, bias gradients likewise from the [64] column sums.
__device__ __forceinline__ void wgrad_reduce_sl_body(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                     int n_s, int n_l, float* __restrict__ dW, int64_t lddw,
                                                     float* __restrict__ db, int accumulate, float4* lds) {
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int k0 = blockIdx.x * 64 + tc * 4;
    const bool is_bias = k0 >= kTile;
    if (is_bias && (k0 - kTile >= kSLOut || db == nullptr)) return;  // whole workgroup: 64 columns per block
    const float zr = part_b[(int64_t)(n_s + n_l) * kSLOut + 1];
    const bool plain = part_b[(int64_t)(n_s + n_l) * kSLOut + 2] != 0.f;  // tiles in [o][i] order (comb_bwd_eff2_kernel)
    const float* p = is_bias ? part_b + (k0 - kTile) : part_w + k0;
    const int64_t stride = is_bias ? kSLOut : kTile;
    auto run_sum = [&](int b0, int b1) __attribute__((always_inline)) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = b0 + tr; b < b1; b += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = b + 16 * u;
                v[u] = bb < b1 ? *reinterpret_cast<const float4*>(p + (int64_t)bb * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        return s;
    };
    const float4 sS = run_sum(0, n_s), sL = run_sum(n_s, n_s + n_l);
    lds[threadIdx.x] = sS;
    lds[kBlock + threadIdx.x] = sL;
    __syncthreads();
    if (tr != 0) return;
    float4 S = sS, L = sL;
    for (int r = 1; r < 16; ++r) {
        const float4 a = lds[r * 16 + tc], b = lds[kBlock + r * 16 + tc];
        S.x += a.x; S.y += a.y; S.z += a.z; S.w += a.w;
        L.x += b.x; L.y += b.y; L.z += b.z; L.w += b.w;
    }
    const float sv[4] = {S.x, S.y, S.z, S.w}, lv[4] = {L.x, L.y, L.z, L.w};
    const float cl = 2.f * zr - 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = k0 + q;
        const float g1 = (1.f - zr) * sv[q] + cl * lv[q], g0 = zr * sv[q] - cl * lv[q];
        if (!is_bias) {
            // decode k = ((t*4+u)*16 + reg)*64 + lane  ->  (o, i) = (2m + t, 4n + u)
            const int lane = k & 63, reg = (k >> 6) & 15, tu = k >> 10;
            const int t = tu >> 2, u = tu & 3;
            const int m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), n = lane & 31;
            const int o = plain ? k >> 7 : 2 * m + t, i = plain ? k & 127 : 4 * n + u;
            float* d1 = dW + (int64_t)o * lddw + i;
            float* d0 = dW + (int64_t)(kSLOut + o) * lddw + i;
            *d1 = accumulate ? *d1 + g1 : g1;
            *d0 = accumulate ? *d0 + g0 : g0;
        } else {
            const int o = k - kTile;
            db[o] = accumulate ? db[o] + g1 : g1;
            db[kSLOut + o] = accumulate ? db[kSLOut + o] + g0 : g0;
        }
    }
}

// narrow form (dense_narrow.hip; hidden <= 32): plain row-major slab partials [n_slabs][stride] = dW[O][I] then db[O]
__device__ __forceinline__ void wgrad_reduce_narrow_body(const float* __restrict__ part, int n_slabs, int stride, int O, int I,
                                                         float* __restrict__ dW, int64_t lddw, float* __restrict__ db,
                                                         int accumulate, float4* lds) {
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int k0 = blockIdx.x * 64 + tc * 4;
    if (blockIdx.x * 64 >= stride) return;  // (workgroup-uniform)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k0 < stride)
        for (int b = tr; b < n_slabs; b += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = b + 16 * u;
                v[u] = bb < n_slabs ? *reinterpret_cast<const float4*>(part + (int64_t)bb * stride + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
    lds[threadIdx.x] = s;
    __syncthreads();
    if (tr != 0 || k0 >= stride) return;
    for (int r = 1; r < 16; ++r) {
        const float4 o = lds[r * 16 + tc];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = k0 + q;
        if (k < O * I) {
            float* d = dW + (int64_t)(k / I) * lddw + (k % I);
            *d = accumulate ? *d + sv[q] : sv[q];
        } else if (k < O * I + O && db) {
            const int o = k - O * I;
            db[o] = accumulate ? db[o] + sv[q] : sv[q];
        }
    }
}

__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                  int n_slabs, int ny, int nz, int O, int I, float* __restrict__ dW,
                                                  int64_t lddw, float* __restrict__ db, int accumulate, float4* lds,
                                                  int chunk /* z * ny + y */) {
    const int z = chunk / ny, y = chunk % ny;
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int k0 = blockIdx.x * 64 + tc * 4;       // first of this thread's 4 columns
    const bool is_bias = k0 >= kTile;
    if (is_bias && (y != 0 || db == nullptr)) return;  // whole workgroup: blockIdx.x is uniform
    const float* header = part_b + (int64_t)nz * n_slabs * kOT;
    const bool eff = header[0] != 0.f;
    const float zr = header[1];
    const bool plain = header[2] != 0.f;  // tiles in [o][i] order (wgrad_trans_staged2_body)
    // header[0] == 3: effective-weight form whose labeled-rows tiles are LIST tiles (wgrad128_comb_kernel): the first
    // header[3] places of a (z >= nz / 2, y) block instead of one per slab
    const int n_list = header[0] == 3.f ? (int)header[3] : n_slabs;
    const int zz = eff ? z % (nz / 2) : z;
    const float cs = !eff ? 1.f : (z < nz / 2 ? 1.f - zr : zr);
    const float cl = !eff ? 0.f : (z < nz / 2 ? 2.f * zr - 1.f : 1.f - 2.f * zr);
    auto slab_sum = [&](int zsrc) __attribute__((always_inline)) {
        const float* p;
        int64_t stride;
        if (!is_bias) {
            p = part_w + (int64_t)(zsrc * ny + y) * n_slabs * kTile + k0;
            stride = kTile;
        } else {
            p = part_b + (int64_t)zsrc * n_slabs * kOT + (k0 - kTile);
            stride = kOT;
        }
        const int n_here = (eff && zsrc >= nz / 2) ? n_list : n_slabs;   // tiles of this block that hold sums
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = tr; b < n_here; b += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = b + 16 * u;
                v[u] = bb < n_here ? *reinterpret_cast<const float4*>(p + (int64_t)bb * stride)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        return s;
    };
    float4 s = slab_sum(zz);
    if (eff) {  // (workgroup-uniform)
        const float4 l = slab_sum(nz / 2 + zz);
        s = make_float4(cs * s.x + cl * l.x, cs * s.y + cl * l.y, cs * s.z + cl * l.z, cs * s.w + cl * l.w);
    }
    lds[threadIdx.x] = s;
    __syncthreads();
    if (tr != 0) return;
    for (int r = 1; r < 16; ++r) {
        const float4 o = lds[r * 16 + tc];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = k0 + q;
        if (!is_bias) {
            // decode k = ((t*2+u)*16 + reg)*64 + lane  ->  (o, i)
            const int lane = k & 63, reg = (k >> 6) & 15, tu = k >> 10;
            const int t = tu >> 1, u = tu & 1;
            const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), cc = lane & 31;
            const int o = z * kOT + (plain ? k >> 6 : 4 * r + t), i = y * kIT + (plain ? k & 63 : 2 * cc + u);
            if (o < O && i < I) {
                float* d = dW + (int64_t)o * lddw + i;
                *d = accumulate ? *d + sv[q] : sv[q];
            }
        } else {
            const int o = z * kOT + (k - kTile);
            if (o < O) db[o] = accumulate ? db[o] + sv[q] : sv[q];
        }
    }
}

__global__ __launch_bounds__(kBlock) void wgrad_reduce_kernel(const float* __restrict__ part_w,
                                                              const float* __restrict__ part_b, int n_slabs, int ny,
                                                              int nz, int O, int I, float* __restrict__ dW, int64_t lddw,
                                                              float* __restrict__ db, int accumulate) {
    __shared__ float4 lds[kBlock];
    wgrad_reduce_body(part_w, part_b, n_slabs, ny, nz, O, I, dW, lddw, db, accumulate, lds, blockIdx.y);
}

// Several weight gradients reduced by ONE launch (blockIdx.z = job): the partial kernels of a backward pass write to
// separate scratch buffers and their reductions are deferred to the end of the pass (glass_linear_wgrad_reduce_batch_f32).
struct ReduceJob {
    const float *part_w, *part_b;
    int n_slabs, ny, nz, O, I, accumulate;
    int n_l;  // > 0: S / L form (n_slabs = n_s slab tiles, then n_l labeled-row tiles; ny = nz = 1)
              // < 0: narrow form, -n_l = the slab stride in floats (ny = nz = 1)
    float* dW;
    int64_t lddw;
    float* db;
};
constexpr int kMaxReduceJobs = 8;
struct ReduceBatch {
    ReduceJob job[kMaxReduceJobs];
};

__global__ __launch_bounds__(kBlock) void wgrad_reduce_batch_kernel(ReduceBatch batch) {
    __shared__ float4 lds[2 * kBlock];
    const ReduceJob& j = batch.job[blockIdx.z];
    if ((int)blockIdx.y >= j.ny * j.nz) return;
    if (j.n_l > 0) {
        wgrad_reduce_sl_body(j.part_w, j.part_b, j.n_slabs, j.n_l, j.dW, j.lddw, j.db, j.accumulate, lds);
        return;
    }
    if (j.n_l < 0) {
        wgrad_reduce_narrow_body(j.part_w, j.n_slabs, -j.n_l, j.O, j.I, j.dW, j.lddw, j.db, j.accumulate, lds);
        return;
    }
    wgrad_reduce_body(j.part_w, j.part_b, j.n_slabs, j.ny, j.nz, j.O, j.I, j.dW, j.lddw, j.db, j.accumulate, lds, blockIdx.y);
}

// ---- thin outputs (O <= 3): the last Linear of the pre-training head maps hidden -> 1 over 131 072 pair rows
// (reference GNNEmb.py: MLP(hidden, hidden, 1, 2)); the MFMA slab kernel wants O % 4 == 0 and a library GEMM of shape
// [1 x N] . [N x I] runs at 288 us (rocprofv3, ppi_bp-shape).  Here: row slabs over the workgroups, TC lanes x float4 over
// the input columns, row slots combined through LDS in fixed order, partials reduced in fp64 by a second tiny launch.
constexpr int kThinMaxO = 3;
constexpr int kThinMaxBlocks = 1024;

template <int O>
__global__ __launch_bounds__(kBlock) void wgrad_thin_partial_kernel(const float* __restrict__ G, int64_t ldg,
                                                                    const float* __restrict__ X, int64_t ldx, int64_t N, int I,
                                                                    int rows_per_blk, int tc_log2, float* __restrict__ part) {
    __shared__ float red[kBlock * (O * 4 + O)];
    const int TC = 1 << tc_log2, li = threadIdx.x & (TC - 1), slot = threadIdx.x >> tc_log2, n_slot = kBlock >> tc_log2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = r0 + rows_per_blk < N ? r0 + rows_per_blk : N;
    float* my = part + (size_t)blockIdx.x * (O * I + O);
    for (int ch = 0; ch * TC * 4 < I; ++ch) {  // uniform trip count (barriers inside)
        const int c0 = (ch * TC + li) * 4;
        float acc[O][4], accb[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            accb[o] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[o][k] = 0.f;
        }
        if (c0 < I)
            for (int64_t r = r0 + slot; r < r1; r += n_slot) {
                const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + c0);
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const float g = G[r * ldg + o];
                    acc[o][0] = fmaf(g, x.x, acc[o][0]);
                    acc[o][1] = fmaf(g, x.y, acc[o][1]);
                    acc[o][2] = fmaf(g, x.z, acc[o][2]);
                    acc[o][3] = fmaf(g, x.w, acc[o][3]);
                    accb[o] += g;
                }
            }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < O; ++o) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[threadIdx.x * (O * 4 + O) + o * 4 + k] = acc[o][k];
            red[threadIdx.x * (O * 4 + O) + O * 4 + o] = accb[o];
        }
        __syncthreads();
        if (slot == 0 && c0 < I) {
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float s[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
                for (int t = 0; t < n_slot; ++t) {
                    const float* q = red + ((t << tc_log2) + li) * (O * 4 + O);
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += q[o * 4 + k];
                    sb += q[O * 4 + o];
                }
                *reinterpret_cast<float4*>(my + o * I + c0) = make_float4(s[0], s[1], s[2], s[3]);
                if (ch == 0 && li == 0) my[O * I + o] = sb;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void wgrad_thin_reduce_kernel(const float* __restrict__ part, int n_blk, int O, int I,
                                                                   float* __restrict__ dW, int64_t lddw, float* __restrict__ db,
                                                                   int accumulate) {
    // four outputs per workgroup, 64 slots striding over the partial blocks; slots combined in fixed order
    __shared__ double red[kBlock];
    const int ob = threadIdx.x & 3, slot = threadIdx.x >> 2, stride = O * I + O;
    const int idx = blockIdx.x * 4 + ob;
    double s = 0.0;
    if (idx < stride)
        for (int b = slot; b < n_blk; b += kBlock / 4) s += (double)part[(size_t)b * stride + idx];
    red[threadIdx.x] = s;
    __syncthreads();
    if (slot != 0 || idx >= stride) return;
    double t = 0.0;
    for (int r = 0; r < kBlock / 4; ++r) t += red[r * 4 + ob];
    const float v = (float)t;
    if (idx < O * I) {
        float* d = dW + (int64_t)(idx / I) * lddw + idx % I;
        *d = accumulate ? *d + v : v;
    } else if (db) {
        float* d = db + (idx - O * I);
        *d = accumulate ? *d + v : v;
    }
}

static int thin_blocks(int64_t N) {
    const int64_t b = ceil_div(N, 64);
    return (int)(b < kThinMaxBlocks ? b : kThinMaxBlocks);
}

WgradGeom wgrad_geom(int64_t N, int64_t O, int64_t I) {
    WgradGeom g;
    if (wgrad128_comb_shape(N, O, I)) {
        // comb pair of hidden 128 in S / L form with list tiles (wgrad128.hip): slab workgroups + the list workgroups fill one
        // round of the chip; the buffer keeps the effective-weight layout (an L block has room for n_slabs tiles, n_l are used)
        int64_t room = 256 - wgrad128_comb_lists(N);
        if (room < 64) room = 64;
        int64_t rows = ceil_div(ceil_div(N, room), (int64_t)32) * 32;
        if (rows < 64) rows = 64;
        g.rows_per_slab = (int)rows;
        g.n_slabs = (int)ceil_div(N, rows);
        if (g.n_slabs < wgrad128_comb_lists(N)) g.n_slabs = wgrad128_comb_lists(N);  // (an L block holds n_l tiles)
        g.ny = (int)ceil_div(I, kIT);
        g.nz = (int)ceil_div(O, kOT);
        g.part_w_floats = (int64_t)g.n_slabs * g.ny * g.nz * kTile;
        g.part_b_floats = (int64_t)g.n_slabs * g.nz * kOT + kWgradHeaderFloats;
        return g;
    }
    if (wgrad128_shape(N, O, I)) {
        // one workgroup per slab covers all four tiles (wgrad128.hip): about one slab per CU, whole 32-row stages
        int64_t rows = ceil_div(ceil_div(N, (int64_t)256), (int64_t)32) * 32;
        if (rows < 64) rows = 64;
        g.rows_per_slab = (int)rows;
        g.n_slabs = (int)ceil_div(N, rows);
        g.ny = (int)ceil_div(I, kIT);
        g.nz = (int)ceil_div(O, kOT);
        g.part_w_floats = (int64_t)g.n_slabs * g.ny * g.nz * kTile;
        g.part_b_floats = (int64_t)g.n_slabs * g.nz * kOT + kWgradHeaderFloats;
        return g;
    }
    // Slab count: about one workgroup per CU over all (slab, input-chunk, output-chunk) triples, at most 128
    // slabs (measured on MI355X, us for 128/64 slabs vs 256: N=17 080 O=128 I=128 22 vs 31; N=50 000 O=256 I=128
    // 52 vs 80 — fewer, longer slabs amortise the pipeline fill and halve the partial traffic).
    const int64_t tiles = ceil_div(I, kIT) * ceil_div(O, kOT);
    int64_t max_slabs = kMaxSlabs / tiles;
    // comb-shaped (I == O, an even number of output tiles): in the effective-weight form half of the output tiles hold the
    // few labeled rows only, so the other half gets twice the slabs to fill the chip
    if (I == O && ceil_div(O, kOT) % 2 == 0) max_slabs *= 2;
    const int64_t lim = max_slabs;
    if (max_slabs > 128) max_slabs = 128;
    // Small graphs: this kernel shares ONE launch with the data gradient of the same pair (dual_bwd_kernel: N/64 row
    // tiles + these slabs, two workgroups per CU).  Keep the sum within the 512 resident workgroups of the chip — a
    // handful of surplus workgroups would wait for a free slot and run as a second round (28 vs ~21 us at ppi_bp-shape).
    if (N <= 100000) {
        const int64_t room = (2 * 256 - 4 - ceil_div(N, 64)) / tiles;
        // ... and use all of them: inside the shared launch the weight-gradient workgroups are the longer ones (phase
        // stamps at ppi_bp-shape, trans pair: 126 slabs of 136 rows live 17.9 us beside data-gradient tiles done after
        // 14.6 us; 238 slabs of 72 rows: the launch 19.5 -> 17.5 us, the deferred reduction + 0.9 us)
        if (room >= 32) max_slabs = room < lim ? room : lim;
    }
    if (max_slabs < 32) max_slabs = 32;
    int64_t rows = ceil_div(N, max_slabs);
    if (rows < 64) rows = 64;
    rows = ceil_div(rows, 8) * 8;  // whole row pairs for each of the 4 waves
    // (the staged bodies of hidden 64's fused launches — one 128 x 64 tile per slab — walk 16-row stages, 32-row ones in the split form)
    if (N <= kFusedBwdMaxRows) rows = ceil_div(rows, tiles == 1 ? 32 : 16) * (tiles == 1 ? 32 : 16);
    g.rows_per_slab = (int)rows;
    g.n_slabs = (int)ceil_div(N, rows);
    g.ny = (int)ceil_div(I, kIT);
    g.nz = (int)ceil_div(O, kOT);
    g.part_w_floats = (int64_t)g.n_slabs * g.ny * g.nz * kTile;
    g.part_b_floats = (int64_t)g.n_slabs * g.nz * kOT + kWgradHeaderFloats;  // + the mode header behind the bias partials
    return g;
}

// The batched reduction carrying a K1 product whose plan holds workgroup items only (the selection product of the embedding
// backward: G = S^T dh, reference impl/models.py:248 backward): both only need the backward chain to be finished, so they
// share a launch — grid slices z < n_jobs reduce, the slices behind them run one plan item per workgroup.
struct SelJob {
    const int32_t *col, *items;
    const float *val, *X;
    int64_t ldx;
    float* Y;
    int64_t ldy;
    float* partials;
    int H, n_items;
};

template <int LPR>
__global__ __launch_bounds__(kBlock) void wgrad_reduce_sel_kernel(ReduceBatch batch, int n_jobs, SelJob sel) {
    __shared__ float4 lds[2 * kBlock];
    if ((int)blockIdx.z >= n_jobs) {
        const int item = (((int)blockIdx.z - n_jobs) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (item >= sel.n_items) return;
        // (blockIdx.y inside long_item_body is the column tile: LPR * 4 >= H here, one tile)
        const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
        const int grp = lane / LPR, sub = lane % LPR;
        const int coff = sub * 4;
        const bool col_ok = coff < sel.H;
        const int32_t* it = sel.items + 4 * (int64_t)item;
        const int row = it[0], eb = it[1], ee = it[2], slot = it[3];
        const int per = ((ee - eb + 4 * kWave - 1) / (4 * kWave)) * kWave;
        const int e0 = min(eb + w * per, ee), e1 = min(e0 + per, ee);
        float* lf = reinterpret_cast<float*>(lds);
        Vec<4> acc;
        acc.zero();
        gather_edges<4, LPR, 8, false>(acc, sel.col, sel.val, sel.X + coff, sel.ldx, e0, e1, lane, grp, col_ok);
        reduce_groups<4, LPR>(acc);
        if (grp == 0) acc.store(&lf[(w * LPR + sub) * 4]);
        __syncthreads();
        if (w == 0 && grp == 0 && col_ok) {
            Vec<4> s2, t;
            s2.load(&lf[sub * 4]);
#pragma unroll
            for (int k = 1; k < kBlock / kWave; ++k) {
                t.load(&lf[(k * LPR + sub) * 4]);
                s2.add(t);
            }
            float* dst = slot < 0 ? sel.Y + (int64_t)row * sel.ldy : sel.partials + (int64_t)slot * sel.H;
            s2.store(dst + coff);
        }
        return;
    }
    const ReduceJob& j = batch.job[blockIdx.z];
    if ((int)blockIdx.y >= j.ny * j.nz) return;
    if (j.n_l > 0) {
        wgrad_reduce_sl_body(j.part_w, j.part_b, j.n_slabs, j.n_l, j.dW, j.lddw, j.db, j.accumulate, lds);
        return;
    }
    if (j.n_l < 0) {
        wgrad_reduce_narrow_body(j.part_w, j.n_slabs, -j.n_l, j.O, j.I, j.dW, j.lddw, j.db, j.accumulate, lds);
        return;
    }
    wgrad_reduce_body(j.part_w, j.part_b, j.n_slabs, j.ny, j.nz, j.O, j.I, j.dW, j.lddw, j.db, j.accumulate, lds, blockIdx.y);
}

WgradSLGeom wgrad_sl_geom(int64_t N, int64_t lab_cap) {
    WgradSLGeom g;
    g.n_l = (int)ceil_div(lab_cap > 0 ? lab_cap : 1, 64);
    if (GLASS_COMB_BWD_V2 && N <= kFusedBwdMaxRows) {
        // staged backward (comb_bwd_eff2_kernel, dense.hip): every 64-row tile of the data gradient writes its own S tile
        g.rows_per_slab = 64;
        g.n_s = (int)ceil_div(N, 64);
        g.part_w_floats = (int64_t)(g.n_s + g.n_l) * kTile;
        g.part_b_floats = (int64_t)(g.n_s + g.n_l) * kSLOut + kWgradHeaderFloats;
        return g;
    }
    // small graphs: the slabs share ONE launch with the data gradient's row tiles (comb_bwd_eff_kernel: N/64 + n_l row
    // tiles, two workgroups per CU) — keep the sum within the 512 resident workgroups of the chip
    int64_t room = N <= 100000 ? 2 * 256 - 4 - ceil_div(N, 64) - 2 * g.n_l : 256;
    if (room > 256) room = 256;
    if (room < 32) room = 32;
    int64_t rows = ceil_div(N, room);
    if (rows < 64) rows = 64;
    rows = ceil_div(rows, 8) * 8;  // whole row pairs for each of the 4 waves
    if (N <= kFusedBwdMaxRows) rows = ceil_div(rows, 32) * 32;  // whole 32-row stages of the staged body's split form (dense.hip)
    g.rows_per_slab = (int)rows;
    g.n_s = (int)ceil_div(N, rows);
    g.part_w_floats = (int64_t)(g.n_s + g.n_l) * kTile;
    g.part_b_floats = (int64_t)(g.n_s + g.n_l) * kSLOut + kWgradHeaderFloats;
    return g;
}

// stand-alone launch of the S / L partials (large graphs; small ones run them inside comb_bwd_eff_kernel, dense.hip)
__global__ __launch_bounds__(kBlock, 1) void wgrad_sl_kernel(WgradSL a, int64_t N, float zr, float* __restrict__ part_w,
                                                             float* __restrict__ part_b) {
    __shared__ float lds[2 * kTile + 8 * kSLOut];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float* header = part_b + (int64_t)(a.n_s + a.n_l) * kSLOut;
        header[0] = 2.f;
        header[1] = zr;
        header[2] = 0.f;  // (tiles in the permuted accumulator order)
    }
    wgrad_sl_body<4>(a, N, blockIdx.x, part_w, part_b, lds, lds + 2 * kTile);
}

void launch_wgrad_sl(const WgradSL& a, int64_t N, float zr, float* part_w, float* part_b, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_sl_kernel, dim3((unsigned)(a.n_s + a.n_l)), dim3(kBlock), 0, st, a, N, zr, part_w, part_b);
}

// ---- fused Adam over the flat parameter arena --------------------------------------------------
// torch.optim.Adam (single-tensor formulation, amsgrad=False, maximize=False):
//   m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// step_dev = int64[2]: (steps completed, ticket).  Every workgroup reads the count first and takes a ticket last;
// the workgroup that takes the last ticket publishes count + 1 and clears the ticket — the bias correction needs
// no counter kernel of its own (a dependent launch of ~4 us on a ~500 us step).
__global__ __launch_bounds__(kBlock) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                      const float* __restrict__ lr_dev, float beta1, float beta2,
                                                      float eps, float weight_decay,
                                                      int64_t* __restrict__ step_dev) {
    const int64_t step_now = step_dev[0] + 1;
    const AdamCoef c = adam_coef(step_now, lr_dev[0], beta1, beta2, eps, weight_decay);
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock)
        adam_update(c, p, g[k], m, v, k);
    adam_ticket(step_dev, step_now);
}

}  // namespace glass

using namespace glass;

extern "C" int64_t glass_linear_wgrad_ws_bytes(int64_t N, int64_t O, int64_t I) {
    if (N <= 0 || O <= 0 || I <= 0) return GLASS_E_ARG;
    const WgradGeom g = wgrad_geom(N, O, I);
    int64_t floats = g.part_w_floats + g.part_b_floats;
    if (O <= kThinMaxO && (int64_t)thin_blocks(N) * (O * I + O) > floats) floats = (int64_t)thin_blocks(N) * (O * I + O);
    if (O <= 2 * 32 && I <= 2 * 32) {  // narrow pairs (dense_narrow.hip)
        int n_slabs, stride;
        narrow_wgrad_geom(N, O, I, &n_slabs, &stride);
        if ((int64_t)n_slabs * stride > floats) floats = (int64_t)n_slabs * stride;
    }
    if (wgrad_tiled_shape(N, O, I)) {  // glass_dual_linear_wgrad_f32 takes the tiled kernel there: room for either geometry
        const TiledWgradGeom t = wgrad_tiled_geom(N, O, I);
        if (t.part_w_floats + t.part_b_floats > floats) floats = t.part_w_floats + t.part_b_floats;
    }
    return floats * (int64_t)sizeof(float);
}

extern "C" int glass_linear_wgrad_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t N, int64_t O,
                                      int64_t I, float* dW, int64_t lddw, float* db, int accumulate, void* ws,
                                      void* stream) {
    GLASS_REQUIRE(G && X && dW && ws, "linear_wgrad: null pointer");
    GLASS_REQUIRE(N > 0 && O > 0 && I > 0 && ldg >= O && ldx >= I && lddw >= I, "linear_wgrad: bad sizes");
    if (O <= kThinMaxO && I % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(ws)) {  // thin outputs (hidden -> 1 heads)
        hipStream_t st = (hipStream_t)stream;
        const int nb = thin_blocks(N);
        const int rows = (int)ceil_div(N, nb);
        const int tc = pow2_ceil_cap(I / 4, 64);
        int tl = 0;
        while ((1 << tl) < tc) ++tl;
        float* part = (float*)ws;
        if (O == 1)
            hipLaunchKernelGGL(wgrad_thin_partial_kernel<1>, dim3(nb), dim3(kBlock), 0, st, G, ldg, X, ldx, N, (int)I, rows, tl, part);
        else if (O == 2)
            hipLaunchKernelGGL(wgrad_thin_partial_kernel<2>, dim3(nb), dim3(kBlock), 0, st, G, ldg, X, ldx, N, (int)I, rows, tl, part);
        else
            hipLaunchKernelGGL(wgrad_thin_partial_kernel<3>, dim3(nb), dim3(kBlock), 0, st, G, ldg, X, ldx, N, (int)I, rows, tl, part);
        hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)ceil_div(O * I + O, (int64_t)4)), dim3(kBlock), 0, st, part,
                           nb, (int)O, (int)I, dW, lddw, db, accumulate);
        return launch_status("glass_linear_wgrad_f32");
    }
    if (O % 4 || I % 2 || ldg % 4 || ldx % 2 || !aligned16(G) || (reinterpret_cast<uintptr_t>(X) & 7u)) {
        set_error("linear_wgrad: needs O%%4==0, I%%2==0, ldg%%4==0, ldx%%2==0 and 16-B/8-B aligned G/X "
                  "(O=%lld I=%lld ldg=%lld ldx=%lld)", (long long)O, (long long)I, (long long)ldg, (long long)ldx);
        return GLASS_E_UNSUPPORTED;  // caller falls back to a library GEMM for odd shapes
    }
    hipStream_t st = (hipStream_t)stream;
    const WgradGeom g = wgrad_geom(N, O, I);
    float* part_w = (float*)ws;
    float* part_b = part_w + g.part_w_floats;
    float* header = part_b + g.part_b_floats - kWgradHeaderFloats;
    hipLaunchKernelGGL((wgrad_partial_kernel<false, false>), dim3(g.n_slabs, g.ny, g.nz), dim3(kBlock), 0, st, G, ldg, X, ldx, N,
                       (int)O, (int)I, g.rows_per_slab, part_w, db ? part_b : nullptr, header, WgradSynth{});
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((kTile + kOT) / 64, g.ny * g.nz), dim3(kBlock), 0, st,
                       part_w, part_b, g.n_slabs, g.ny, g.nz, (int)O, (int)I, dW, lddw, db, accumulate);
    return launch_status("glass_linear_wgrad_f32");
}

extern "C" int glass_dual_linear_wgrad_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt,
                                           const uint8_t* mask, double z_ratio, int act, const float* X, int64_t ldx,
                                           const float* X2, int64_t ldx2, int64_t N, int64_t H, float* dW,
                                           int64_t lddw, float* db, int accumulate, void* ws, void* stream) {
    const CallOptions call_options(act);  // act word -> activation code + this call's options (dense_common.h)
    GLASS_REQUIRE(dsrc && mask && X && ws && N > 0 && H > 0, "dual_linear_wgrad: null pointer");
    const int64_t O = 2 * H, I = X2 ? 2 * H : H;
    GLASS_REQUIRE(ldd >= H && ldx >= H && (!dW || lddw >= I) && (!X2 || ldx2 >= H) && (act == GLASS_ACT_NONE || (T && ldt >= O)),
                  "dual_linear_wgrad: bad sizes");
    if (narrow_shape_ok(H)) {  // hidden <= 32: thread-owned outputs over 256-row slabs (dense_narrow.hip), any alignment
        const WgradSynth nsy{dsrc, ldd, act != GLASS_ACT_NONE ? T : nullptr, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio),
                             act, (int)H, X2, ldx2};
        int rc = launch_narrow_wgrad(nsy, X, ldx, N, O, I, (float*)ws, (hipStream_t)stream);
        if (rc || !dW) return rc;
        const void* wsp = ws;
        const int32_t accum = accumulate;
        return glass_linear_wgrad_reduce_batch_f32(1, &wsp, &N, &O, &I, &dW, &lddw, &db, &accum, nullptr, stream);
    }
    if (H % 64 || ldd % 4 || ldx % 2 || (X2 && ldx2 % 2) || !aligned16(dsrc) || (reinterpret_cast<uintptr_t>(X) & 7u) ||
        (X2 && (reinterpret_cast<uintptr_t>(X2) & 7u)) || (act != GLASS_ACT_NONE && (ldt % 4 || !aligned16(T)))) {
        set_error("dual_linear_wgrad: needs H%%64==0 and aligned operands (H=%lld)", (long long)H);
        return GLASS_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const WgradSynth sy{dsrc, ldd, act != GLASS_ACT_NONE ? T : nullptr, ldt, mask, (float)z_ratio, (float)(1.0 - z_ratio),
                        act, (int)H, X2, ldx2};
    if (wgrad_tiled_shape(N, O, I)) {  // wide layer on a large graph: LDS-tiled kernel (wgrad_tiled.hip)
        if (ldx % 4 || !aligned16(X) || (X2 && (ldx2 % 4 || !aligned16(X2)))) {
            set_error("dual_linear_wgrad: the tiled kernel needs ld %% 4 == 0 and 16-B aligned inputs");
            return GLASS_E_UNSUPPORTED;
        }
        if (reinterpret_cast<uintptr_t>(mask) & 1u) {  // (the eight-wave kernel reads a thread's two label bytes as one 16-bit load)
            set_error("dual_linear_wgrad: the tiled kernel needs a 2-byte aligned label mask");
            return GLASS_E_UNSUPPORTED;
        }
        const TiledWgradGeom t = wgrad_tiled_geom(N, O, I);
        {  // 32-bit buffer offsets: a slab plus the whole triples of steps read past it must stay below 2^31 bytes on every operand
            int64_t ldmax = ldd > ldx ? ldd : ldx;
            if (act != GLASS_ACT_NONE && ldt > ldmax) ldmax = ldt;
            if (X2 && ldx2 > ldmax) ldmax = ldx2;
            if (((int64_t)t.rows_per_slab + 6 * 16) * ldmax * 4 >= (int64_t)1 << 31) {
                set_error("dual_linear_wgrad: leading dimension %lld too wide for the tiled kernel's 32-bit slab offsets",
                          (long long)ldmax);
                return GLASS_E_UNSUPPORTED;
            }
        }
        float* pw = (float*)ws;
        launch_tiled_wgrad_partial(X, ldx, N, O, I, sy, pw, (db || !dW) ? pw + t.part_w_floats : nullptr, st);
        if (dW) launch_tiled_wgrad_reduce(pw, pw + t.part_w_floats, N, O, I, dW, lddw, db, accumulate, st);
        return launch_status("glass_dual_linear_wgrad_f32 (tiled)");
    }
    const WgradGeom g = wgrad_geom(N, O, I);
    float* part_w = (float*)ws;
    float* part_b = part_w + g.part_w_floats;
    float* header = part_b + g.part_b_floats - kWgradHeaderFloats;
    // comb pair (virtual concatenation as the input, no activation factor) with an even number of output tiles: the
    // effective-weight form of the partials (wgrad_common.h) — half the matrix work; the labeled-row list of a slab must
    // fit the workgroup's LDS image area
    const bool eff = X2 != nullptr && act == GLASS_ACT_NONE && I == O && O == 2 * H && g.nz % 2 == 0 &&
                     g.rows_per_slab <= 2 * kTile - 64;
    float* pb_arg = (db || !dW) ? part_b : nullptr;
    const dim3 grid(g.n_slabs, g.ny, g.nz);
    // hidden 128 (the widths of the tiled family whose graph or layer is too small for wgrad_tiled.hip): the product form of
    // that family (this call's options) applies here too — six bf16 partial products per fp32 product
    const bool split = tiled_shape_ok(H) && tiled_split_products() && ldx % 2 == 0;
    if (split && !eff && X2 == nullptr && wgrad128_shape(N, O, I) && ldx % 4 == 0 && aligned16(X) &&
        N * std::max(std::max(ldd, ldx), act != GLASS_ACT_NONE ? ldt : (int64_t)0) * 4 < (1ll << 31))
        // rows shared through LDS, one workgroup per slab (32-bit buffer offsets: checked above)
        launch_wgrad128_trans(X, ldx, N, g.rows_per_slab, g.n_slabs, part_w, pb_arg, header, sy, st);
    else if (split && eff && wgrad128_comb_shape(N, O, I) && ldx % 4 == 0 && ldx2 % 4 == 0 && aligned16(X) && aligned16(X2) &&
             N * std::max(std::max(ldd, ldx), ldx2) * 4 < (1ll << 31))
        // S tile per slab with the rows shared through LDS + list tiles for the labeled rows (wgrad128.hip)
        launch_wgrad128_comb(dsrc, ldd, X, ldx, X2, ldx2, mask, N, g.rows_per_slab, g.n_slabs, (float)z_ratio, part_w, pb_arg, header, st);
    else if (split && eff)
        hipLaunchKernelGGL((wgrad_partial_split_kernel<false, true>), grid, dim3(kBlock), 0, st, X, ldx, N, (int)O, (int)I,
                           g.rows_per_slab, part_w, pb_arg, header, sy);
    else if (split && act != GLASS_ACT_NONE)
        hipLaunchKernelGGL((wgrad_partial_split_kernel<true, false>), grid, dim3(kBlock), 0, st, X, ldx, N, (int)O, (int)I,
                           g.rows_per_slab, part_w, pb_arg, header, sy);
    else if (split)
        hipLaunchKernelGGL((wgrad_partial_split_kernel<false, false>), grid, dim3(kBlock), 0, st, X, ldx, N, (int)O, (int)I,
                           g.rows_per_slab, part_w, pb_arg, header, sy);
    else if (eff)
        hipLaunchKernelGGL((wgrad_partial_kernel<true, true>), grid, dim3(kBlock), 0, st, nullptr, 0, X,
                           ldx, N, (int)O, (int)I, g.rows_per_slab, part_w, pb_arg, header, sy);
    else
        hipLaunchKernelGGL((wgrad_partial_kernel<true, false>), grid, dim3(kBlock), 0, st, nullptr, 0, X,
                           ldx, N, (int)O, (int)I, g.rows_per_slab, part_w, pb_arg, header, sy);
    if (dW)  // dW == NULL: partial sums only; the caller reduces later with glass_linear_wgrad_reduce_batch_f32
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((kTile + kOT) / 64, g.ny * g.nz), dim3(kBlock), 0, st, part_w,
                           part_b, g.n_slabs, g.ny, g.nz, (int)O, (int)I, dW, lddw, db, accumulate);
    return launch_status("glass_dual_linear_wgrad_f32");
}

static int reduce_batch_impl(int64_t n_jobs, const void* const* ws, const int64_t* N, const int64_t* O, const int64_t* I,
                             float* const* dW, const int64_t* lddw, float* const* db, const int32_t* accumulate,
                             const int64_t* lab_cap, const SelJob* sel, void* stream) {
    GLASS_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || (ws && N && O && I && dW && lddw && db && accumulate)),
                  "wgrad_reduce_batch: null pointer");
    if (n_jobs == 0 && !sel) return 0;
    hipStream_t st = (hipStream_t)stream;
    bool sel_done = sel == nullptr;
    for (int64_t j0 = 0; j0 < n_jobs || !sel_done; j0 += kMaxReduceJobs) {
        const int nj = (int)(n_jobs - j0 < kMaxReduceJobs ? (n_jobs - j0 > 0 ? n_jobs - j0 : 0) : kMaxReduceJobs);
        ReduceBatch b;
        int max_chunks = 0;
        for (int k = 0; k < kMaxReduceJobs; ++k) b.job[k] = ReduceJob{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, nullptr};
        for (int k = 0; k < nj; ++k) {
            const int64_t j = j0 + k;
            GLASS_REQUIRE(ws[j] && dW[j] && N[j] > 0 && O[j] > 0 && I[j] > 0 && lddw[j] >= I[j],
                          "wgrad_reduce_batch: bad job %lld", (long long)j);
            if (lab_cap && lab_cap[j] > 0) {  // S / L partials of glass_comb_eff_bwd_f32 (hidden 64: O = I = 128)
                GLASS_REQUIRE(O[j] == 2 * kSLOut && I[j] == 2 * kSLOut && db[j], "wgrad_reduce_batch: job %lld is not a comb pair of hidden 64", (long long)j);
                const WgradSLGeom g = wgrad_sl_geom(N[j], lab_cap[j]);
                const float* part_w = (const float*)ws[j];
                b.job[k] = ReduceJob{part_w, part_w + g.part_w_floats, g.n_s, 1, 1, (int)O[j], (int)I[j], accumulate[j], g.n_l,
                                     dW[j], lddw[j], db[j]};
                if (max_chunks < 1) max_chunks = 1;
                continue;
            }
            if (O[j] < 128 && narrow_shape_ok(O[j] / 2)) {  // hidden <= 32: plain slab partials of dense_narrow.hip
                int n_slabs, stride;
                narrow_wgrad_geom(N[j], O[j], I[j], &n_slabs, &stride);
                b.job[k] = ReduceJob{(const float*)ws[j], nullptr, n_slabs, 1, 1, (int)O[j], (int)I[j], accumulate[j], -stride,
                                     dW[j], lddw[j], db[j]};
                if (max_chunks < 1) max_chunks = 1;
                continue;
            }
            if (wgrad_tiled_shape(N[j], O[j], I[j])) {  // partials of the tiled kernel (what glass_dual_linear_wgrad_f32
                                                        // writes at this shape): their own reduce launch
                const TiledWgradGeom t = wgrad_tiled_geom(N[j], O[j], I[j]);
                const float* pw = (const float*)ws[j];
                launch_tiled_wgrad_reduce(pw, pw + t.part_w_floats, N[j], O[j], I[j], dW[j], lddw[j], db[j], accumulate[j], st);
                continue;  // b.job[k] stays empty (ny * nz = 0: its blocks return at once)
            }
            const WgradGeom g = wgrad_geom(N[j], O[j], I[j]);
            const float* part_w = (const float*)ws[j];
            b.job[k] = ReduceJob{part_w, part_w + g.part_w_floats, g.n_slabs, g.ny, g.nz, (int)O[j], (int)I[j],
                                 accumulate[j], 0, dW[j], lddw[j], db[j]};
            if (g.ny * g.nz > max_chunks) max_chunks = g.ny * g.nz;
        }
        const unsigned gx = (kTile + kOT) / 64;
        if (!sel_done) {  // the product's items ride behind this batch of jobs
            sel_done = true;
            const unsigned gy = (unsigned)(max_chunks > 0 ? max_chunks : 1);
            const unsigned zs = (unsigned)ceil_div(sel->n_items, (int64_t)gx * gy);
            const dim3 grid(gx, gy, (unsigned)nj + zs);
            if (sel->H <= 64)
                hipLaunchKernelGGL((wgrad_reduce_sel_kernel<16>), grid, dim3(kBlock), 0, st, b, nj, *sel);
            else if (sel->H <= 128)
                hipLaunchKernelGGL((wgrad_reduce_sel_kernel<32>), grid, dim3(kBlock), 0, st, b, nj, *sel);
            else
                hipLaunchKernelGGL((wgrad_reduce_sel_kernel<64>), grid, dim3(kBlock), 0, st, b, nj, *sel);
        } else if (max_chunks > 0) {
            hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(gx, max_chunks, nj), dim3(kBlock), 0, st, b);
        }
    }
    return launch_status("glass_linear_wgrad_reduce_batch_f32");
}

extern "C" int glass_linear_wgrad_reduce_batch_f32(int64_t n_jobs, const void* const* ws, const int64_t* N,
                                                   const int64_t* O, const int64_t* I, float* const* dW,
                                                   const int64_t* lddw, float* const* db, const int32_t* accumulate,
                                                   const int64_t* lab_cap, void* stream) {
    return reduce_batch_impl(n_jobs, ws, N, O, I, dW, lddw, db, accumulate, lab_cap, nullptr, stream);
}

// The batched reduction + a K1 product Y = M @ X in ONE launch when M's plan holds workgroup items only (no sweep items:
// the selection matrix of the embedding backward) and X is 16-B aligned with H % 4 == 0, H <= 256; otherwise the two calls
// one after the other.  The plan's reduce launch (rows cut into several chunks) follows as in glass_spmm_csr_f32 — pass a
// header copy with the reduce count zeroed when the consumer sums the partial rows itself.
extern "C" int glass_wgrad_reduce_spmm_f32(int64_t n_jobs, const void* const* ws, const int64_t* N, const int64_t* O,
                                           const int64_t* I, float* const* dW, const int64_t* lddw, float* const* db,
                                           const int32_t* accumulate, const int64_t* lab_cap, const int32_t* rowptr,
                                           const int32_t* col, const float* val, const float* X, int64_t ldx, float* Y,
                                           int64_t ldy, int64_t n_rows, int64_t H, const int32_t* hdr,
                                           const int32_t* plan_dev, void* ws_spmm, void* stream) {
    GLASS_REQUIRE(hdr && plan_dev && X && Y, "wgrad_reduce_spmm: null pointer");
    const bool fusable = hdr[H_MAGIC] == kPlanMagic && hdr[H_VER] == kPlanVersion && hdr[H_NROWS] == n_rows &&
                         hdr[H_NSWEEP] == 0 && hdr[H_NLONG] > 0 && H % 4 == 0 && H <= 256 && ldx % 4 == 0 && ldy % 4 == 0 &&
                         aligned16(X) && aligned16(Y) && col && val && (hdr[H_NSLOTS] == 0 || (ws_spmm && aligned16(ws_spmm)));
    if (!fusable) {
        const int rc = glass_linear_wgrad_reduce_batch_f32(n_jobs, ws, N, O, I, dW, lddw, db, accumulate, lab_cap, stream);
        return rc ? rc : glass_spmm_csr_f32(rowptr, col, val, X, ldx, Y, ldy, n_rows, H, hdr, plan_dev, ws_spmm, stream);
    }
    const SelJob sel{col, plan_dev + hdr[H_OFF_LONG], val, X, ldx, Y, ldy, (float*)ws_spmm, (int)H, hdr[H_NLONG]};
    const int rc = reduce_batch_impl(n_jobs, ws, N, O, I, dW, lddw, db, accumulate, lab_cap, &sel, stream);
    if (rc) return rc;
    if (hdr[H_NREDUCE] > 0)  // rows cut into several chunks: the plan's own reduce launch
        return glass_spmm_reduce_rows_f32((const float*)ws_spmm, Y, ldy, H, plan_dev + hdr[H_OFF_REDUCE], hdr[H_NREDUCE], stream);
    return 0;
}

extern "C" int glass_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                                   int64_t* step_dev, void* stream) {
    GLASS_REQUIRE(param && grad && exp_avg && exp_avg_sq && lr_dev && step_dev && n > 0, "adam_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = ceil_div(n, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, param, grad, exp_avg, exp_avg_sq, n,
                       lr_dev, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, step_dev);
    return launch_status("glass_adam_step_f32");
}
