// Shared by the two weight-gradient kernel families: linear.hip (operands straight from global memory, 128 x 64 tile,
// small graphs / narrow layers) and wgrad_tiled.hip (LDS-tiled 128 x 256 tile: hidden >= 256 on large graphs).
#pragma once
#include "common.h"
#include "split_mma.h"
#include "dense_common.h"

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

struct WgradGeom {
    int n_slabs, rows_per_slab, ny, nz;
    int64_t part_w_floats, part_b_floats;
};
WgradGeom wgrad_geom(int64_t N, int64_t O, int64_t I);  // linear.hip
// wgrad128.hip: the trans pair of hidden 128 with the slab's rows shared through LDS (one workgroup per slab, all four tiles)
bool wgrad128_shape(int64_t N, int64_t O, int64_t I);
struct WgradSynth;
void launch_wgrad128_trans(const float* X, int64_t ldx, int64_t N, int rows_per_slab, int n_slabs, float* part_w, float* part_b,
                           float* header, const WgradSynth& sy, hipStream_t st);
bool wgrad128_comb_shape(int64_t N, int64_t O, int64_t I);
int wgrad128_comb_lists(int64_t N);
void launch_wgrad128_comb(const float* dc, int64_t ldd, const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* mask,
                          int64_t N, int rows_per_slab, int n_slabs, float zr, float* part_w, float* part_b, float* header,
                          hipStream_t st);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kOT = 128;            // outputs per workgroup (4 tiles of 32, strided by 4)
constexpr int kIT = 64;             // inputs per workgroup  (2 tiles of 32, strided by 2)
constexpr int kTile = kOT * kIT;    // 8192 accumulators per workgroup
constexpr int kMaxSlabs = 256;

// idx of accumulator (t,u,reg,lane) in the permuted partial layout
__device__ __forceinline__ int acc_index(int t, int u, int reg, int lane) { return ((t * 2 + u) * 16 + reg) * 64 + lane; }

// The end of a partial-sum workgroup: the four waves' accumulator images combined through LDS (waves 0,1 store; waves 2,3
// add; everyone sums the pair), the partial tile and the bias partial written.
__device__ __forceinline__ void wgrad_partial_tail(f32x16 (&acc)[4][2], const float4& bsum, bool lab_tile, int bx, int by, int bz,
                                                   int gx, int gy, float* __restrict__ part_w, float* __restrict__ part_b,
                                                   float* lds, float* lds_b) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    // ---- combine the 4 waves through LDS: waves 0,1 store; waves 2,3 add; everyone sums the pair ----
    if (lab_tile) __syncthreads();  // the row list lies in the image area: every wave is done reading it
    if (w < 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[w * kTile + acc_index(t, u, k, lane)] = acc[t][u][k];
    }
    *reinterpret_cast<float4*>(&lds_b[(w * 2 + h) * kOT + 4 * c]) = bsum;
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[(w - 2) * kTile + acc_index(t, u, k, lane)] += acc[t][u][k];
    }
    __syncthreads();
    const int64_t tile_id = ((int64_t)bz * gy + by) * gx + bx;
    float* pw = part_w + tile_id * kTile;
    for (int k = threadIdx.x * 4; k < kTile; k += kBlock * 4) {
        const float4 a = *reinterpret_cast<const float4*>(&lds[k]);
        const float4 b = *reinterpret_cast<const float4*>(&lds[kTile + k]);
        *reinterpret_cast<float4*>(pw + k) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    if (by == 0 && part_b && threadIdx.x < kOT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds_b[k * kOT + threadIdx.x];
        part_b[((int64_t)bz * gx + bx) * kOT + threadIdx.x] = s;
    }
}

// Partial sums of one (slab bx, input tile by, output tile bz) of a gx x gy x gz launch, by one 256-thread workgroup.
// lds: 2 * kTile floats (two wave-sized accumulator images), lds_b: 8 * kOT floats (bias partials [wave*2 + h][o]).
// kStages: pipeline stages of raw loads kept in flight (4 on large graphs; 2 in the fused backward launch of small
// graphs, where a slab holds only a few stages and registers decide whether two workgroups share a CU).
// EFF (comb pair, no activation factor; gz = number of output tiles, even): effective-weight form of the partials, as in
// wgrad_tiled.hip — output tiles bz < gz/2 hold S = sum over ALL rows of dc^T X (coefficient 1), tiles bz >= gz/2 hold
// L = the same sum over the slab's LABELED rows, found once per workgroup (ordered list in LDS) and walked through the
// same pipeline; the reduce kernels form dW1 = (1-z) S + (2z-1) L, dW0 = z S - (2z-1) L from the mode header.
constexpr int kWgradHeaderFloats = 4;  // behind the bias partials: [0] = 1.0f in effective-weight form, [1] = z_ratio
template <bool SYNTH, int kStages, bool EFF = false>
__device__ __forceinline__ void wgrad_partial_body(const float* __restrict__ G, int64_t ldg,
                                                   const float* __restrict__ X, int64_t ldx, int64_t N, int O, int I,
                                                   int rows_per_slab, float* __restrict__ part_w,
                                                   float* __restrict__ part_b, const WgradSynth& sy, int bx, int by, int bz,
                                                   int gx, int gy, float* lds, float* lds_b, int gz = 0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const bool lab_tile = EFF && bz >= gz / 2;
    const int o0 = (lab_tile ? bz - gz / 2 : bz) * kOT + 4 * c;   // this lane's 4 outputs
    const int i0 = by * kIT + 2 * c;   // this lane's 2 inputs
    const bool o_ok = o0 < O, i_ok = i0 < I;   // O % 4 == 0 and I % 2 == 0 (checked on the host)
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = min(N, r0 + rows_per_slab);

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][u][k] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // Wave w takes row pairs p = w, w+4, ...  One pipeline stage = two row pairs (16 MFMAs = 1024
    // cycles of matrix work); kStages stages are kept in flight because a stage's loads take about
    // one loaded-memory latency (~2 us) — with a single stage of prefetch the loop ran 4x slower
    // than the MFMA rate at N = 1 M (profiles/r01: 6.9 ms vs 1.7 ms of matrix time).
    // A stage holds RAW loads only (rows past the slab are clamped to its last row and zeroed at use),
    // so that no load has to be waited for when it is issued; the mix / ELU' synthesis of G happens
    // right before the MFMAs.  (Synthesising at load time put an s_waitcnt vmcnt(0) into every stage.)
    struct Stage {
        float4 g[2], t[2];
        float2 x[2];
        int mk[2];
        bool live[2];
    };
    Stage st[kStages];
    const bool first = o0 < sy.H;  // SYNTH: this lane's four outputs lie in the f1 half
    // EFF, labeled-rows tile: the slab's labeled rows in row order (LDS, over the accumulator images, which are only
    // written after the loop); the pipeline then walks list positions instead of rows
    int* lab_list = reinterpret_cast<int*>(lds);
    int n_lab = 0;
    if (lab_tile) {
        int* cnt = lab_list + 2 * kTile - 8;  // per-wave counts at the far end of the image area
        const int span = (int)(r1 - r0);
        int base = 0;
        for (int c0 = 0; c0 < span; c0 += kBlock) {  // chunks of 256 rows, ordered compaction by wave ballots
            const int64_t n = r0 + c0 + (int)threadIdx.x;
            const bool flag = c0 + (int)threadIdx.x < span && sy.mask[n] != 0;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) cnt[w] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int ww = 0; ww < w; ++ww) off += cnt[ww];
            if (flag) lab_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = c0 + (int)threadIdx.x;
            base += cnt[0] + cnt[1] + cnt[2] + cnt[3];
            __syncthreads();
        }
        n_lab = base;
    }
    const int64_t r_end = lab_tile ? r0 + n_lab : r1;  // the pipeline's row counter runs over [r0, r_end)
    auto load_stage = [&](int64_t nb, Stage& S) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t pos = nb + h + 8 * s;
            const bool in = pos < r_end;
            const int64_t want = !lab_tile ? pos : (in ? r0 + lab_list[pos - r0] : r1);
            const int64_t nn = want < r1 ? want : r1 - 1;
            S.live[s] = in && want < r1;
            S.g[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            S.x[s] = make_float2(0.f, 0.f);
            if (!SYNTH) {
                if (o_ok) S.g[s] = *reinterpret_cast<const float4*>(G + nn * ldg + o0);
                if (i_ok) S.x[s] = *reinterpret_cast<const float2*>(X + nn * ldx + i0);
            } else {  // O = 2H and I are multiples of the tile sizes: every lane is in range
                S.g[s] = *reinterpret_cast<const float4*>(sy.dsrc + nn * sy.ldd + (first ? o0 : o0 - sy.H));
                if (sy.act != GLASS_ACT_NONE) S.t[s] = *reinterpret_cast<const float4*>(sy.T + nn * sy.ldt + o0);
                S.mk[s] = sy.mask[nn];
                S.x[s] = (i0 < sy.H || sy.X2 == nullptr)
                             ? *reinterpret_cast<const float2*>(X + nn * ldx + i0)
                             : *reinterpret_cast<const float2*>(sy.X2 + nn * sy.ldx2 + (i0 - sy.H));
            }
        }
    };
    const int64_t nb0 = r0 + 2 * w;  // wave-uniform (MFMA needs every lane in the loop)
#pragma unroll
    for (int k = 0; k < kStages; ++k) load_stage(nb0 + 16 * k, st[k]);
    for (int64_t nb = nb0; nb < r_end; nb += 16 * kStages) {
#pragma unroll
        for (int k = 0; k < kStages; ++k) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float4 g = st[k].g[s];
                if (SYNTH && !EFF) {
                    const float cf = ((st[k].mk[s] != 0) == first) ? sy.zr : sy.omz;
                    g.x *= cf; g.y *= cf; g.z *= cf; g.w *= cf;
                    if (sy.act != GLASS_ACT_NONE) {
                        const float4 t = st[k].t[s];
                        g.x *= act_grad(sy.act, t.x); g.y *= act_grad(sy.act, t.y); g.z *= act_grad(sy.act, t.z); g.w *= act_grad(sy.act, t.w);
                        }
                }
                if (!st[k].live[s]) g = make_float4(0.f, 0.f, 0.f, 0.f);
                const float gv[4] = {g.x, g.y, g.z, g.w};
                const float xv[2] = {st[k].x[s].x, st[k].x[s].y};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[t], xv[u], acc[t][u], 0, 0, 0);
                bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the refill below from being sunk into later stages
            load_stage(nb + 16 * (k + kStages), st[k]);  // refill this stage, kStages ahead
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    wgrad_partial_tail(acc, bsum, lab_tile, bx, by, bz, gx, gy, part_w, part_b, lds, lds_b);
}

// ---- the same partial sums with split products (split_mma.h) --------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 sums over 16 node rows per instruction: lane (c, h) feeds rows 8h .. 8h+7 of a 16-row group for
// its four outputs / two inputs, so a wave takes WHOLE 16-row groups (w, w + 4, ...) instead of row pairs, loads eight rows
// per lane and stage (float4 of the gradient, of the pre-activation when there is an activation, float2 of the input — the
// same whole-row coalesced loads as above), synthesises dZ, cuts every 8-row column into three bf16 pieces in registers and
// runs 8 tiles x 6 partial products = 48 MFMAs (1 536 cycles) per group where the f32-input form runs 64 x 64 cycles for
// the same 16 rows.  No LDS in the loop; two stages of raw loads in flight (refilled as soon as their values are cut).
// Accumulator layout, partial layout, labeled-rows tiles and the workgroup tail are those of wgrad_partial_body.
template <bool ACT, bool EFF>
__device__ __forceinline__ void wgrad_partial_split_body(const float* __restrict__ X, int64_t ldx, int64_t N, int O, int I,
                                                         int rows_per_slab, float* __restrict__ part_w,
                                                         float* __restrict__ part_b, const WgradSynth& sy, int bx, int by,
                                                         int bz, int gx, int gy, float* lds, float* lds_b, int gz) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const bool lab_tile = EFF && bz >= gz / 2;
    const int o0 = (lab_tile ? bz - gz / 2 : bz) * kOT + 4 * c;   // this lane's 4 outputs
    const int i0 = by * kIT + 2 * c;                              // this lane's 2 inputs
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = min(N, r0 + rows_per_slab);

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][u][k] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool first = o0 < sy.H;  // this lane's four outputs lie in the f1 half
    const float e_neg = sy.act == GLASS_ACT_ELU ? 1.f : 0.f;  // act'(t) = t > 0 ? 1 : e_neg * exp(t)  (ELU / ReLU)
    // EFF, labeled-rows tile: the slab's labeled rows in row order (as wgrad_partial_body)
    int* lab_list = reinterpret_cast<int*>(lds);
    int n_lab = 0;
    if (lab_tile) {
        int* cnt = lab_list + 2 * kTile - 8;
        const int span = (int)(r1 - r0);
        int base = 0;
        for (int c0 = 0; c0 < span; c0 += kBlock) {
            const int64_t n = r0 + c0 + (int)threadIdx.x;
            const bool flag = c0 + (int)threadIdx.x < span && sy.mask[n] != 0;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) cnt[w] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int ww = 0; ww < w; ++ww) off += cnt[ww];
            if (flag) lab_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = c0 + (int)threadIdx.x;
            base += cnt[0] + cnt[1] + cnt[2] + cnt[3];
            __syncthreads();
        }
        n_lab = base;
    }
    const int64_t r_end = lab_tile ? r0 + n_lab : r1;  // the row counter runs over [r0, r_end)
    const float* xsrc = (i0 < sy.H || sy.X2 == nullptr) ? X + i0 : sy.X2 + (i0 - sy.H);
    const int64_t xld = (i0 < sy.H || sy.X2 == nullptr) ? ldx : sy.ldx2;
    const float* gsrc = sy.dsrc + (first ? o0 : o0 - sy.H);
    struct Stage {
        float4 g[8], t[ACT ? 8 : 1];
        float2 x[8];
        int mk[EFF ? 1 : 8];
        int nlive;  // rows 0 .. nlive-1 of this lane's eight exist
    };
    auto load_stage = [&](int64_t pb, Stage& S) __attribute__((always_inline)) {
        const int64_t p0 = pb + 8 * h;
        const int64_t left = r_end - p0;
        S.nlive = left < 0 ? 0 : (left > 8 ? 8 : (int)left);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t pos = p0 + r;
            const int64_t want = !lab_tile ? pos : (pos < r_end ? r0 + lab_list[pos - r0] : r1);
            const int64_t nn = want < r1 ? want : r1 - 1;  // clamped: loads never wait on a predicate, zeroed at use
            S.g[r] = *reinterpret_cast<const float4*>(gsrc + nn * sy.ldd);
            if (ACT) S.t[r] = *reinterpret_cast<const float4*>(sy.T + nn * sy.ldt + o0);
            if (!EFF) S.mk[r] = sy.mask[nn];
            S.x[r] = *reinterpret_cast<const float2*>(xsrc + nn * xld);
        }
    };
    auto cut8 = [](const float (&v)[8], uint4 (&f)[3]) __attribute__((always_inline)) {
        split2(v[0], v[1], f[0].x, f[1].x, f[2].x);
        split2(v[2], v[3], f[0].y, f[1].y, f[2].y);
        split2(v[4], v[5], f[0].z, f[1].z, f[2].z);
        split2(v[6], v[7], f[0].w, f[1].w, f[2].w);
    };
    auto run_stage = [&](Stage& S, int64_t refill_pb) __attribute__((always_inline)) {
        float gv[4][8], xv[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float4 g = S.g[r];
            if (!EFF) {
                const float cf = ((S.mk[r] != 0) == first) ? sy.zr : sy.omz;
                g.x *= cf; g.y *= cf; g.z *= cf; g.w *= cf;
                if (ACT) {
                    const float4 t = S.t[r];
                    g.x *= t.x > 0.f ? 1.f : e_neg * __expf(t.x);
                    g.y *= t.y > 0.f ? 1.f : e_neg * __expf(t.y);
                    g.z *= t.z > 0.f ? 1.f : e_neg * __expf(t.z);
                    g.w *= t.w > 0.f ? 1.f : e_neg * __expf(t.w);
                }
            }
            if (r >= S.nlive) g = make_float4(0.f, 0.f, 0.f, 0.f);
            bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
            gv[0][r] = g.x; gv[1][r] = g.y; gv[2][r] = g.z; gv[3][r] = g.w;
            xv[0][r] = S.x[r].x; xv[1][r] = S.x[r].y;
        }
        uint4 A[4][3], B[2][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) cut8(gv[t], A[t]);
#pragma unroll
        for (int u = 0; u < 2; ++u) cut8(xv[u], B[u]);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(refill_pb, S);  // this stage's registers are free: two groups ahead
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) split_mma(acc[t][u], A[t], B[u]);
    };
    Stage st[2];
    const int64_t pb0 = r0 + 16 * w;  // wave-uniform: 16-row groups w, w + 4, ...
    load_stage(pb0, st[0]);
    load_stage(pb0 + 64, st[1]);
    for (int64_t pb = pb0; pb < r_end; pb += 128) {
        run_stage(st[0], pb + 128);
        run_stage(st[1], pb + 192);
    }
    wgrad_partial_tail(acc, bsum, lab_tile, bx, by, bz, gx, gy, part_w, part_b, lds, lds_b);
}

// ---- trans pair at hidden 64, staged form (inside the fused backward launch of small graphs) --------------------------
// The register-pipelined body above walks a slab two row pairs at a time with 2 stages of prefetch: ~0.5 us of MFMAs per
// stage against a ~2 us loaded round trip, so a 72-row slab lives 16 us for 2 us of matrix work.  Here the slab goes
// through LDS in 16-row stages like the staged forward kernels (dense.hip): every thread loads one float4 of the output
// gradient, of each pre-activation half and of the pair's input per stage (coalesced, buffer-addressed: no conditional
// memory instruction), synthesises its eight elements of dZ = mix'(dout) . ELU'(T) and writes dZ and the input TRANSPOSED
// ([column][16 rows]); the loads of stage s + 2 stay in flight across the LDS-only barrier of stage s.  A wave owns 32 of
// the 128 outputs x all 64 inputs (8 accumulator tiles of 16 x 16) and reads both operands four rows per b128.  Partial
// tile in plain [o][i] order (header[2] = 1); bias partial = row sums of the dZ tile.
constexpr int kStg2RT = 20;                              // row stride of the transposed tiles (floats)
constexpr int kStg2Floats = (128 + 64) * kStg2RT;        // dZ^T [128][RT] + X^T [64][RT] per buffer

__device__ __forceinline__ void wgrad_trans_staged2_body(const float* __restrict__ X, int64_t ldx, int64_t N, int rows_per_slab,
                                                         float* __restrict__ part_w, float* __restrict__ part_b,
                                                         const WgradSynth& sy, int bx, float* lds) {
    constexpr int H = 64, RT = kStg2RT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid >> 4, ga = tid & 15;
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < N ? r0 + rows_per_slab : N;
    const int n_st = (int)((r1 - r0 + 15) / 16);
    const buf_rsrc r_d = make_rsrc(sy.dsrc, N * sy.ldd * 4), r_t = make_rsrc(sy.T, N * sy.ldt * 4), r_x = make_rsrc(X, N * ldx * 4);
    const buf_rsrc r_m = make_rsrc(sy.mask, N);
    struct Raw {
        float4 d, t1, t0, x;
        unsigned mk;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int64_t r = r0 + 16 * st + rs;
        const bool ok = r < r1;
        const int ri = (int)r;
        R.d = buf_load4(r_d, ok ? (int)((ri * sy.ldd + 4 * ga) * 4) : kBufOOB);
        R.t1 = buf_load4(r_t, ok ? (int)((ri * sy.ldt + 4 * ga) * 4) : kBufOOB);
        R.t0 = buf_load4(r_t, ok ? (int)((ri * sy.ldt + H + 4 * ga) * 4) : kBufOOB);
        R.x = buf_load4(r_x, ok ? (int)((ri * ldx + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_m, ok ? ri : kBufOOB, 0, 0);
    };
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        float* dzT = lds + (st & 1) * kStg2Floats;
        float* xT = dzT + 128 * RT;
        const float c1 = R.mk ? sy.zr : sy.omz, c0 = R.mk ? sy.omz : sy.zr;
        const float d[4] = {R.d.x, R.d.y, R.d.z, R.d.w}, xv[4] = {R.x.x, R.x.y, R.x.z, R.x.w};
        const float t1[4] = {R.t1.x, R.t1.y, R.t1.z, R.t1.w}, t0[4] = {R.t0.x, R.t0.y, R.t0.z, R.t0.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float z1 = d[k] * c1, z0 = d[k] * c0;
            z1 *= act_grad(sy.act, t1[k]), z0 *= act_grad(sy.act, t0[k]);
            dzT[(4 * ga + k) * RT + rs] = z1;
            dzT[(H + 4 * ga + k) * RT + rs] = z0;
            xT[(4 * ga + k) * RT + rs] = xv[k];
        }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    wg_f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[a][it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;  // bias partial of output o = tid (threads < 128)
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    for (int st = 0; st < n_st; ++st) {
        const float* dzT = lds + (st & 1) * kStg2Floats;
        const float* xT = dzT + 128 * RT;
        float4 at[2], bt[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) at[a] = *reinterpret_cast<const float4*>(dzT + (16 * (2 * w + a) + j) * RT + 4 * q);
#pragma unroll
        for (int it = 0; it < 4; ++it) bt[it] = *reinterpret_cast<const float4*>(xT + (16 * it + j) * RT + 4 * q);
        if (tid < 128) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = *reinterpret_cast<const float4*>(dzT + tid * RT + 4 * v);
                bsum += (t4.x + t4.y) + (t4.z + t4.w);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float av[4] = {at[a].x, at[a].y, at[a].z, at[a].w};
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (!GLASS_MFMA_KEEP((e * 2 + a) * 4 + it)) continue;
                    const float bv[4] = {bt[it].x, bt[it].y, bt[it].z, bt[it].w};
                    acc[a][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[a][it], 0, 0, 0);
                }
            }
        if (st + 1 < n_st) {
            if (st & 1) {
                commit(st + 1, rawA);
                issue(st + 3, rawA);
            } else {
                commit(st + 1, rawB);
                issue(st + 3, rawB);
            }
            lds_barrier();
        }
    }
    // partial tile, plain [o][i]: o = 16 (2w + a) + 4q + r, i = 16 it + j
    float* pw = part_w + (int64_t)bx * kTile;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[(16 * (2 * w + a) + 4 * q + r) * H + 16 * it + j] = acc[a][it][r];
    if (part_b && tid < 128) part_b[(int64_t)bx * kOT + tid] = bsum;
}

// ---- comb pair at hidden 64: weight gradient in effective-weight ("S / L") form -------------------------------------
// dW1 = sum_r w1(r) dc[r]^T c[r], dW0 = sum_r w0(r) dc[r]^T c[r] with c = [g || x_] and (w1, w0) = (z, 1-z) on labeled rows,
// (1-z, z) elsewhere, so with  S = sum over ALL rows of dc^T c  and  L = the same sum over the LABELED rows:
//     dW1 = (1-z) S + (2z-1) L,   dW0 = z S - (2z-1) L      (bias gradients likewise from the column sums of dc).
// S is ONE [H x 2H] product over the rows instead of the [2H x 2H] of the plain form (half the matrix work); L walks the
// batch's unique labeled rows (glass_batch_labels) — a few workgroups.  Roles of the two MFMA operands swapped against
// wgrad_partial_body: a lane reads float2 of dc (outputs o = 2m + t) and float4 of c (inputs i = 4n + u), so ONE workgroup
// tile covers all 64 x 128 outputs: acc[t < 2][u < 4], the same 8 accumulator tiles.  Blocks [0, n_s) take row slabs,
// blocks [n_s, n_s + n_l) take 64 list positions each; partial tiles [n_s + n_l][kTile], bias partials [n_s + n_l][64],
// then the mode header {2, z_ratio, n_s, n_l}; combined by the reduce kernels (linear.hip).
struct WgradSL {
    const float* dc; int64_t ldd;
    const float* X; int64_t ldx;     // g   [N,H]
    const float* X2; int64_t ldx2;   // x_  [N,H]
    const int32_t* lab_rows;
    const int32_t* lab_count;
    int n_s, rows_per_slab, n_l;
    GnBwdSrc src;  // src.acc != nullptr: dc is derived on load (staged body only)
    const uint64_t* rng_state;
};
struct WgradSLGeom {
    int n_s, rows_per_slab, n_l;
    int64_t part_w_floats, part_b_floats;  // part_b_floats includes the header
};
WgradSLGeom wgrad_sl_geom(int64_t N, int64_t lab_cap);  // linear.hip
void launch_wgrad_sl(const WgradSL& a, int64_t N, float zr, float* part_w, float* part_b, hipStream_t st);  // linear.hip
constexpr int kSLOut = 64;  // outputs of the S / L tile (H)
constexpr int64_t kFusedBwdMaxRows = 100000;  // up to here the data + weight gradient of a pair share one launch
#ifndef GLASS_COMB_BWD_V2
#define GLASS_COMB_BWD_V2 0  // (dense.hip: measured slower, kept as laboratory code)
#endif

__device__ __forceinline__ int acc_index_sl(int t, int u, int reg, int lane) { return ((t * 4 + u) * 16 + reg) * 64 + lane; }

// S / L tiles in the staged form (see wgrad_trans_staged2_body): the slab's rows — or, for an L tile, the listed labeled
// rows (their indices fetched up front: at most 64 per tile) — go through LDS in 16-row stages as TRANSPOSED tiles of dc
// ([64][16]) and of [g || x_] ([128][16]); a wave owns 16 outputs x all 128 inputs.  Plain [o][i] tile (header[2] = 1).
__device__ __forceinline__ void wgrad_sl_staged2_body(const WgradSL& a, int64_t N, int blk, float* __restrict__ part_w,
                                                      float* __restrict__ part_b, float* lds) {
    constexpr int H = 64, RT = kStg2RT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid >> 4, ga = tid & 15;
    const bool lab = blk >= a.n_s;
    int64_t r0, r_end;
    if (!lab) {
        r0 = (int64_t)blk * a.rows_per_slab;
        r_end = min(N, r0 + a.rows_per_slab);
    } else {
        const int64_t n_lab = a.lab_count[0];
        r0 = (int64_t)(blk - a.n_s) * 64;
        r_end = min(n_lab, r0 + 64);
    }
    const int n_st = r_end > r0 ? (int)((r_end - r0 + 15) / 16) : 0;
    int li[4] = {-1, -1, -1, -1};  // L tile: the rows of this thread's list positions of stages 0 .. 3
    if (lab) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int64_t pos = r0 + 16 * st + rs;
            li[st] = pos < r_end ? a.lab_rows[pos] : -1;
        }
    }
    const GnBwdSrc& src = a.src;
    const bool src_on = src.acc != nullptr;
    Drop sdrop = src.drop;
    if (src_on && sdrop.p > 0.f) {
        sdrop.seed = a.rng_state[0];
        sdrop.step = a.rng_state[1];
    }
    float* coef_s = lds + 2 * kStg2Floats;  // [5][64] (src)
    const buf_rsrc r_d = src_on ? make_rsrc(src.dy, N * src.lddy * 4) : make_rsrc(a.dc, N * a.ldd * 4);
    const int64_t ld_d = src_on ? src.lddy : a.ldd;
    const buf_rsrc r_sx = make_rsrc(src_on ? src.x : a.X, src_on ? N * src.ldx * 4 : 0);
    const buf_rsrc r_sad = make_rsrc((src_on && src.addend) ? src.addend : a.X, (src_on && src.addend) ? N * src.ldadd * 4 : 0);
    const buf_rsrc r_g = make_rsrc(a.X, N * a.ldx * 4), r_x = make_rsrc(a.X2, N * a.ldx2 * 4);
    struct Raw {
        float4 d, g, x, sx, sad;
        int row;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int64_t pos = r0 + 16 * st + rs;
        int row = pos < r_end ? (int)pos : -1;
        if (lab) row = st == 0 ? li[0] : st == 1 ? li[1] : st == 2 ? li[2] : st == 3 ? li[3] : -1;
        R.row = row;
        R.d = buf_load4(r_d, row >= 0 ? (int)((row * ld_d + 4 * ga) * 4) : kBufOOB);
        R.sx = buf_load4(r_sx, row >= 0 ? (int)((row * src.ldx + 4 * ga) * 4) : kBufOOB);
        R.sad = buf_load4(r_sad, row >= 0 ? (int)((row * src.ldadd + 4 * ga) * 4) : kBufOOB);
        R.g = buf_load4(r_g, row >= 0 ? (int)((row * a.ldx + 4 * ga) * 4) : kBufOOB);
        R.x = buf_load4(r_x, row >= 0 ? (int)((row * a.ldx2 + 4 * ga) * 4) : kBufOOB);
    };
    auto commit = [&](int buf, const Raw& R) __attribute__((always_inline)) {
        float* dcT = lds + buf * kStg2Floats;
        float* inT = dcT + H * RT;
        float4 dcv = R.d;
        if (src_on) {
            float sds[4] = {1.f, 1.f, 1.f, 1.f};
            if (sdrop.p > 0.f) drop_scales<4>(sdrop, R.row < 0 ? 0 : R.row, 4 * ga, sds);
            dcv = gn_bwd_apply4(R.d, R.sx, R.sad, coef_s, 4 * ga, src.act, sds);
            if (R.row < 0) dcv = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float d[4] = {dcv.x, dcv.y, dcv.z, dcv.w}, g[4] = {R.g.x, R.g.y, R.g.z, R.g.w}, x[4] = {R.x.x, R.x.y, R.x.z, R.x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dcT[(4 * ga + k) * RT + rs] = d[k];
            inT[(4 * ga + k) * RT + rs] = g[k];
            inT[(H + 4 * ga + k) * RT + rs] = x[k];
        }
    };
    wg_f32x4 acc[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) acc[it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;  // bias partial of output o = tid (threads < 64)
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* dcT = lds + buf * kStg2Floats;
        const float* inT = dcT + H * RT;
        const float4 at = *reinterpret_cast<const float4*>(dcT + (16 * w + j) * RT + 4 * q);
        float4 bt[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) bt[it] = *reinterpret_cast<const float4*>(inT + (16 * it + j) * RT + 4 * q);
        if (tid < H) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = *reinterpret_cast<const float4*>(dcT + tid * RT + 4 * v);
                bsum += (t4.x + t4.y) + (t4.z + t4.w);
            }
        }
        const float av[4] = {at.x, at.y, at.z, at.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (!GLASS_MFMA_KEEP(e * 8 + it)) continue;
                const float bv[4] = {bt[it].x, bt[it].y, bt[it].z, bt[it].w};
                acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[it], 0, 0, 0);
            }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    if (src_on) {
        gn_bwd_coef_nobarrier(src.acc, src.n_rep, N, src.saved, src.gamma, src.alpha, nullptr, nullptr, nullptr, 0, false, coef_s);
        lds_barrier();  // coefficients before the first stage is prepared
    }
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    for (int st = 0; st < n_st; st += 2) {  // two stages per trip: buffers / registers alternate statically
        compute(0);
        commit(1, rawB);      // stage st + 1 (zeros beyond the end)
        issue(st + 3, rawB);
        lds_barrier();
        compute(1);
        commit(0, rawA);      // stage st + 2
        issue(st + 4, rawA);
        lds_barrier();
    }
    float* pw = part_w + (int64_t)blk * kTile;
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(16 * w + 4 * q + r) * (2 * H) + 16 * it + j] = acc[it][r];
    if (tid < H) part_b[(int64_t)blk * kSLOut + tid] = bsum;
}

// ---- the two staged bodies above in the split product form (hidden 64, round 6) ----------------------------------------
// v_mfma_f32_16x16x32_bf16 sums over 32 node rows per instruction, so a stage is 32 rows: every thread loads TWO rows (2 rp,
// 2 rp + 1) x one column quad per operand, synthesises its elements, cuts each column's row pair into three bf16 pieces
// (split2: the pair packed in one word — exactly the unit the transposed image stores) and writes three words per column.
// Image per operand: [piece][column][32 rows] bf16 = 64 bytes per column, NO padding (two buffers of dZ^T [128] + X^T [64]
// = 72 KiB: two workgroups per CU stay resident); the 16-byte chunk q (rows 8 q .. 8 q + 7) of column c lies at chunk
// q ^ ((c >> 2) & 3), so the 16 lanes of a fragment read (16 consecutive columns, one q) touch 16 different bank groups
// 48 MFMAs of 16 cycles per 32 rows where the f32-input form runs 64 of 32.  Bias partials: the loaders sum what they
// synthesise (lane shuffles over the row pairs, one pass through LDS over the waves).  Partial tiles as the f32 bodies write them.
constexpr int kStg2sWordsA = 3 * 128 * 16, kStg2sWordsB = 3 * 64 * 16;   // 32-bit words of the two images of one buffer (trans pair)
constexpr int kStg2sWords = kStg2sWordsA + kStg2sWordsB;                 // 9 216 words = 36 KiB per buffer (S / L: the same sum)
constexpr size_t kStg2sLdsBytes = (size_t)(2 * kStg2sWords + 5 * 64) * 4;  // + the [5][64] coefficients of a derived dc
// ... and the four columns of a quad are rotated by the quad's group ((c >> 4) & 3), so that the 64 lanes of a STORE (16 column
// quads x 4 row pairs, one k) land in 64 different banks as well (unrotated: 16 banks, every store 4-way conflicted — with
// eight waves per CU storing 36 words per thread and stage that, not the MFMAs, would pace the stage).
__device__ __forceinline__ int stg2s_slot(int c) { return (c & ~3) | ((c + (c >> 4)) & 3); }
__device__ __forceinline__ int stg2s_word(int c, int rp) { return stg2s_slot(c) * 16 + ((((rp >> 2) ^ (c >> 2)) & 3) << 2) + (rp & 3); }
__device__ __forceinline__ int stg2s_chunk(int c, int q) { return stg2s_slot(c) * 4 + ((q ^ (c >> 2)) & 3); }

#define GLASS_SPLIT6(ACC, AF, BF)                                                                                     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[1]), __builtin_bit_cast(bf16x8, BF[1]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[2]), __builtin_bit_cast(bf16x8, BF[0]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[0]), __builtin_bit_cast(bf16x8, BF[2]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[1]), __builtin_bit_cast(bf16x8, BF[0]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[0]), __builtin_bit_cast(bf16x8, BF[1]), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AF[0]), __builtin_bit_cast(bf16x8, BF[0]), ACC, 0, 0, 0)

__device__ __forceinline__ void wgrad_trans_staged2s_body(const float* __restrict__ X, int64_t ldx, int64_t N, int rows_per_slab,
                                                          float* __restrict__ part_w, float* __restrict__ part_b,
                                                          const WgradSynth& sy, int bx, float* lds) {
    constexpr int H = 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rp = tid >> 4, ga = tid & 15;  // loader role: row pair rp of the stage, columns 4 ga .. 4 ga + 3
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < N ? r0 + rows_per_slab : N;
    const int n_st = (int)((r1 - r0 + 31) / 32);
    const buf_rsrc r_d = make_rsrc(sy.dsrc, N * sy.ldd * 4), r_t = make_rsrc(sy.T, N * sy.ldt * 4), r_x = make_rsrc(X, N * ldx * 4);
    const buf_rsrc r_m = make_rsrc(sy.mask, N);
    struct Raw {
        float4 d[2], t1[2], t0[2], x[2];
        unsigned mk[2];
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = r0 + 32 * st + 2 * rp + u;
            const bool ok = r < r1;
            const int ri = (int)r;
            R.d[u] = buf_load4(r_d, ok ? (int)((ri * sy.ldd + 4 * ga) * 4) : kBufOOB);
            R.t1[u] = buf_load4(r_t, ok ? (int)((ri * sy.ldt + 4 * ga) * 4) : kBufOOB);
            R.t0[u] = buf_load4(r_t, ok ? (int)((ri * sy.ldt + H + 4 * ga) * 4) : kBufOOB);
            R.x[u] = buf_load4(r_x, ok ? (int)((ri * ldx + 4 * ga) * 4) : kBufOOB);
            R.mk[u] = __builtin_amdgcn_raw_buffer_load_b8(r_m, ok ? ri : kBufOOB, 0, 0);
        }
    };
    float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs0[4] = {0.f, 0.f, 0.f, 0.f};  // bias partials of outputs 4 ga + k / H + 4 ga + k over this thread's rows
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        unsigned* dz = reinterpret_cast<unsigned*>(lds) + (st & 1) * kStg2sWords;
        unsigned* xi = dz + kStg2sWordsA;
        float z1[2][4], z0[2][4], xv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float c1 = R.mk[u] ? sy.zr : sy.omz, c0 = R.mk[u] ? sy.omz : sy.zr;
            const float d[4] = {R.d[u].x, R.d[u].y, R.d[u].z, R.d[u].w};
            const float t1[4] = {R.t1[u].x, R.t1[u].y, R.t1[u].z, R.t1[u].w}, t0[4] = {R.t0[u].x, R.t0[u].y, R.t0[u].z, R.t0[u].w};
            xv[u][0] = R.x[u].x, xv[u][1] = R.x[u].y, xv[u][2] = R.x[u].z, xv[u][3] = R.x[u].w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                z1[u][k] = d[k] * c1 * act_grad(sy.act, t1[k]);
                z0[u][k] = d[k] * c0 * act_grad(sy.act, t0[k]);
                bs1[k] += z1[u][k];
                bs0[k] += z0[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned hi, mid, lo;
            const int c = 4 * ga + k;
            split2(z1[0][k], z1[1][k], hi, mid, lo);
            int wd = stg2s_word(c, rp);
            dz[wd] = hi, dz[128 * 16 + wd] = mid, dz[2 * 128 * 16 + wd] = lo;
            split2(z0[0][k], z0[1][k], hi, mid, lo);
            wd = stg2s_word(H + c, rp);
            dz[wd] = hi, dz[128 * 16 + wd] = mid, dz[2 * 128 * 16 + wd] = lo;
            split2(xv[0][k], xv[1][k], hi, mid, lo);
            wd = stg2s_word(c, rp);
            xi[wd] = hi, xi[64 * 16 + wd] = mid, xi[2 * 64 * 16 + wd] = lo;
        }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    wg_f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[a][it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    D_STAMP(4, 1);
    for (int st = 0; st < n_st; ++st) {
        if (st == 1) D_STAMP(4, 2);
        const uint4* dz = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(lds) + (st & 1) * kStg2sWords);
        const uint4* xi = dz + kStg2sWordsA / 4;
        uint4 af[2][3], bf[4][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) af[a][pc] = dz[pc * 128 * 4 + stg2s_chunk(16 * (2 * w + a) + j, q)];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bf[it][pc] = xi[pc * 64 * 4 + stg2s_chunk(16 * it + j, q)];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int it = 0; it < 4; ++it) { GLASS_SPLIT6(acc[a][it], af[a], bf[it]); }
        if (st + 1 < n_st) {
            if (st & 1) {
                commit(st + 1, rawA);
                issue(st + 3, rawA);
            } else {
                commit(st + 1, rawB);
                issue(st + 3, rawB);
            }
            lds_barrier();
        }
    }
    D_STAMP(4, 3);
    // partial tile, plain [o][i]: o = 16 (2w + a) + 4q + r, i = 16 it + j
    float* pw = part_w + (int64_t)bx * kTile;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[(16 * (2 * w + a) + 4 * q + r) * H + 16 * it + j] = acc[a][it][r];
    D_STAMP(4, 7);
    if (part_b) {  // bias partial: the wave's four row-pair slots of a column by shuffles, the four waves through LDS
        lds_barrier();  // (every wave is done reading the last stage's image)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bs1[k] += __shfl_xor(bs1[k], 16);
            bs0[k] += __shfl_xor(bs0[k], 16);
            bs1[k] += __shfl_xor(bs1[k], 32);
            bs0[k] += __shfl_xor(bs0[k], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds[w * 128 + 4 * ga + k] = bs1[k];
                lds[w * 128 + H + 4 * ga + k] = bs0[k];
            }
        }
        lds_barrier();
        if (tid < 128) part_b[(int64_t)bx * kOT + tid] = (lds[tid] + lds[128 + tid]) + (lds[256 + tid] + lds[384 + tid]);
    }
}

__device__ __forceinline__ void wgrad_sl_staged2s_body(const WgradSL& a, int64_t N, int blk, float* __restrict__ part_w,
                                                       float* __restrict__ part_b, float* lds) {
    constexpr int H = 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rp = tid >> 4, ga = tid & 15;
    const bool lab = blk >= a.n_s;
    int64_t r0, r_end;
    if (!lab) {
        r0 = (int64_t)blk * a.rows_per_slab;
        r_end = min(N, r0 + a.rows_per_slab);
    } else {
        const int64_t n_lab = a.lab_count[0];
        r0 = (int64_t)(blk - a.n_s) * 64;
        r_end = min(n_lab, r0 + 64);
    }
    const int n_st = r_end > r0 ? (int)((r_end - r0 + 31) / 32) : 0;
    int li[2][2] = {{-1, -1}, {-1, -1}};  // L tile: the rows of this thread's list positions of stages 0, 1
    if (lab) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t pos = r0 + 32 * st + 2 * rp + u;
                li[st][u] = pos < r_end ? a.lab_rows[pos] : -1;
            }
    }
    const GnBwdSrc& src = a.src;
    const bool src_on = src.acc != nullptr;
    Drop sdrop = src.drop;
    if (src_on && sdrop.p > 0.f) {
        sdrop.seed = a.rng_state[0];
        sdrop.step = a.rng_state[1];
    }
    float* coef_s = lds + 2 * kStg2sWords;  // [5][64] (src)
    const buf_rsrc r_d = src_on ? make_rsrc(src.dy, N * src.lddy * 4) : make_rsrc(a.dc, N * a.ldd * 4);
    const int64_t ld_d = src_on ? src.lddy : a.ldd;
    const buf_rsrc r_sx = make_rsrc(src_on ? src.x : a.X, src_on ? N * src.ldx * 4 : 0);
    const buf_rsrc r_sad = make_rsrc((src_on && src.addend) ? src.addend : a.X, (src_on && src.addend) ? N * src.ldadd * 4 : 0);
    const buf_rsrc r_g = make_rsrc(a.X, N * a.ldx * 4), r_x = make_rsrc(a.X2, N * a.ldx2 * 4);
    struct Raw {
        float4 d[2], g[2], x[2], sx[2], sad[2];
        int row[2];
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t pos = r0 + 32 * st + 2 * rp + u;
            int row = pos < r_end ? (int)pos : -1;
            if (lab) row = st == 0 ? li[0][u] : st == 1 ? li[1][u] : -1;
            R.row[u] = row;
            R.d[u] = buf_load4(r_d, row >= 0 ? (int)((row * ld_d + 4 * ga) * 4) : kBufOOB);
            R.sx[u] = buf_load4(r_sx, row >= 0 ? (int)((row * src.ldx + 4 * ga) * 4) : kBufOOB);
            R.sad[u] = buf_load4(r_sad, row >= 0 ? (int)((row * src.ldadd + 4 * ga) * 4) : kBufOOB);
            R.g[u] = buf_load4(r_g, row >= 0 ? (int)((row * a.ldx + 4 * ga) * 4) : kBufOOB);
            R.x[u] = buf_load4(r_x, row >= 0 ? (int)((row * a.ldx2 + 4 * ga) * 4) : kBufOOB);
        }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};  // bias partial of outputs 4 ga + k over this thread's rows
    auto commit = [&](int buf, const Raw& R) __attribute__((always_inline)) {
        unsigned* dcI = reinterpret_cast<unsigned*>(lds) + buf * kStg2sWords;  // dc^T: 64 columns; [g || x_]^T: 128 columns
        unsigned* inI = dcI + kStg2sWordsB;
        float d[2][4], g[2][4], x[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float4 dcv = R.d[u];
            if (src_on) {
                float sds[4] = {1.f, 1.f, 1.f, 1.f};
                if (sdrop.p > 0.f) drop_scales<4>(sdrop, R.row[u] < 0 ? 0 : R.row[u], 4 * ga, sds);
                dcv = gn_bwd_apply4(R.d[u], R.sx[u], R.sad[u], coef_s, 4 * ga, src.act, sds);
                if (R.row[u] < 0) dcv = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            d[u][0] = dcv.x, d[u][1] = dcv.y, d[u][2] = dcv.z, d[u][3] = dcv.w;
            g[u][0] = R.g[u].x, g[u][1] = R.g[u].y, g[u][2] = R.g[u].z, g[u][3] = R.g[u].w;
            x[u][0] = R.x[u].x, x[u][1] = R.x[u].y, x[u][2] = R.x[u].z, x[u][3] = R.x[u].w;
#pragma unroll
            for (int k = 0; k < 4; ++k) bs[k] += d[u][k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned hi, mid, lo;
            const int c = 4 * ga + k;
            split2(d[0][k], d[1][k], hi, mid, lo);
            int wd = stg2s_word(c, rp);
            dcI[wd] = hi, dcI[64 * 16 + wd] = mid, dcI[2 * 64 * 16 + wd] = lo;
            split2(g[0][k], g[1][k], hi, mid, lo);
            inI[wd] = hi, inI[128 * 16 + wd] = mid, inI[2 * 128 * 16 + wd] = lo;
            split2(x[0][k], x[1][k], hi, mid, lo);
            wd = stg2s_word(H + c, rp);
            inI[wd] = hi, inI[128 * 16 + wd] = mid, inI[2 * 128 * 16 + wd] = lo;
        }
    };
    wg_f32x4 acc[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) acc[it] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const uint4* dcI = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned*>(lds) + buf * kStg2sWords);
        const uint4* inI = dcI + kStg2sWordsB / 4;
        uint4 af[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[pc] = dcI[pc * 64 * 4 + stg2s_chunk(16 * w + j, q)];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            uint4 bf[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bf[pc] = inI[pc * 128 * 4 + stg2s_chunk(16 * it + j, q)];
            GLASS_SPLIT6(acc[it], af, bf);
        }
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    if (src_on) {
        gn_bwd_coef_nobarrier(src.acc, src.n_rep, N, src.saved, src.gamma, src.alpha, nullptr, nullptr, nullptr, 0, false, coef_s);
        lds_barrier();  // coefficients before the first stage is prepared
    }
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    D_STAMP(3, 1);
    for (int st = 0; st < n_st; ++st) {
        if (st == 1) D_STAMP(3, 2);
        compute(st & 1);
        if (st + 1 < n_st) {
            if (st & 1) {
                commit(0, rawA);
                issue(st + 3, rawA);
            } else {
                commit(1, rawB);
                issue(st + 3, rawB);
            }
            lds_barrier();
        }
    }
    D_STAMP(3, 3);
    float* pw = part_w + (int64_t)blk * kTile;
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(16 * w + 4 * q + r) * (2 * H) + 16 * it + j] = acc[it][r];
    D_STAMP(3, 7);
    lds_barrier();  // (every wave is done reading the last stage's image)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bs[k] += __shfl_xor(bs[k], 16);
        bs[k] += __shfl_xor(bs[k], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[w * 64 + 4 * ga + k] = bs[k];
    }
    lds_barrier();
    if (tid < H) part_b[(int64_t)blk * kSLOut + tid] = (lds[tid] + lds[64 + tid]) + (lds[128 + tid] + lds[192 + tid]);
}

template <int kStages>
__device__ __forceinline__ void wgrad_sl_body(const WgradSL& a, int64_t N, int blk, float* __restrict__ part_w,
                                              float* __restrict__ part_b, float* lds, float* lds_b) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const bool lab = blk >= a.n_s;
    int64_t r0, r_end;
    if (!lab) {
        r0 = (int64_t)blk * a.rows_per_slab;
        r_end = min(N, r0 + a.rows_per_slab);
    } else {
        const int64_t n_lab = a.lab_count[0];
        r0 = (int64_t)(blk - a.n_s) * 64;
        r_end = min(n_lab, r0 + 64);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][u][k] = 0.f;
    float2 bsum = make_float2(0.f, 0.f);
    struct Stage {
        float2 g[2];
        float4 x[2];
        bool live[2];
    };
    Stage st[kStages];
    const float* xbase = c < 16 ? a.X + 4 * c : a.X2 + (4 * c - 64);
    const int64_t xld = c < 16 ? a.ldx : a.ldx2;
    auto load_stage = [&](int64_t nb, Stage& S) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t pos = nb + h + 8 * s;
            const bool in = pos < r_end;
            int64_t row = in ? pos : 0;
            if (lab) row = in ? a.lab_rows[pos] : 0;
            S.live[s] = in;
            S.g[s] = *reinterpret_cast<const float2*>(a.dc + row * a.ldd + 2 * c);
            S.x[s] = *reinterpret_cast<const float4*>(xbase + row * xld);
        }
    };
    const int64_t nb0 = r0 + 2 * w;  // wave-uniform (MFMA needs every lane in the loop)
#pragma unroll
    for (int k = 0; k < kStages; ++k) load_stage(nb0 + 16 * k, st[k]);
    for (int64_t nb = nb0; nb < r_end; nb += 16 * kStages) {
#pragma unroll
        for (int k = 0; k < kStages; ++k) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float2 g = st[k].g[s];
                if (!st[k].live[s]) g = make_float2(0.f, 0.f);
                const float gv[2] = {g.x, g.y};
                const float xv[4] = {st[k].x[s].x, st[k].x[s].y, st[k].x[s].z, st[k].x[s].w};
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[t], xv[u], acc[t][u], 0, 0, 0);
                bsum.x += g.x;
                bsum.y += g.y;
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the refill below from being sunk into later stages
            load_stage(nb + 16 * (k + kStages), st[k]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- combine the 4 waves through LDS: waves 0,1 store; waves 2,3 add; everyone sums the pair ----
    if (w < 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[w * kTile + acc_index_sl(t, u, k, lane)] = acc[t][u][k];
    }
    *reinterpret_cast<float2*>(&lds_b[(w * 2 + h) * kSLOut + 2 * c]) = bsum;
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[(w - 2) * kTile + acc_index_sl(t, u, k, lane)] += acc[t][u][k];
    }
    __syncthreads();
    float* pw = part_w + (int64_t)blk * kTile;
    for (int k = threadIdx.x * 4; k < kTile; k += kBlock * 4) {
        const float4 p = *reinterpret_cast<const float4*>(&lds[k]);
        const float4 q = *reinterpret_cast<const float4*>(&lds[kTile + k]);
        *reinterpret_cast<float4*>(pw + k) = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
    }
    if (threadIdx.x < kSLOut) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds_b[k * kSLOut + threadIdx.x];
        part_b[(int64_t)blk * kSLOut + threadIdx.x] = s;
    }
}

// ---- small graphs: the whole slab through LDS ------------------------------------------------------------------------
// In the fused backward launch of a small graph a slab is <= 80 rows and a wave's share of it 10 row pairs: the register
// pipeline above then IS its fill — every stage waits a full memory round trip for loads issued two stages earlier, and
// the weight-gradient branch, not the data gradient, sets the launch's length.  Here the workgroup requests the slab's
// operands ONCE, all loads in flight together (one round trip), parks them in LDS (the gradient operand already
// synthesised), and the MFMA loop reads LDS only.  Same pairs per wave in the same order, same combine: bitwise the results
// of the pipelined bodies.  The staging area is the combine area (2 * kTile floats), reused after a barrier.
constexpr int kStageRows = 80;  // 80 rows x 192 floats = 60 KiB of the 64 KiB image area

// SYNTH form (trans pair): dZ [rows][2H = 128] synthesised while staging, X [rows][64].  O = 128, I = 64, one tile.
__device__ __forceinline__ void wgrad_synth_staged_body(const float* __restrict__ X, int64_t ldx, int64_t N, int rows_per_slab,
                                                        float* __restrict__ part_w, float* __restrict__ part_b,
                                                        const WgradSynth& sy, int bx, int gx, float* lds, float* lds_b) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t r0 = (int64_t)bx * rows_per_slab;
    const int nrows = (int)(min(N, r0 + rows_per_slab) - r0);
    const int nrows8 = (nrows + 7) & ~7;
    float* dz_s = lds;                       // [kStageRows][128]
    float* x_s = lds + kStageRows * 128;     // [kStageRows][64]
    // every load of the slab is issued before the first LDS store (registers: the accumulators are not live yet)
    constexpr int kSynthItems = kStageRows * 32 / kBlock, kXItems = kStageRows * 16 / kBlock;  // 10, 5
    float4 dv[kSynthItems], tv[kSynthItems], xv4[kXItems];
    float cfv[kSynthItems];
#pragma unroll
    for (int k = 0; k < kSynthItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item >> 5, oq = item & 31;
        dv[k] = tv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        cfv[k] = 0.f;
        if (row < nrows) {
            const int64_t n = r0 + row;
            dv[k] = *reinterpret_cast<const float4*>(sy.dsrc + n * sy.ldd + 4 * (oq & 15));
            if (sy.act != GLASS_ACT_NONE) tv[k] = *reinterpret_cast<const float4*>(sy.T + n * sy.ldt + 4 * oq);
            cfv[k] = ((sy.mask[n] != 0) == (oq < 16)) ? sy.zr : sy.omz;
        }
    }
#pragma unroll
    for (int k = 0; k < kXItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item >> 4, q = item & 15;
        xv4[k] = row < nrows ? *reinterpret_cast<const float4*>(X + (r0 + row) * ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < kSynthItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item >> 5, oq = item & 31;
        if (row >= nrows8) continue;
        float4 g = make_float4(dv[k].x * cfv[k], dv[k].y * cfv[k], dv[k].z * cfv[k], dv[k].w * cfv[k]);
        if (sy.act != GLASS_ACT_NONE) {
            g.x *= act_grad(sy.act, tv[k].x); g.y *= act_grad(sy.act, tv[k].y); g.z *= act_grad(sy.act, tv[k].z); g.w *= act_grad(sy.act, tv[k].w);
        }
        *reinterpret_cast<float4*>(dz_s + row * 128 + 4 * oq) = g;
    }
#pragma unroll
    for (int k = 0; k < kXItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item >> 4, q = item & 15;
        if (row < nrows8) *reinterpret_cast<float4*>(x_s + row * 64 + 4 * q) = xv4[k];
    }
    __syncthreads();
    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][u][k] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = w; 2 * p < nrows8; p += 4) {  // wave w: row pairs w, w + 4, ... (as the pipelined body)
        const int row = 2 * p + h;
        const float4 g = *reinterpret_cast<const float4*>(dz_s + row * 128 + 4 * c);
        const float2 x = *reinterpret_cast<const float2*>(x_s + row * 64 + 2 * c);
        const float gv[4] = {g.x, g.y, g.z, g.w};
        const float xv[2] = {x.x, x.y};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[t], xv[u], acc[t][u], 0, 0, 0);
        bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
    }
    __syncthreads();  // every wave is done reading the staged slab: the area becomes the combine area
    if (w < 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[w * kTile + acc_index(t, u, k, lane)] = acc[t][u][k];
    }
    *reinterpret_cast<float4*>(&lds_b[(w * 2 + h) * kOT + 4 * c]) = bsum;
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[(w - 2) * kTile + acc_index(t, u, k, lane)] += acc[t][u][k];
    }
    __syncthreads();
    float* pw = part_w + (int64_t)bx * kTile;  // (tile id = slab: one input tile, one output tile)
    for (int k = threadIdx.x * 4; k < kTile; k += kBlock * 4) {
        const float4 a = *reinterpret_cast<const float4*>(&lds[k]);
        const float4 b = *reinterpret_cast<const float4*>(&lds[kTile + k]);
        *reinterpret_cast<float4*>(pw + k) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    if (part_b && threadIdx.x < kOT) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds_b[k * kOT + threadIdx.x];
        part_b[(int64_t)bx * kOT + threadIdx.x] = s;
    }
    (void)gx;
}

// S / L form (comb pair at hidden 64): dc [rows][64], c = [g || x_] [rows][128]; labeled-row blocks gather their rows.
__device__ __forceinline__ void wgrad_sl_staged_body(const WgradSL& a, int64_t N, int blk, float* __restrict__ part_w,
                                                     float* __restrict__ part_b, float* lds, float* lds_b) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const bool lab = blk >= a.n_s;
    int64_t r0;
    int nrows;
    if (!lab) {
        r0 = (int64_t)blk * a.rows_per_slab;
        nrows = (int)(min(N, r0 + a.rows_per_slab) - r0);
    } else {
        const int64_t n_lab = a.lab_count[0];
        r0 = (int64_t)(blk - a.n_s) * 64;
        nrows = (int)max((int64_t)0, min(n_lab, r0 + 64) - r0);
    }
    const int nrows8 = (nrows + 7) & ~7;
    float* dc_s = lds;                      // [kStageRows][64]
    float* x_s = lds + kStageRows * 64;     // [kStageRows][128]
    // every load of the slab is issued before the first LDS store (registers: the accumulators are not live yet);
    // item = (row, quad): 16 quads of dc, 16 of g, 16 of x_ per row
    constexpr int kItems = kStageRows * 48 / kBlock;  // 15
    float4 v[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item / 48, q = item % 48;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows) {
            const int64_t n = lab ? (int64_t)a.lab_rows[r0 + row] : r0 + row;
            v[k] = q < 16 ? *reinterpret_cast<const float4*>(a.dc + n * a.ldd + 4 * q)
                   : q < 32 ? *reinterpret_cast<const float4*>(a.X + n * a.ldx + 4 * (q - 16))
                            : *reinterpret_cast<const float4*>(a.X2 + n * a.ldx2 + 4 * (q - 32));
        }
    }
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int item = threadIdx.x + kBlock * k, row = item / 48, q = item % 48;
        if (row >= nrows8) continue;
        if (q < 16) *reinterpret_cast<float4*>(dc_s + row * 64 + 4 * q) = v[k];
        else *reinterpret_cast<float4*>(x_s + row * 128 + 4 * (q - 16)) = v[k];
    }
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][u][k] = 0.f;
    float2 bsum = make_float2(0.f, 0.f);
    for (int p = w; 2 * p < nrows8; p += 4) {
        const int row = 2 * p + h;
        const float2 g = *reinterpret_cast<const float2*>(dc_s + row * 64 + 2 * c);
        const float4 x = *reinterpret_cast<const float4*>(x_s + row * 128 + 4 * c);
        const float gv[2] = {g.x, g.y};
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[t], xv[u], acc[t][u], 0, 0, 0);
        bsum.x += g.x;
        bsum.y += g.y;
    }
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[w * kTile + acc_index_sl(t, u, k, lane)] = acc[t][u][k];
    }
    *reinterpret_cast<float2*>(&lds_b[(w * 2 + h) * kSLOut + 2 * c]) = bsum;
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) lds[(w - 2) * kTile + acc_index_sl(t, u, k, lane)] += acc[t][u][k];
    }
    __syncthreads();
    float* pw = part_w + (int64_t)blk * kTile;
    for (int k = threadIdx.x * 4; k < kTile; k += kBlock * 4) {
        const float4 p = *reinterpret_cast<const float4*>(&lds[k]);
        const float4 q = *reinterpret_cast<const float4*>(&lds[kTile + k]);
        *reinterpret_cast<float4*>(pw + k) = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
    }
    if (threadIdx.x < kSLOut) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds_b[k * kSLOut + threadIdx.x];
        part_b[(int64_t)blk * kSLOut + threadIdx.x] = s;
    }
}

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
