// K5: the dense half of GLASSConv fused with its label-conditioned mix, on the fp32 matrix cores.
//
//   forward  (reference impl/models.py:158-162 and 167-173)
//     trans: T = x_ @ [W1;W0]^T + [b1|b0];  m = mix(ELU(T1), ELU(T0))        writes T (kept for backward) and m
//     comb : y = mix(C1, C0),  C = [g || x_] @ [Wc1;Wc0]^T + [bc1|bc0]        reads g and x_ in place (no cat),
//                                                                              never materialises C
//   backward data gradient:  dIN = dZ @ Wstack, with dZ = mix'(dsrc) (* ELU'(T)) synthesised on the fly
//   (the [N,2H] gradient of the stacked Linear output is never written).
//
// Why here and not in the vendor GEMM: at hidden=64 these GEMMs are tiny in two dimensions
// ([N,64..128] x [64..128,128]); hipBLASLt takes 12-18 us each and the mix/ELU/cat around them are 5-7
// separate elementwise launches of ~5 us (profiles/r01_bench_*): launch-bound, not FLOP-bound.
//
// Mapping (v_mfma_f32_16x16x4_f32; exact fp32 fma chain): a wave owns 16 rows and ALL output columns.
// Lane (i = l&15, q = l>>4) holds the contiguous K-chunk [q*KT/4, (q+1)*KT/4) of row i of A and, per
// 16-column output tile, the same chunk of row (16t+i) of the [out][in] weight — both are plain 16-B
// loads of consecutive floats; MFMA step s multiplies element s of the four chunks (the matrix core
// sums over k in any order, so no transposes or shuffles are needed).  The weight slice of a K-pass is
// staged once per workgroup in LDS, pre-permuted into consumption order (see WStage).  Hidden size 64; 128 / 256 / 512
// take the LDS-tiled kernels of dense_tiled.hip.
#include "common.h"
#include "dense_common.h"
#include "labels_body.h"
#include "split_mma.h"
#include "wgrad_common.h"
#include "emb_table.h"

#ifndef GLASS_WGRAD_STAGED
#define GLASS_WGRAD_STAGED 0  // laboratory switch: slabs through LDS whole (wgrad_*_staged_body) — measured slower, see DESIGN.md
#endif

#include <stdlib.h>

namespace glass {

#ifndef GLASS_COMB_BWD_V2
// comb backward at hidden 64 as ONE staged pass per 64-row tile (comb_bwd_eff2_kernel: data gradient + the tile's own
// weight-gradient partial).  Correct (the parity suite passes on it) and OFF: 18.1 us against 14.9 us for the two-kinds-of-
// workgroup launch at ppi_bp-shape — 64 MFMAs + 70 KB of LDS reads per stage make a stage 1.9 us, and 280 such workgroups
// on 256 CUs leave 24 CUs with two of them (their waves share the matrix cores: lives of 17 us beside a mean of 11.9).
#define GLASS_COMB_BWD_V2 0
#endif
#ifndef GLASS_SL_STAGED2
#define GLASS_SL_STAGED2 1  // comb pair's S / L weight-gradient tiles through LDS in 16-row stages (wgrad_sl_staged2_body)
#endif
#ifndef GLASS_COMB_DGRAD_V2
#define GLASS_COMB_DGRAD_V2 1  // comb data gradient at hidden 64 in the staged form inside the fused backward launch (comb_dgrad2_body)
#endif
#ifndef GLASS_TRANS_DGRAD_V2
#define GLASS_TRANS_DGRAD_V2 1  // trans data gradient at hidden 64 in the staged form (trans_dgrad2_body)
#endif
#ifndef GLASS_TRANS_WGRAD_STAGED2
#define GLASS_TRANS_WGRAD_STAGED2 1  // trans pair's weight-gradient slabs through LDS in 16-row stages (wgrad_trans_staged2_body)
#endif
#ifndef GLASS_TRANS_FWD_V2
#define GLASS_TRANS_FWD_V2 1  // trans forward at hidden 64 in the same form (trans_fwd2_kernel)
#endif
#ifndef GLASS_COMB_FWD_V2
#define GLASS_COMB_FWD_V2 1  // comb forward at hidden 64: weights in registers, rows through LDS in stages (comb_fwd_eff2_kernel)
#endif
#ifndef GLASS_FUSED_WGRAD_STAGES
#define GLASS_FUSED_WGRAD_STAGES 2  // pipeline stages of the weight-gradient workgroups inside the fused backward launches
#endif
#if !GLASS_LAB && (GLASS_WGRAD_STAGED || GLASS_COMB_BWD_V2 || !GLASS_SL_STAGED2 || !GLASS_COMB_DGRAD_V2 || !GLASS_TRANS_DGRAD_V2 ||       \
                   !GLASS_TRANS_WGRAD_STAGED2 || !GLASS_TRANS_FWD_V2 || !GLASS_COMB_FWD_V2)
#error "the variant switches above select laboratory forms whose kernels live in tools/lab/: build with -DGLASS_LAB=1 (tools/build_trace.sh)"
#endif
#ifdef GLASS_DENSE_TRACE
__device__ unsigned long long* g_dense_trace;
__device__ int g_dense_trace_sel;
#endif


typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKC = 16;  // K elements per lane per pass (64 per wave-step group)

__device__ __forceinline__ void load16(float (&dst)[kKC], const float* p, bool ok) {
#pragma unroll
    for (int v = 0; v < kKC / 4; ++v) {
        float4 t = ok ? *reinterpret_cast<const float4*>(p + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[4 * v] = t.x; dst[4 * v + 1] = t.y; dst[4 * v + 2] = t.z; dst[4 * v + 3] = t.w;
    }
}

// Output-column permutation: tile t = 4*grp + k holds, in lane j, output column 64*grp + 4*j + k, so the four
// tiles of a group give every lane FOUR CONSECUTIVE columns of a row -> 16-B stores / loads in the epilogue
// (the MFMA does not care which 16 weight rows form a tile).
__device__ __forceinline__ int tile_col(int t, int j) { return 64 * (t >> 2) + 4 * j + (t & 3); }

// ---- weights through LDS --------------------------------------------------------------------------
// Every wave needs the whole [NT x KT] weight; read straight from L2 that is 1000+ waves pulling the
// same 32-64 KiB in the same order (hot L2 lines: the first version ran at 18-22 us per launch for
// 3-4 us of MFMA work).  Instead the workgroup copies the weight slice of one K-pass into LDS once,
// already permuted into the order the lanes consume it:  image[(t*4 + v)*64 + lane] (float4) holds
// elements 4v..4v+3 of lane (j,q)'s 16-float chunk of weight row tile_col(t, j)  ->  every wave-level
// ds_read_b128 is one contiguous KiB (no bank conflicts) and the L2 sees one fetch per workgroup.
// The permutation itself is done once per training step for all weights by pack_batch_kernel.
template <int NT, int THREADS>
struct WStage {
    static constexpr int NTILES = NT / 16;
    static constexpr int kVecs = NTILES * 4 * 64;          // float4 per pass image
    static constexpr int kPerThread = kVecs / THREADS;     // staging float4 per thread (4 or 8 at hidden 64 and 128)
    static constexpr int kHalf = kPerThread / 2;
    static_assert(kPerThread % 2 == 0, "staging is held as two half arrays");
    // two arrays of <= 16 registers: a single 32-register array sits exactly at the compiler's promotion limit and went
    // to scratch memory inside the fused backward kernel
    float4 ra[kHalf], rb[kHalf];
    // global -> registers: pass kc of the PACKED weight (glass_dense_pack_batch_f32 wrote it in image
    // order once per step), so this is a fully coalesced 16-B-per-lane copy.  (Gathering the image from
    // the row-major weight here cost ~4 us per pass: 64 scattered 16-B reads per wave-instruction.)
    __device__ __forceinline__ void fetch(const float* __restrict__ Wimg, int kc) {
        const float4* src = reinterpret_cast<const float4*>(Wimg) + (int64_t)kc * kVecs;
#pragma unroll
        for (int n = 0; n < kHalf; ++n) {
            ra[n] = src[threadIdx.x + THREADS * n];
            rb[n] = src[threadIdx.x + THREADS * (kHalf + n)];
        }
    }
    // registers -> LDS image (consecutive threads write consecutive float4)
    __device__ __forceinline__ void commit(float4* __restrict__ image) const {
#pragma unroll
        for (int n = 0; n < kHalf; ++n) {
            image[threadIdx.x + THREADS * n] = ra[n];
            image[threadIdx.x + THREADS * (kHalf + n)] = rb[n];
        }
    }
};

// acc[t] += A_chunk . W_tile_chunk for every 16-column tile t, weights from the LDS image of this pass.
// Tiles go in pairs (two independent accumulators hide the 40-cycle dependent-MFMA latency); the next
// pair's chunks are read from LDS while the current pair's MFMAs issue.
// A wave may own only part of the output columns (column split, hidden 128: two wave groups per 16 rows): its
// NTILES local tiles are the image tiles base0 .. base0+HALF-1 followed by base1 .. (two runs: the f1 and f0
// halves of a Linear pair; HALF == NTILES: one run).
template <int NTILES, int HALF>
__device__ __forceinline__ void mfma_pass_lds(f32x4 (&acc)[NTILES], const float (&a)[kKC], const float4* image,
                                              int lane, int base0, int base1) {
    static_assert(NTILES % 2 == 0, "tiles are processed in pairs");
    auto read_tile = [&](int tl, float4 (&dst)[4]) __attribute__((always_inline)) {
        const int t = tl < HALF ? base0 + tl : base1 + (tl - HALF);
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[v] = image[(t * 4 + v) * 64 + lane];
    };
    auto mfma_pair = [&](int t, const float4 (&b0v)[4], const float4 (&b1v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float b0[4] = {b0v[v].x, b0v[v].y, b0v[v].z, b0v[v].w};
            const float b1[4] = {b1v[v].x, b1v[v].y, b1v[v].z, b1v[v].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * v + e], b0[e], acc[t], 0, 0, 0);
                acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * v + e], b1[e], acc[t + 1], 0, 0, 0);
            }
        }
    };
    float4 b[2][2][4];
    read_tile(0, b[0][0]);
    read_tile(1, b[0][1]);
#pragma unroll
    for (int t = 0; t < NTILES; t += 2) {
        const int cur = (t >> 1) & 1;
        if (t + 2 < NTILES) {
            read_tile(t + 2, b[cur ^ 1][0]);
            read_tile(t + 3, b[cur ^ 1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above ahead of the MFMAs below
        mfma_pair(t, b[cur][0], b[cur][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// Whole product for one wave's 16 rows: passes over K in chunks of 64 (16 per lane), weight slices double-buffered
// through LDS.  The A operand of pass kc is produced in two steps so that the loads of pass kc+1 stay in flight across the
// MFMAs of pass kc: `issue(kc, raw)` only starts the global loads into `raw`; `finish(kc, raw, a)` (run after the current
// pass) turns them into the operand chunk (prologue arithmetic).  `between()` runs once the loads of the FIRST pass are in
// flight: it may issue further loads and write LDS but holds no barrier (on CDNA a workgroup barrier drains the wave's
// outstanding loads — vmcnt counts loads and stores and the barrier's release fence waits for it — so a barrier there
// would serialise two memory round trips); the barrier behind the first commit publishes what it wrote.  The forward
// kernels derive their GraphNorm coefficients there, the accumulator loads joining the round trip of the operand loads.
// (Requesting BOTH passes up front — one round trip for everything — was slower: 17.8 vs 16.6 us for the comb forward,
// 18.0 vs 17.5 for the trans backward; the burst of every workgroup's loads is bandwidth-bound, and the pipelined form
// starts its MFMAs when half of the data has arrived.)
template <int NT, int KT, int NLOC, int HALF, int THREADS, typename Raw, typename Issue, typename Finish, typename Between = NoHook>
__device__ __forceinline__ void staged_product(f32x4 (&acc)[NLOC], const float* __restrict__ W, float4* lds, int lane,
                                               int base0, int base1, Issue issue, Finish finish, Between between = Between()) {
    constexpr int NKC = KT / 4 / kKC;
    constexpr int kVecs = WStage<NT, THREADS>::kVecs;
    static_assert(WStage<NT, THREADS>::kPerThread <= 8, "weight image too large for one staging object");
    WStage<NT, THREADS> ws;
    ws.fetch(W, 0);
    Raw raw;
    issue(0, raw);
    between();
    ws.commit(lds);
    __syncthreads();
    float a[kKC];
    finish(0, raw, a);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        if (kc + 1 < NKC) {
            ws.fetch(W, kc + 1);
            issue(kc + 1, raw);
        }
        __builtin_amdgcn_sched_barrier(0);  // the loads above are issued before the MFMAs below
        mfma_pass_lds<NLOC, HALF>(acc, a, lds + (kc & 1) * kVecs, lane, base0, base1);
        if (kc + 1 < NKC) {
            ws.commit(lds + ((kc + 1) & 1) * kVecs);  // the other image: last read in pass kc - 1, before the previous barrier
            finish(kc + 1, raw, a);
            __syncthreads();
        }
    }
}

// ---- forward ----------------------------------------------------------------------------------
// Optional GraphNorm prologue on the xa operand: xa is the INPUT of a GraphNorm whose statistics are already final
// (saved[4C] = mean, rstd, scale, shift); the lanes normalise (+ ELU + dropout) their chunk while loading it, use
// it as the MFMA operand and write it to `side` (the layer's backward and, for the trans pair, the comb pair of the
// same layer read it) — the GraphNorm apply launch and its read of xa disappear.
struct FwdRaw {
    float x[kKC];
};

// coef_s (LDS): scale[C] | shift[C] of the prologue's GraphNorm (gn_fwd_coef_block)
__device__ __forceinline__ void gn_prologue16(float (&a)[kKC], const float* coef_s, const GnPrologue& pro,
                                              const Drop& drop, int64_t row, int col0) {
#pragma unroll
    for (int v = 0; v < kKC / 4; ++v) {
        const float4 s4 = *reinterpret_cast<const float4*>(coef_s + col0 + 4 * v);
        const float4 h4 = *reinterpret_cast<const float4*>(coef_s + pro.C + col0 + 4 * v);
        const float scale[4] = {s4.x, s4.y, s4.z, s4.w}, shift[4] = {h4.x, h4.y, h4.z, h4.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop.p > 0.f) drop_scales<4>(drop, row, col0 + 4 * v, ds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float h = fmaf(a[4 * v + k], scale[k], shift[k]);
            h = act_fast(pro.act, h);
            a[4 * v + k] = h * ds[k];
        }
        if (pro.side)
            *reinterpret_cast<float4*>(pro.side + row * pro.lds + col0 + 4 * v) =
                make_float4(a[4 * v], a[4 * v + 1], a[4 * v + 2], a[4 * v + 3]);
    }
}

// CS = column split: CS wave groups of 4 waves share the same 64 rows, each owning 1/CS of the 64-column output
// groups (hidden 64: CS = 1, a wave holds all 8 tiles; hidden 128: CS = 2, 8 of the 16 tiles per wave -> the
// accumulators, staging registers and operand chunks fit the register file without spills).  Only CS = 1 (hidden 64) is
// instantiated now: hidden 128 runs on dense_tiled.hip.
template <int H, bool COMB, int CS, int RW>
__global__ __launch_bounds__(kWave * RW * CS) void dual_fwd_kernel(const float* __restrict__ xa, int64_t lda,
                                                               const float* __restrict__ xb, int64_t ldb,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               const uint8_t* __restrict__ mask, float zr, float omz,
                                                               int act, float* __restrict__ T, int64_t ldt,
                                                               float* __restrict__ out, int64_t ldo, int64_t N,
                                                               double* __restrict__ stats, int stats_exact, GnPrologue pro,
                                                               const int64_t* __restrict__ xa_index, int xa_rows) {
    constexpr int KT = COMB ? 2 * H : H, KQ = KT / 4, NT = 2 * H;
    constexpr int THREADS = kWave * RW * CS;  // RW row waves (16 rows each) x CS column groups
    constexpr int NG = H / 64;       // 64-column groups per half
    constexpr int NGL = NG / CS;     // ... owned by one wave
    constexpr int NLOC = 8 * NGL;    // local tiles: 4*NGL of the f1 half, then 4*NGL of the f0 half
    static_assert(KQ % kKC == 0 && NG % CS == 0, "hidden size must be a multiple of 64 * CS");
    D_STAMP(2, 0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rw = w % RW, cg = w / RW;  // row wave, column group
    const int i = lane & 15, q = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * RW + rw) * 16;
    const int64_t row = row0 + i;
    const bool row_ok = row < N;
    // this lane's K-chunk of its A row: q*KQ .. (q+1)*KQ of [xa || xb]
    const float* arow;
    if (!COMB) {
        // xa_index: row n of the operand is row xa_index[n] of xa (the embedding table: the lookup of
        // impl/models.py:248 happens in this load; the prologue's side output is the [N,H] layer input)
        int64_t src = row;
        if (xa_index) {
            src = row_ok ? xa_index[row] : 0;
            src = src < 0 ? 0 : (src >= xa_rows ? xa_rows - 1 : src);
        }
        arow = xa + src * lda + q * KQ;
    } else {
        arow = (q < 2) ? xa + row * lda + q * KQ : xb + row * ldb + (q - 2) * KQ;  // KQ = H/2
    }
    extern __shared__ float4 lds_w[];
    f32x4 acc[NLOC];
#pragma unroll
    for (int t = 0; t < NLOC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    const bool pro_lane = pro.saved != nullptr && row_ok && (!COMB || q < 2);  // lanes whose chunk belongs to xa
    GnPrologue pro_w = pro;
    if (cg != 0) pro_w.side = nullptr;  // the wave groups of a row tile compute the same operand; one writes it
    // scale | shift of the prologue's GraphNorm in LDS: copied from `saved`, or derived here from the exact accumulators its
    // producers added to (gn_acc.h: no finalize launch between them and this kernel)
    __shared__ __attribute__((aligned(16))) float gn_coef_s[2 * H];
    staged_product<NT, KT, NLOC, 4 * NGL, THREADS, FwdRaw>(
        acc, W, lds_w, lane, 4 * NGL * cg, 4 * (NG + NGL * cg),
        [&](int kc, FwdRaw& raw) __attribute__((always_inline)) { load16(raw.x, arow + kc * kKC, row_ok); },
        [&](int kc, const FwdRaw& raw, float (&a)[kKC]) __attribute__((always_inline)) {
            if (kc == 0) D_STAMP(2, 5);
#pragma unroll
            for (int s2 = 0; s2 < kKC; ++s2) a[s2] = raw.x[s2];
            if (pro_lane) gn_prologue16(a, gn_coef_s, pro_w, drop, row, q * KQ + kc * kKC);
            if (kc == 0) D_STAMP(2, 6);
        },
        [&]() __attribute__((always_inline)) {  // (its accumulator loads join the round trip of the operand loads above)
            if (pro.saved) gn_fwd_coef_nobarrier<H, THREADS>(pro.src, pro.saved, N, gn_coef_s);
            D_STAMP(2, 1);
        });
    D_STAMP(2, 2);
    // epilogue: acc[4gl+k][reg] is row row0 + 4q + reg, column 64g + 4i + k (g = NGL*cg + gl) -> float4 per (row, group)
    float ssum[NGL][4], ssq[NGL][4];  // this lane's column sums over its (up to) 4 rows, for the GraphNorm that follows
#pragma unroll
    for (int g = 0; g < NGL; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) ssum[g][k] = ssq[g][k] = 0.f;
    float4 bv[2 * NGL];
#pragma unroll
    for (int gl = 0; gl < NGL; ++gl) {
        bv[gl] = *reinterpret_cast<const float4*>(bias + 64 * (NGL * cg + gl) + 4 * i);
        bv[NGL + gl] = *reinterpret_cast<const float4*>(bias + H + 64 * (NGL * cg + gl) + 4 * i);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t r = row0 + 4 * q + reg;
        if (r >= N) continue;
        const float w1 = mask[r] ? zr : omz, w0 = mask[r] ? omz : zr;
#pragma unroll
        for (int gl = 0; gl < NGL; ++gl) {
            const int c = 64 * (NGL * cg + gl) + 4 * i;
            float v1[4], v0[4];
            const float b1[4] = {bv[gl].x, bv[gl].y, bv[gl].z, bv[gl].w};
            const float b0[4] = {bv[NGL + gl].x, bv[NGL + gl].y, bv[NGL + gl].z, bv[NGL + gl].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v1[k] = acc[4 * gl + k][reg] + b1[k];
                v0[k] = acc[4 * (NGL + gl) + k][reg] + b0[k];
            }
            if (T) {
                *reinterpret_cast<float4*>(T + r * ldt + c) = make_float4(v1[0], v1[1], v1[2], v1[3]);
                *reinterpret_cast<float4*>(T + r * ldt + H + c) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a1 = v1[k], a0 = v0[k];
                a1 = act_fast(act, a1), a0 = act_fast(act, a0);
                o[k] = w1 * a1 + w0 * a0;
                ssum[gl][k] += o[k];
                ssq[gl][k] = fmaf(o[k], o[k], ssq[gl][k]);
            }
            *reinterpret_cast<float4*>(out + r * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    D_STAMP(2, 3);
    if (stats == nullptr) return;
    // Column statistics of `out` for the GraphNorm that consumes it (its statistics pass is skipped):
    // stats[blockIdx.x][2][H] = per-workgroup sum / sum of squares over its 16*RW rows, in fp64 from here on.
    __syncthreads();  // every wave is done with the weight images in LDS
    double* red = reinterpret_cast<double*>(lds_w);  // [RW row waves][H][2]
#pragma unroll
    for (int gl = 0; gl < NGL; ++gl)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double s = (double)ssum[gl][k], q2 = (double)ssq[gl][k];
            s += __shfl_xor(s, 16);
            q2 += __shfl_xor(q2, 16);
            s += __shfl_xor(s, 32);
            q2 += __shfl_xor(q2, 32);
            if (q == 0) {
                const int c = 64 * (NGL * cg + gl) + 4 * i + k;
                red[(rw * H + c) * 2] = s;
                red[(rw * H + c) * 2 + 1] = q2;
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += THREADS) {
        double s = 0.0, q2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < RW; ++ww) {
            s += red[(ww * H + c) * 2];
            q2 += red[(ww * H + c) * 2 + 1];
        }
        if (stats_exact) {  // exact accumulators (gn_acc.h): the consumer folds them, no finalize launch
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 0, c, H, s, kAccScaleFwd);
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 1, c, H, q2, kAccScaleFwd);
        } else {
            stats[((size_t)blockIdx.x * 2) * H + c] = s;
            stats[((size_t)blockIdx.x * 2 + 1) * H + c] = q2;
        }
    }    D_STAMP(2, 4);
}

struct DgradRaw {
    float d[kKC], t[kKC];
};

// Optional epilogue of the data-gradient kernel: its first H output columns are the gradient dy of a GraphNorm
// OUTPUT (conv.gn's for the comb pair, gns[l]'s for the next layer's trans pair), so the two backward column sums of
// that GraphNorm — S1 = sum g, S2 = sum g*xhat with g = dy * dropmask * act'(x*scale + shift) — are accumulated here
// from the tile in registers plus one read of the GraphNorm input x, instead of by a statistics launch re-reading
// dy and x.  partial[blockIdx.x][2][H] doubles, consumed by glass_graphnorm_bwd_from_stats_f32.
// ---- backward data gradient ---------------------------------------------------------------------
// out[N, NT] = dZ[N, 2H] @ Wstack[2H, NT] (+ addend), dZ[n, o] = coef(n, o<H) * dsrc[n, o mod H] * act'(T[n, o]);
// WT = Wstack^T stored [NT][2H] so that weight chunks are contiguous (refreshed once per step).
struct DgradArgs {
    const float* dsrc; int64_t ldd;
    const float* T; int64_t ldt;
    const uint8_t* mask; float zr, omz;
    int act;
    const float* WT;
    const float* addend; int64_t ldadd;
    Drop drop;
    const uint64_t* rng_state;
    float* out; int64_t ldo;
    int64_t N;
    GnBwdStats gs;
};

// One row tile (`block`) of the data gradient; lds_w = the workgroup's dynamic LDS (weight images, then the statistics
// reduction).  A device function so that it can also be one branch of the fused backward launch (dual_bwd_kernel).
// Plain by-value arguments: with the argument struct passed by reference the compiler kept the weight staging registers
// of the wider variants in scratch memory.
template <int H, int NT, int CS, int RW>
__device__ __forceinline__ void dual_dgrad_body(const float* __restrict__ dsrc, int64_t ldd, const float* __restrict__ T,
                                                int64_t ldt, const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                const float* __restrict__ WT, const float* __restrict__ addend,
                                                int64_t ldadd, Drop drop, const uint64_t* __restrict__ rng_state,
                                                float* __restrict__ out, int64_t ldo, int64_t N, GnBwdStats gs, int block,
                                                float4* lds_w) {
    constexpr int KT = 2 * H, KQ = KT / 4;
    constexpr int THREADS = kWave * RW * CS;
    constexpr int NGO = NT / 64;      // 64-column output groups
    constexpr int NGL = NGO / CS;     // ... owned by one wave (one contiguous run of 4*NGL tiles)
    constexpr int NLOC = 4 * NGL;
    static_assert(KQ % kKC == 0 && NGO % CS == 0 && NLOC % 2 == 0, "unsupported shape");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rw = w % RW, cg = w / RW;
    const int i = lane & 15, q = lane >> 4;
    const int64_t row0 = ((int64_t)block * RW + rw) * 16;
    const int64_t row = row0 + i;
    const bool row_ok = row < N;
    const bool first = q < 2;  // lanes q=0,1 hold the f1 half (o < H), q=2,3 the f0 half
    float coef = 0.f;
    if (row_ok) coef = (mask[row] != 0) == first ? zr : omz;
    const float* drow = dsrc + row * ldd + (q & 1) * KQ;   // o mod H
    const float* trow = T ? T + row * ldt + q * KQ : nullptr;
    f32x4 acc[NLOC];
#pragma unroll
    for (int t = 0; t < NLOC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    staged_product<NT, KT, NLOC, NLOC, THREADS, DgradRaw>(
        acc, WT, lds_w, lane, NLOC * cg, 0,
        [&](int kc, DgradRaw& raw) __attribute__((always_inline)) {
            load16(raw.d, drow + kc * kKC, row_ok);
            if (act != GLASS_ACT_NONE) load16(raw.t, trow + kc * kKC, row_ok);
        },
        [&](int, const DgradRaw& raw, float (&a)[kKC]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < kKC; ++s) {
                float v = raw.d[s] * coef;
                if (act != GLASS_ACT_NONE) v *= act_grad(act, raw.t[s]);
                a[s] = v;
            }
        });
    D_STAMP(4, 2);
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    constexpr int NGS = H / 64;  // 64-column groups of the GraphNorm half (global groups 0 .. NGS-1)
    float s1[NGL][4], s2[NGL][4];
    float4 g_mu[NGL], g_rstd[NGL], g_scale[NGL], g_shift[NGL], g_al[NGL];
    if (gs.partial) {
        if (gs.drop.p > 0.f) {
            gs.drop.seed = rng_state[0];
            gs.drop.step = rng_state[1];
        }
#pragma unroll
        for (int gl = 0; gl < NGL; ++gl) {
            const int g = NGL * cg + gl;
            const int c = 64 * (g < NGS ? g : 0) + 4 * i;
            g_mu[gl] = *reinterpret_cast<const float4*>(gs.saved + c);
            g_rstd[gl] = *reinterpret_cast<const float4*>(gs.saved + H + c);
            g_scale[gl] = *reinterpret_cast<const float4*>(gs.saved + 2 * H + c);
            g_shift[gl] = *reinterpret_cast<const float4*>(gs.saved + 3 * H + c);
            g_al[gl] = *reinterpret_cast<const float4*>(gs.alpha + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) s1[gl][k] = s2[gl][k] = 0.f;
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t r = row0 + 4 * q + reg;
        if (r >= N) continue;
#pragma unroll
        for (int gl = 0; gl < NGL; ++gl) {
            const int g = NGL * cg + gl;
            const int c = 64 * g + 4 * i;
            float4 v = make_float4(acc[4 * gl][reg], acc[4 * gl + 1][reg], acc[4 * gl + 2][reg], acc[4 * gl + 3][reg]);
            if (addend) {
                const float4 ad = *reinterpret_cast<const float4*>(addend + r * ldadd + c);
                v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
            }
            if (drop.p > 0.f) {  // gradient w.r.t. the pre-dropout tensor: same mask as the forward drew
                float ds[4];
                drop_scales<4>(drop, r, c, ds);
                v.x *= ds[0]; v.y *= ds[1]; v.z *= ds[2]; v.w *= ds[3];
            }
            *reinterpret_cast<float4*>(out + r * ldo + c) = v;
            if (gs.partial && g < NGS) {
                const float4 x4 = *reinterpret_cast<const float4*>(gs.x + r * gs.ldx + c);
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, dy[4] = {v.x, v.y, v.z, v.w};
                const float mu[4] = {g_mu[gl].x, g_mu[gl].y, g_mu[gl].z, g_mu[gl].w};
                const float rs[4] = {g_rstd[gl].x, g_rstd[gl].y, g_rstd[gl].z, g_rstd[gl].w};
                const float sc[4] = {g_scale[gl].x, g_scale[gl].y, g_scale[gl].z, g_scale[gl].w};
                const float sh[4] = {g_shift[gl].x, g_shift[gl].y, g_shift[gl].z, g_shift[gl].w};
                const float al[4] = {g_al[gl].x, g_al[gl].y, g_al[gl].z, g_al[gl].w};
                float ds[4] = {1.f, 1.f, 1.f, 1.f};
                if (gs.drop.p > 0.f) drop_scales<4>(gs.drop, r, c, ds);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float gp = dy[k] * ds[k];
                    gp *= act_grad(gs.act, fmaf(xv[k], sc[k], sh[k]));
                    const float xhat = (xv[k] - al[k] * mu[k]) * rs[k];
                    s1[gl][k] += gp;
                    s2[gl][k] = fmaf(gp, xhat, s2[gl][k]);
                }
            }
        }
    }
    D_STAMP(4, 3);
    if (gs.partial == nullptr) return;
    __syncthreads();  // every wave is done with the weight images in LDS
    double* red = reinterpret_cast<double*>(lds_w);  // [RW row waves][H][2]
#pragma unroll
    for (int gl = 0; gl < NGL; ++gl) {
        const int g = NGL * cg + gl;
        if (g >= NGS) continue;  // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double a = (double)s1[gl][k], b2 = (double)s2[gl][k];
            a += __shfl_xor(a, 16);
            b2 += __shfl_xor(b2, 16);
            a += __shfl_xor(a, 32);
            b2 += __shfl_xor(b2, 32);
            if (q == 0) {
                red[(rw * H + 64 * g + 4 * i + k) * 2] = a;
                red[(rw * H + 64 * g + 4 * i + k) * 2 + 1] = b2;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += THREADS) {
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < RW; ++ww) {
            a += red[(ww * H + c) * 2];
            b2 += red[(ww * H + c) * 2 + 1];
        }
        if (gs.exact) {  // exact accumulators (gn_acc.h): no finalize launch behind this kernel
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 0, c, H, a, kAccScaleBwd);
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 1, c, H, b2, kAccScaleBwd);
        } else {
            gs.partial[((size_t)block * 2) * H + c] = a;
            gs.partial[((size_t)block * 2 + 1) * H + c] = b2;
        }
    }
}

// ---- trans data gradient in the staged form (hidden 64): out[N,64] = dZ[N,128] . Wstack (+ addend) (* the input's dropout mask)
// Same structure as the staged forward kernels: a wave owns 16 output columns (its slice of W^T: 32 registers), the rows
// go through LDS in four 16-row stages.  The LOADER threads synthesise dZ = mix'(dout) . ELU'(T) (eight elements each) and
// prepare, per element of the output tile, what the epilogue needs as plain multiply-adds: the addend, the keep-scale of
// the layer input's dropout, and for the GraphNorm whose output gradient this is  u = keep-scale . act'(x scale + shift)
// and  xhat . u  — one dropout hash per float4, none in the epilogue; no conditional memory instruction in the stage loop.
// Image: layout kLayoutWave16Cols of W^T ([64 outputs][128 = f1 | f0]).
// SP (hidden 64, inside the fused backward launch): the product in the split form (split_mma.h) — the loader cuts its eight
// elements of dZ once into three bf16 planes ([piece][row][k], rows 136 bf16 apart), the wave its weight slice once; 24 MFMAs
// of 16 cycles per stage instead of 32 of 32.  In the fused launch two workgroups share a CU's matrix pipes (this body's tiles and
// the weight-gradient slabs): there the matrix time IS the launch time (a timing-only build that skipped 5 of 8 MFMAs in the
// four backward bodies: 0.2418 -> 0.2311 ms per step at ppi_bp-shape).
template <int H, bool SP = false>
__device__ __forceinline__ void trans_dgrad2_body(const float* __restrict__ dsrc, int64_t ldd, const float* __restrict__ T,
                                                  int64_t ldt, const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                  const float* __restrict__ WT, const float* __restrict__ addend, int64_t ldadd,
                                                  Drop drop, const uint64_t* __restrict__ rng_state, float* __restrict__ out,
                                                  int64_t ldo, int64_t N, GnBwdStats gs, int block, float* lds) {
    static_assert(H == 64 || H == 128, "H / 16 waves x 16 columns");
    constexpr int NTL = H / 16, KF4 = (2 * H) / 16;
    constexpr int KT = 2 * H, RA = KT + 4, RP = H + 4;
    static_assert(!SP || H == 64, "split form: hidden 64");
    constexpr int RSB = KT + 8, kPlane = 16 * RSB;  // SP: bf16 per row of a piece plane, per plane
    constexpr int kAFloats = SP ? (3 * kPlane) / 2 : 16 * RA;
    constexpr int kBuf = kAFloats + 4 * 16 * RP;  // A | ADD | M | U | XU  (floats per stage buffer)
    int* rows_s = reinterpret_cast<int*>(lds + 2 * kBuf);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    const int64_t r0 = (int64_t)block * 64;
    const buf_rsrc r_d = make_rsrc(dsrc, N * ldd * 4), r_t = make_rsrc(T ? T : dsrc, T ? N * ldt * 4 : 0);
    const buf_rsrc r_m = make_rsrc(mask, N), r_out = make_rsrc(out, N * ldo * 4);
    const buf_rsrc r_add = make_rsrc(addend ? addend : dsrc, addend ? N * ldadd * 4 : 0);
    const buf_rsrc r_x = make_rsrc(gs.partial ? gs.x : dsrc, gs.partial ? N * gs.ldx * 4 : 0);
    const float4* img = reinterpret_cast<const float4*>(WT);
    float4 bw[KF4];
#pragma unroll
    for (int tt = 0; tt < KF4; ++tt) bw[tt] = img[(((tt >> 2) * NTL + w) * 4 + (tt & 3)) * 64 + lane];
    uint4 bwc[SP ? KF4 / 2 : 1][3];  // SP: block b = the lane's k = (KT / 4) q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < KF4 / 2; ++b) {
            const Split4 c0 = split4(bw[2 * b]), c1 = split4(bw[2 * b + 1]);
            bwc[b][0] = make_uint4(c0.hi.x, c0.hi.y, c1.hi.x, c1.hi.y);
            bwc[b][1] = make_uint4(c0.mid.x, c0.mid.y, c1.mid.x, c1.mid.y);
            bwc[b][2] = make_uint4(c0.lo.x, c0.lo.y, c1.lo.x, c1.lo.y);
        }
    }
    int my_row[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) my_row[st] = r0 + 16 * st + rs < N ? (int)(r0 + 16 * st + rs) : -1;
    struct Raw {
        float4 d, t1, t0, ad, x;
        unsigned mk;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int r = my_row[st];
        const bool ok = r >= 0;
        R.d = buf_load4(r_d, ok ? (int)((r * ldd + 4 * ga) * 4) : kBufOOB);
        R.t1 = buf_load4(r_t, ok ? (int)((r * ldt + 4 * ga) * 4) : kBufOOB);
        R.t0 = buf_load4(r_t, ok ? (int)((r * ldt + H + 4 * ga) * 4) : kBufOOB);
        R.ad = buf_load4(r_add, ok ? (int)((r * ldadd + 4 * ga) * 4) : kBufOOB);
        R.x = buf_load4(r_x, ok ? (int)((r * gs.ldx + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_m, ok ? r : kBufOOB, 0, 0);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    const bool drop_on = drop.p > 0.f, gn_on = gs.partial != nullptr, gdrop_on = gn_on && gs.drop.p > 0.f;
    Drop gdrop = gs.drop;
    if (drop_on || gdrop_on) {
        drop.seed = gdrop.seed = rng_state[0];
        drop.step = gdrop.step = rng_state[1];
    }
    float g_mu[4] = {0.f, 0.f, 0.f, 0.f}, g_rs[4] = {0.f, 0.f, 0.f, 0.f}, g_al[4] = {0.f, 0.f, 0.f, 0.f};
    float g_sc[4] = {0.f, 0.f, 0.f, 0.f}, g_sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (gn_on) {
        const float4 m4 = *reinterpret_cast<const float4*>(gs.saved + 4 * ga), r4 = *reinterpret_cast<const float4*>(gs.saved + H + 4 * ga);
        const float4 s4 = *reinterpret_cast<const float4*>(gs.saved + 2 * H + 4 * ga), h4 = *reinterpret_cast<const float4*>(gs.saved + 3 * H + 4 * ga);
        const float4 a4 = *reinterpret_cast<const float4*>(gs.alpha + 4 * ga);
        g_mu[0] = m4.x, g_mu[1] = m4.y, g_mu[2] = m4.z, g_mu[3] = m4.w;
        g_rs[0] = r4.x, g_rs[1] = r4.y, g_rs[2] = r4.z, g_rs[3] = r4.w;
        g_sc[0] = s4.x, g_sc[1] = s4.y, g_sc[2] = s4.z, g_sc[3] = s4.w;
        g_sh[0] = h4.x, g_sh[1] = h4.y, g_sh[2] = h4.z, g_sh[3] = h4.w;
        g_al[0] = a4.x, g_al[1] = a4.y, g_al[2] = a4.z, g_al[3] = a4.w;
    }
    if (tid < 64) rows_s[tid] = r0 + tid < N ? (int)(r0 + tid) : -1;
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        float* A = lds + (st & 1) * kBuf;
        float* ADD = A + kAFloats;
        float* M = ADD + 16 * RP;
        float* U = M + 16 * RP;
        float* XU = U + 16 * RP;
        const int r = my_row[st] < 0 ? 0 : my_row[st];
        const float c1 = R.mk ? zr : omz, c0 = R.mk ? omz : zr;
        const float d[4] = {R.d.x, R.d.y, R.d.z, R.d.w}, t1[4] = {R.t1.x, R.t1.y, R.t1.z, R.t1.w}, t0[4] = {R.t0.x, R.t0.y, R.t0.z, R.t0.w};
        const float xv[4] = {R.x.x, R.x.y, R.x.z, R.x.w};
        float z1[4], z0[4], m[4] = {1.f, 1.f, 1.f, 1.f}, gds[4] = {1.f, 1.f, 1.f, 1.f}, u[4], xu[4];
        if (drop_on) drop_scales<4>(drop, r, 4 * ga, m);
        if (gdrop_on) drop_scales<4>(gdrop, r, 4 * ga, gds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z1[k] = d[k] * c1;
            z0[k] = d[k] * c0;
            z1[k] *= act_grad(act, t1[k]), z0[k] *= act_grad(act, t0[k]);
            float uk = gds[k];
            uk *= act_grad(gs.act, fmaf(xv[k], g_sc[k], g_sh[k]));
            u[k] = uk;
            xu[k] = (xv[k] - g_al[k] * g_mu[k]) * g_rs[k] * uk;
        }
        if constexpr (SP) {
            const Split4 s1c = split4(make_float4(z1[0], z1[1], z1[2], z1[3])), s0c = split4(make_float4(z0[0], z0[1], z0[2], z0[3]));
            unsigned short* P = reinterpret_cast<unsigned short*>(A) + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = s1c.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = s1c.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = s1c.lo;
            *reinterpret_cast<uint2*>(P + H) = s0c.hi;
            *reinterpret_cast<uint2*>(P + kPlane + H) = s0c.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane + H) = s0c.lo;
        } else {
            *reinterpret_cast<float4*>(A + rs * RA + 4 * ga) = make_float4(z1[0], z1[1], z1[2], z1[3]);
            *reinterpret_cast<float4*>(A + rs * RA + H + 4 * ga) = make_float4(z0[0], z0[1], z0[2], z0[3]);
        }
        *reinterpret_cast<float4*>(ADD + rs * RP + 4 * ga) = R.ad;
        *reinterpret_cast<float4*>(M + rs * RP + 4 * ga) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(U + rs * RP + 4 * ga) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4*>(XU + rs * RP + 4 * ga) = make_float4(xu[0], xu[1], xu[2], xu[3]);
    };
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    float s1 = 0.f, s2 = 0.f;
    const int c = 16 * w + j;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const float* A = lds + (st & 1) * kBuf;
        const float* ADD = A + kAFloats;
        const float* M = ADD + 16 * RP;
        const float* U = M + 16 * RP;
        const float* XU = U + 16 * RP;
        float4 a4[SP ? 1 : KF4];
        uint4 af[SP ? KF4 / 2 : 1][3];
        if constexpr (SP) {
            const unsigned short* P = reinterpret_cast<const unsigned short*>(A) + j * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[b][pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
        } else {
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(A + j * RA + (KT / 4) * q + 4 * tt);
        }
        int rv[4];
        float ad[4], mm[4], uu[4], xx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rv[r] = rows_s[16 * st + 4 * q + r];
            ad[r] = ADD[(4 * q + r) * RP + c];
            mm[r] = M[(4 * q + r) * RP + c];
            uu[r] = U[(4 * q + r) * RP + c];
            xx[r] = XU[(4 * q + r) * RP + c];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
#define GLASS_SMMA16(ACC, B, pa, pb)                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[B][pa]), __builtin_bit_cast(bf16x8, bwc[B][pb]), ACC, 0, 0, 0)
#pragma unroll
            for (int b = 0; b < KF4 / 2; b += 2) {  // two K blocks side by side (two chains), small terms first
                GLASS_SMMA16(acc0, b, 1, 1); GLASS_SMMA16(acc1, b + 1, 1, 1);
                GLASS_SMMA16(acc0, b, 2, 0); GLASS_SMMA16(acc1, b + 1, 2, 0);
                GLASS_SMMA16(acc0, b, 0, 2); GLASS_SMMA16(acc1, b + 1, 0, 2);
                GLASS_SMMA16(acc0, b, 1, 0); GLASS_SMMA16(acc1, b + 1, 1, 0);
                GLASS_SMMA16(acc0, b, 0, 1); GLASS_SMMA16(acc1, b + 1, 0, 1);
                GLASS_SMMA16(acc0, b, 0, 0); GLASS_SMMA16(acc1, b + 1, 0, 0);
            }
#undef GLASS_SMMA16
        } else {
#pragma unroll
            for (int tt = 0; tt < KF4; tt += 2) {
                const float x0[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w}, y0[4] = {bw[tt].x, bw[tt].y, bw[tt].z, bw[tt].w};
                const float x1[4] = {a4[tt + 1].x, a4[tt + 1].y, a4[tt + 1].z, a4[tt + 1].w};
                const float y1[4] = {bw[tt + 1].x, bw[tt + 1].y, bw[tt + 1].z, bw[tt + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!GLASS_MFMA_KEEP(tt * 2 + e)) continue;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], y0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], y1[e], acc1, 0, 0, 0);
                }
            }
        }
        if (st + 1 < 4) {
            commit(st + 1, (st & 1) ? rawA : rawB);
            if (st + 3 < 4) {
                if (st & 1) issue(st + 3, rawA); else issue(st + 3, rawB);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0;
            const float v = (acc0[r] + acc1[r] + ad[r]) * mm[r];
            buf_store1(r_out, live ? (int)((rv[r] * ldo + c) * 4) : kBufOOB, v);
            const float vl = live ? v : 0.f;
            s1 = fmaf(vl, uu[r], s1);
            s2 = fmaf(vl, xx[r], s2);
        }
        if (st + 1 < 4) lds_barrier();
    }
    if (!gn_on) return;
    double a = (double)s1, b2 = (double)s2;
    a += __shfl_xor(a, 16);
    b2 += __shfl_xor(b2, 16);
    a += __shfl_xor(a, 32);
    b2 += __shfl_xor(b2, 32);
    if (q == 0) {
        if (gs.exact) {
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 0, c, H, a, kAccScaleBwd);
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 1, c, H, b2, kAccScaleBwd);
        } else {
            gs.partial[((size_t)block * 2) * H + c] = a;
            gs.partial[((size_t)block * 2 + 1) * H + c] = b2;
        }
    }
}
// The same body with a RUN-TIME number of stages ("tall" row tiles, see comb_fwd_eff3_kernel): a workgroup takes
// `stages_per_wg` consecutive 16-row stages, loads its weight slice and the GraphNorm statistics once and keeps the
// double-buffered stage pipeline running across them.  The split is by STAGES, not by 64-row tiles (round 6): 50 000 rows are
// 782 tiles = 196 workgroups of 4 tiles — 60 of the 256 CUs idle, the others 16 stages deep —, but 3 125 stages = 241
// workgroups of 13.  The backward column sums of a GraphNorm keep their ABI (one entry per 64-row tile, summed by the
// consumer): workgroup b writes ITS sums (float per lane over four stages, double across) to entry b and zeroes the entries
// b + grid, b + 2 grid, ... < n_tiles (grid <= n_tiles: a workgroup covers at least four stages).
// SP: the product in the split form (as trans_dgrad2_body at hidden 64): the loader cuts its eight elements of dZ into three
// bf16 planes, the wave its weight slice once (96 registers); 48 MFMAs of 16 cycles per stage and wave where the f32-input
// form issues 64 of 32.  The GraphNorm's column constants then live in LDS and the epilogue operands are read behind the
// products (register budget: 256).
template <int H, bool SP = false>
__global__ __launch_bounds__(4 * H) void trans_dgrad3_kernel(DgradArgs A, int stages_per_wg) {
    static_assert(H == 64 || H == 128, "H / 16 waves x 16 columns");
    extern __shared__ float4 lds_w3[];
    float* lds = reinterpret_cast<float*>(lds_w3);
    constexpr int NTL = H / 16, KF4 = (2 * H) / 16;
    constexpr int KT = 2 * H, RA = KT + 4, RP = H + 4;
    constexpr int RSB = KT + 8, kPlane = 16 * RSB;  // SP: bf16 per row of a piece plane, per plane
    constexpr int kAFloats = SP ? (3 * kPlane) / 2 : 16 * RA;
    constexpr int kBuf = kAFloats + 4 * 16 * RP;  // A | ADD | M | U | XU  (floats per stage buffer)
    int* rows_s = reinterpret_cast<int*>(lds + 2 * kBuf);  // [2][16]: the rows of the stage in each buffer
    const float* __restrict__ dsrc = A.dsrc;
    const int64_t ldd = A.ldd, ldt = A.ldt, ldadd = A.ldadd, ldo = A.ldo, N = A.N;
    const float zr = A.zr, omz = A.omz;
    const int act = A.act;
    const GnBwdStats gs = A.gs;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    const int64_t stage0 = (int64_t)blockIdx.x * stages_per_wg;
    const int64_t n_stages_all = (N + 15) / 16, n_tiles_all = (N + 63) / 64;
    const int nst = (int)(stage0 + stages_per_wg <= n_stages_all ? stages_per_wg : n_stages_all - stage0);
    const int64_t r0 = stage0 * 16;
    const buf_rsrc r_d = make_rsrc(dsrc, N * ldd * 4), r_t = make_rsrc(A.T ? A.T : dsrc, A.T ? N * ldt * 4 : 0);
    const buf_rsrc r_m = make_rsrc(A.mask, N), r_out = make_rsrc(A.out, N * ldo * 4);
    const buf_rsrc r_add = make_rsrc(A.addend ? A.addend : dsrc, A.addend ? N * ldadd * 4 : 0);
    const buf_rsrc r_x = make_rsrc(gs.partial ? gs.x : dsrc, gs.partial ? N * gs.ldx * 4 : 0);
    struct Raw {
        float4 d, t1, t0, ad, x;
        unsigned mk;
    };
    auto row_of = [&](int st) __attribute__((always_inline)) -> int {  // this thread's row of stage st (-1: none)
        const int64_t rr = r0 + 16 * st + rs;
        return (st < nst && rr < N) ? (int)rr : -1;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int r = row_of(st);
        const bool ok = r >= 0;
        R.d = buf_load4(r_d, ok ? (int)((r * ldd + 4 * ga) * 4) : kBufOOB);
        R.t1 = buf_load4(r_t, ok ? (int)((r * ldt + 4 * ga) * 4) : kBufOOB);
        R.t0 = buf_load4(r_t, ok ? (int)((r * ldt + H + 4 * ga) * 4) : kBufOOB);
        R.ad = buf_load4(r_add, ok ? (int)((r * ldadd + 4 * ga) * 4) : kBufOOB);
        R.x = buf_load4(r_x, ok ? (int)((r * gs.ldx + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_m, ok ? r : kBufOOB, 0, 0);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    const float4* img = reinterpret_cast<const float4*>(A.WT);
    float4 bw[KF4];
#pragma unroll
    for (int tt = 0; tt < KF4; ++tt) bw[tt] = img[(((tt >> 2) * NTL + w) * 4 + (tt & 3)) * 64 + lane];
    uint4 bwc[SP ? KF4 / 2 : 1][3];  // SP: block b = the lane's k = (KT / 4) q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < KF4 / 2; ++b) {
            const Split4 c0 = split4(bw[2 * b]), c1 = split4(bw[2 * b + 1]);
            bwc[b][0] = make_uint4(c0.hi.x, c0.hi.y, c1.hi.x, c1.hi.y);
            bwc[b][1] = make_uint4(c0.mid.x, c0.mid.y, c1.mid.x, c1.mid.y);
            bwc[b][2] = make_uint4(c0.lo.x, c0.lo.y, c1.lo.x, c1.lo.y);
        }
    }
    Drop drop = A.drop;
    const bool drop_on = drop.p > 0.f, gn_on = gs.partial != nullptr, gdrop_on = gn_on && gs.drop.p > 0.f;
    Drop gdrop = gs.drop;
    if (drop_on || gdrop_on) {
        drop.seed = gdrop.seed = A.rng_state[0];
        drop.step = gdrop.step = A.rng_state[1];
    }
    // the GraphNorm's per-column constants of this thread's four columns: registers, or (SP) LDS
    float g_mu[4] = {0.f, 0.f, 0.f, 0.f}, g_rs[4] = {0.f, 0.f, 0.f, 0.f}, g_al[4] = {0.f, 0.f, 0.f, 0.f};
    float g_sc[4] = {0.f, 0.f, 0.f, 0.f}, g_sh[4] = {0.f, 0.f, 0.f, 0.f};
    float* gcol = lds + 2 * kBuf + 32;  // SP: [5][H]  mean | rstd | scale | shift | alpha  (zeros without a GraphNorm)
    if constexpr (SP) {
        for (int k = tid; k < 5 * H; k += 4 * H) gcol[k] = !gn_on ? 0.f : k < 4 * H ? gs.saved[k] : gs.alpha[k - 4 * H];
        lds_barrier();  // (the first commit below reads other threads' constants)
    } else if (gn_on) {
        const float4 m4 = *reinterpret_cast<const float4*>(gs.saved + 4 * ga), r4 = *reinterpret_cast<const float4*>(gs.saved + H + 4 * ga);
        const float4 s4 = *reinterpret_cast<const float4*>(gs.saved + 2 * H + 4 * ga), h4 = *reinterpret_cast<const float4*>(gs.saved + 3 * H + 4 * ga);
        const float4 a4 = *reinterpret_cast<const float4*>(gs.alpha + 4 * ga);
        g_mu[0] = m4.x, g_mu[1] = m4.y, g_mu[2] = m4.z, g_mu[3] = m4.w;
        g_rs[0] = r4.x, g_rs[1] = r4.y, g_rs[2] = r4.z, g_rs[3] = r4.w;
        g_sc[0] = s4.x, g_sc[1] = s4.y, g_sc[2] = s4.z, g_sc[3] = s4.w;
        g_sh[0] = h4.x, g_sh[1] = h4.y, g_sh[2] = h4.z, g_sh[3] = h4.w;
        g_al[0] = a4.x, g_al[1] = a4.y, g_al[2] = a4.z, g_al[3] = a4.w;
    }
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        float* At = lds + (st & 1) * kBuf;
        float* ADD = At + kAFloats;
        float* M = ADD + 16 * RP;
        float* U = M + 16 * RP;
        float* XU = U + 16 * RP;
        const int row = row_of(st);
        const int r = row < 0 ? 0 : row;
        const float c1 = R.mk ? zr : omz, c0 = R.mk ? omz : zr;
        const float d[4] = {R.d.x, R.d.y, R.d.z, R.d.w}, t1[4] = {R.t1.x, R.t1.y, R.t1.z, R.t1.w}, t0[4] = {R.t0.x, R.t0.y, R.t0.z, R.t0.w};
        const float xv[4] = {R.x.x, R.x.y, R.x.z, R.x.w};
        float z1[4], z0[4], m[4] = {1.f, 1.f, 1.f, 1.f}, gds[4] = {1.f, 1.f, 1.f, 1.f}, u[4], xu[4];
        if constexpr (SP) {
            const float4 m4 = *reinterpret_cast<const float4*>(gcol + 4 * ga), r4 = *reinterpret_cast<const float4*>(gcol + H + 4 * ga);
            const float4 s4 = *reinterpret_cast<const float4*>(gcol + 2 * H + 4 * ga), h4 = *reinterpret_cast<const float4*>(gcol + 3 * H + 4 * ga);
            const float4 a4 = *reinterpret_cast<const float4*>(gcol + 4 * H + 4 * ga);
            g_mu[0] = m4.x, g_mu[1] = m4.y, g_mu[2] = m4.z, g_mu[3] = m4.w;
            g_rs[0] = r4.x, g_rs[1] = r4.y, g_rs[2] = r4.z, g_rs[3] = r4.w;
            g_sc[0] = s4.x, g_sc[1] = s4.y, g_sc[2] = s4.z, g_sc[3] = s4.w;
            g_sh[0] = h4.x, g_sh[1] = h4.y, g_sh[2] = h4.z, g_sh[3] = h4.w;
            g_al[0] = a4.x, g_al[1] = a4.y, g_al[2] = a4.z, g_al[3] = a4.w;
        }
        if (drop_on) drop_scales<4>(drop, r, 4 * ga, m);
        if (gdrop_on) drop_scales<4>(gdrop, r, 4 * ga, gds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            z1[k] = d[k] * c1;
            z0[k] = d[k] * c0;
            z1[k] *= act_grad(act, t1[k]), z0[k] *= act_grad(act, t0[k]);
            float uk = gds[k];
            uk *= act_grad(gs.act, fmaf(xv[k], g_sc[k], g_sh[k]));
            u[k] = uk;
            xu[k] = (xv[k] - g_al[k] * g_mu[k]) * g_rs[k] * uk;
        }
        if constexpr (SP) {
            const Split4 s1c = split4(make_float4(z1[0], z1[1], z1[2], z1[3])), s0c = split4(make_float4(z0[0], z0[1], z0[2], z0[3]));
            unsigned short* P = reinterpret_cast<unsigned short*>(At) + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = s1c.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = s1c.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = s1c.lo;
            *reinterpret_cast<uint2*>(P + H) = s0c.hi;
            *reinterpret_cast<uint2*>(P + kPlane + H) = s0c.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane + H) = s0c.lo;
        } else {
            *reinterpret_cast<float4*>(At + rs * RA + 4 * ga) = make_float4(z1[0], z1[1], z1[2], z1[3]);
            *reinterpret_cast<float4*>(At + rs * RA + H + 4 * ga) = make_float4(z0[0], z0[1], z0[2], z0[3]);
        }
        *reinterpret_cast<float4*>(ADD + rs * RP + 4 * ga) = R.ad;
        *reinterpret_cast<float4*>(M + rs * RP + 4 * ga) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(U + rs * RP + 4 * ga) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4*>(XU + rs * RP + 4 * ga) = make_float4(xu[0], xu[1], xu[2], xu[3]);
        if (ga == 0) rows_s[(st & 1) * 16 + rs] = row;
    };
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;  // this lane's column sums: float over four stages (16 values, as the 64-row tiles had it), double across
    const int c = 16 * w + j;
    auto stage = [&](int st, Raw& Rn) __attribute__((always_inline)) {
        const float* At = lds + (st & 1) * kBuf;
        const float* ADD = At + kAFloats;
        const float* M = ADD + 16 * RP;
        const float* U = M + 16 * RP;
        const float* XU = U + 16 * RP;
        float4 a4[SP ? 1 : KF4];
        if constexpr (!SP) {
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(At + j * RA + (KT / 4) * q + 4 * tt);
        }
        int rv[4];
        float ad[4], mm[4], uu[4], xx[4];
        auto epilogue_operands = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rv[r] = rows_s[(st & 1) * 16 + 4 * q + r];
                ad[r] = ADD[(4 * q + r) * RP + c];
                mm[r] = M[(4 * q + r) * RP + c];
                uu[r] = U[(4 * q + r) * RP + c];
                xx[r] = XU[(4 * q + r) * RP + c];
            }
        };
        if constexpr (!SP) epilogue_operands();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
            const unsigned short* P = reinterpret_cast<const unsigned short*>(At) + j * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b) {  // two chains (even / odd K blocks), small terms first
                uint4 af[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
#define GLASS_SMMA16(ACC, pa, pb)                                                                                     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[pa]), __builtin_bit_cast(bf16x8, bwc[b][pb]), ACC, 0, 0, 0)
                if (b & 1) {
                    GLASS_SMMA16(acc1, 1, 1); GLASS_SMMA16(acc1, 2, 0); GLASS_SMMA16(acc1, 0, 2);
                    GLASS_SMMA16(acc1, 1, 0); GLASS_SMMA16(acc1, 0, 1); GLASS_SMMA16(acc1, 0, 0);
                } else {
                    GLASS_SMMA16(acc0, 1, 1); GLASS_SMMA16(acc0, 2, 0); GLASS_SMMA16(acc0, 0, 2);
                    GLASS_SMMA16(acc0, 1, 0); GLASS_SMMA16(acc0, 0, 1); GLASS_SMMA16(acc0, 0, 0);
                }
#undef GLASS_SMMA16
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < KF4; tt += 2) {
                const float x0[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w}, y0[4] = {bw[tt].x, bw[tt].y, bw[tt].z, bw[tt].w};
                const float x1[4] = {a4[tt + 1].x, a4[tt + 1].y, a4[tt + 1].z, a4[tt + 1].w};
                const float y1[4] = {bw[tt + 1].x, bw[tt + 1].y, bw[tt + 1].z, bw[tt + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], y0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], y1[e], acc1, 0, 0, 0);
                }
            }
        }
        if (st + 1 < nst) {
            commit(st + 1, Rn);
            issue(st + 3, Rn);
        }
        if constexpr (SP) {  // the epilogue's operands behind the products and the next stage's cut (register budget)
            __builtin_amdgcn_sched_barrier(0);
            epilogue_operands();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0;
            const float v = (acc0[r] + acc1[r] + ad[r]) * mm[r];
            buf_store1(r_out, live ? (int)((rv[r] * ldo + c) * 4) : kBufOOB, v);
            const float vl = live ? v : 0.f;
            s1 = fmaf(vl, uu[r], s1);
            s2 = fmaf(vl, xx[r], s2);
        }
        if (gn_on && (st & 3) == 3) {
            d1 += (double)s1, d2 += (double)s2;
            s1 = s2 = 0.f;
        }
        if (st + 1 < nst) lds_barrier();
    };
    for (int st = 0; st < nst; st += 2) {
        stage(st, rawB);
        if (st + 1 < nst) stage(st + 1, rawA);
    }
    if (!gn_on) return;
    d1 += (double)s1, d2 += (double)s2;
    d1 += __shfl_xor(d1, 16);
    d2 += __shfl_xor(d2, 16);
    d1 += __shfl_xor(d1, 32);
    d2 += __shfl_xor(d2, 32);
    if (q != 0) return;
    if (gs.exact) {
        gn_acc_add(reinterpret_cast<long long*>(gs.partial), (int)(blockIdx.x % gs.exact), 0, c, H, d1, kAccScaleBwd);
        gn_acc_add(reinterpret_cast<long long*>(gs.partial), (int)(blockIdx.x % gs.exact), 1, c, H, d2, kAccScaleBwd);
        return;
    }
    gs.partial[((size_t)blockIdx.x * 2) * H + c] = d1;
    gs.partial[((size_t)blockIdx.x * 2 + 1) * H + c] = d2;
    for (int64_t e = (int64_t)blockIdx.x + gridDim.x; e < n_tiles_all; e += gridDim.x) {
        gs.partial[((size_t)e * 2) * H + c] = 0.0;
        gs.partial[((size_t)e * 2 + 1) * H + c] = 0.0;
    }
}
constexpr size_t trans_dgrad3_lds(int H) { return (size_t)(2 * (16 * (2 * H + 4) + 4 * 16 * (H + 4)) + 32) * sizeof(float); }
constexpr size_t trans_dgrad3_split_lds(int H) { return (size_t)(2 * ((3 * 16 * (2 * H + 8)) / 2 + 4 * 16 * (H + 4)) + 32 + 5 * H) * sizeof(float); }

constexpr size_t trans_dgrad2_lds(int H) { return (size_t)(2 * (16 * (2 * H + 4) + 4 * 16 * (H + 4)) + 64) * sizeof(float); }
constexpr size_t kTransDgrad2SplitLds = (size_t)(2 * ((3 * 16 * (2 * 64 + 8)) / 2 + 4 * 16 * (64 + 4)) + 64) * sizeof(float);  // SP at hidden 64
constexpr size_t kTransDgrad2Lds = trans_dgrad2_lds(64);

template <int H>
__global__ __launch_bounds__(4 * H) void trans_dgrad2_kernel(DgradArgs A) {
    extern __shared__ float4 lds_w[];
    trans_dgrad2_body<H>(A.dsrc, A.ldd, A.T, A.ldt, A.mask, A.zr, A.omz, A.act, A.WT, A.addend, A.ldadd, A.drop, A.rng_state, A.out,
                         A.ldo, A.N, A.gs, blockIdx.x, reinterpret_cast<float*>(lds_w));
}

template <int H, int NT, int CS, int RW>
__global__ __launch_bounds__(kWave * RW * CS) void dual_dgrad_kernel(DgradArgs A) {
    extern __shared__ float4 lds_w[];
    if (GLASS_TRANS_DGRAD_V2 && H == 64 && NT == 64) {
        trans_dgrad2_body<64>(A.dsrc, A.ldd, A.T, A.ldt, A.mask, A.zr, A.omz, A.act, A.WT, A.addend, A.ldadd, A.drop, A.rng_state,
                              A.out, A.ldo, A.N, A.gs, blockIdx.x, reinterpret_cast<float*>(lds_w));
        return;
    }
    dual_dgrad_body<H, NT, CS, RW>(A.dsrc, A.ldd, A.T, A.ldt, A.mask, A.zr, A.omz, A.act, A.WT, A.addend, A.ldadd, A.drop,
                                   A.rng_state, A.out, A.ldo, A.N, A.gs, blockIdx.x, lds_w);
}

// Fused backward launch of one Linear pair on SMALL graphs (hidden 64): the data gradient (row tiles) and the
// weight-gradient partial sums (row slabs) are independent — both only read the pair's output gradient — and at
// ppi_bp-shape each is a 12-17 us launch that leaves the chip half idle (one wave per SIMD, latency-bound).  As two
// branches of ONE launch their workgroups share the CUs (two per CU: <= 256 registers, 68 KiB LDS) and the pair costs
// about the longer of the two instead of their sum, with one launch boundary less.  (The same overlap through a
// second stream inside the captured step cost more in graph edges than it saved: DESIGN.md §5.)
template <int H, int NT, bool SP = false>
__global__ __launch_bounds__(kBlock, 2) void dual_bwd_kernel(DgradArgs A, int n_dgrad_blocks, const float* __restrict__ X,
                                                            int64_t ldx, int O, int I, int rows_per_slab, int gx, int gy,
                                                            float* __restrict__ part_w, float* __restrict__ part_b,
                                                            float* __restrict__ wg_header, WgradSynth sy) {
    extern __shared__ float4 lds_w[];
    // the weight-gradient workgroups are the long ones (phase stamps: 12.6 / 16.2 us of life against 7.8 / 11.1): they take the
    // FIRST workgroup ids, so the dispatcher starts them first; b = the id in the old order (data-gradient tiles first)
    const int n_wg_blocks = (int)gridDim.x - n_dgrad_blocks;
    const int b = (int)blockIdx.x < n_wg_blocks ? n_dgrad_blocks + (int)blockIdx.x : (int)blockIdx.x - n_wg_blocks;
    // trans pair at hidden 64: the slabs go through LDS in 16-row stages (wgrad_trans_staged2_body; plain [o][i] tiles)
    const bool staged2 = GLASS_TRANS_WGRAD_STAGED2 && NT == 64 && O == 128 && I == 64 && sy.X2 == nullptr && gy == 1;
    if (b == 0 && threadIdx.x == 0) {  // the partials below are in the plain form (mode header read by the reduce launch)
        wg_header[0] = 0.f;
        wg_header[1] = sy.zr;
        wg_header[2] = staged2 ? 1.f : 0.f;
    }
    if (b < n_dgrad_blocks) {
        D_STAMP(4, 0);
        if (GLASS_TRANS_DGRAD_V2 && H == 64 && NT == 64)
            trans_dgrad2_body<64, SP>(A.dsrc, A.ldd, A.T, A.ldt, A.mask, A.zr, A.omz, A.act, A.WT, A.addend, A.ldadd, A.drop,
                                      A.rng_state, A.out, A.ldo, A.N, A.gs, b, reinterpret_cast<float*>(lds_w));
        else
            dual_dgrad_body<H, NT, 1, 4>(A.dsrc, A.ldd, A.T, A.ldt, A.mask, A.zr, A.omz, A.act, A.WT, A.addend, A.ldadd, A.drop,
                                         A.rng_state, A.out, A.ldo, A.N, A.gs, b, lds_w);
        D_STAMP(4, 4);
        return;
    }
    D_STAMP(4, 5);
    const int t = b - n_dgrad_blocks;  // slab fastest, then input tile, then output tile (as the 3-D grid of the stand-alone launch)
    float* lds = reinterpret_cast<float*>(lds_w);
    if (staged2) {
        if constexpr (SP)
            wgrad_trans_staged2s_body(X, ldx, A.N, rows_per_slab, part_w, part_b, sy, t, lds);
        else
            wgrad_trans_staged2_body(X, ldx, A.N, rows_per_slab, part_w, part_b, sy, t, lds);
        D_STAMP(4, 6);
        return;
    }
    if (GLASS_WGRAD_STAGED && NT == 64 && rows_per_slab <= kStageRows && sy.X2 == nullptr) {
        // trans pair on a small graph (one 128 x 64 tile per slab): the whole slab through LDS, one memory round trip
        wgrad_synth_staged_body(X, ldx, A.N, rows_per_slab, part_w, part_b, sy, t, gx, lds, lds + 2 * kTile);
        return;
    }
    wgrad_partial_body<true, GLASS_FUSED_WGRAD_STAGES>(nullptr, 0, X, ldx, A.N, O, I, rows_per_slab, part_w, part_b, sy, t % gx, (t / gx) % gy,
                                t / (gx * gy), gx, gy, lds, lds + 2 * kTile);
    D_STAMP(4, 6);
}

// ---- comb pair through effective per-label weights (hidden 64) ------------------------------------------------------
// The comb pair has no activation between its two Linear layers and the label mix (reference impl/models.py:169-173), and
// the mix weights take two values per row, so
//     y[r] = w1(r) * C1[r] + w0(r) * C0[r] = [g || x_][r] . (w1 W1 + w0 W0)^T + (w1 b1 + w0 b0):
// ONE product per row with one of two effective weights.  At hidden 64 a launch lasts as long as its slowest wave (one
// wave per SIMD), so the saving only shows when EVERY wave runs the short path: all row tiles multiply the unlabeled-row
// weight W_unl = (1-z) W1 + z W0 and do not store their labeled rows; the unique labeled rows of the batch (LabRows, from
// glass_batch_labels) are gathered 16 per wave by a few extra workgroups of the same launch, which multiply W_lab and
// store them.  Every row is written exactly once (no atomics, no ordering between workgroups), the GraphNorm statistics
// of the output come from both kinds of workgroup (one partial each).  Half the MFMAs, half the weight staging, no mix
// in the epilogue; results differ from the two-product form by rounding only.
struct EffRows {
    int64_t row;      // this lane's A-operand row (-1: none)
    int32_t erow[4];  // rows of this lane's accumulator registers (-1: not stored by this workgroup)
    bool extra;
};

// Workgroup-uniform early exit for extra workgroups beyond the list: returns false.
template <int RW>
__device__ __forceinline__ bool eff_rows(EffRows& R, int block, const uint8_t* __restrict__ mask, int64_t N,
                                         const LabRows& lab, int w, int i, int q) {
    R.extra = block >= lab.n_main;
    if (!R.extra) {
        const int64_t row0 = ((int64_t)block * RW + w) * 16;
        R.row = row0 + i < N ? row0 + i : -1;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t r = row0 + 4 * q + reg;
            R.erow[reg] = (r < N && mask[r] == 0) ? (int32_t)r : -1;
        }
        return true;
    }
    const int n_lab = lab.count[0];
    const int base = ((block - lab.n_main) * RW + w) * 16;
    if ((block - lab.n_main) * RW * 16 >= n_lab) return false;
    R.row = base + i < n_lab ? lab.rows[base + i] : -1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int k = base + 4 * q + reg;
        R.erow[reg] = k < n_lab ? lab.rows[k] : -1;
    }
    return true;
}

#if GLASS_LAB
#include "../../tools/lab/comb_fwd_eff_v1.inc"  // first form of the comb forward (GLASS_COMB_FWD_V2=0): laboratory builds only
#endif

// ---- comb forward, second form (hidden 64): weights in registers, rows through LDS in four 16-row stages ---------------
// The first form gives each wave 16 rows and stages the weight image in LDS: every wave of every workgroup is in the same
// phase at the same time — a burst of loads (the launch's 27 MB in ~3 us), then prologue arithmetic on half the lanes, then
// 1.7 us of MFMAs, then a burst of stores: 12.7 us per wave for 1.7 us of matrix work (tools/dense_trace.py).  Here a wave
// owns 16 OUTPUT COLUMNS for all 64 rows of the workgroup: its slice of the effective weight is 32 registers, loaded once,
// straight from the packed image (no LDS staging, no commit barrier); the rows go through LDS in four stages of 16 rows —
// every thread loads, normalises (+dropout) and stores one float4 of each half per stage, so the prologue arithmetic is
// spread over all lanes — double-buffered, with the loads of stage s + 2 in flight across the LDS-only barrier of stage s
// (s_waitcnt lgkmcnt(0) + s_barrier: a full __syncthreads would drain them), so loads, arithmetic, MFMAs and stores of
// different stages overlap inside the workgroup.  Column statistics: a wave owns its columns, so the sums fold by lane
// shuffles and go straight to the accumulators / partials — no LDS reduction.
// Image: layout kLayoutWave16EffFwdCols (tile t = columns 16t .. 16t+15): float4 ((kc*4 + t)*4 + v)*64 + lane holds
// W_eff[16t + j][32q + 16kc + 4v ..+3], lane = j + 16q — exactly lane (j, q)'s B operand for k = 32q + 4(4kc + v) + e.

#ifndef GLASS_STAGE_INTERLEAVE
#define GLASS_STAGE_INTERLEAVE 4
#endif
// NST = 16-row stages per workgroup (4: 64-row tiles; 5: 80-row tiles, taken when that brings the launch down to one
// workgroup per CU — 280 workgroups on 256 CUs leave 24 CUs with two, whose waves share the matrix cores and finish last)
// WG = wave groups: 1 = one wave per SIMD (a stage is 16 rows); 2 = TWO waves per SIMD (hidden 64 only: 8 waves, a stage is
// 32 rows, wave group g multiplies rows 16g .. 16g + 15 of it with the same 16 weight columns) — per stage a SIMD then
// holds two independent instruction streams, so one wave's LDS-read latency, store issue and barrier wait sit under the
// other's 32 MFMAs, and the workgroup passes half as many barriers (3 stages instead of 5 for an 80-row tile).
// SP: the products in the split form (split_mma.h, see comb_fwd_eff3_kernel): rows cut once by the staging thread into three bf16
// planes, the weight slice cut once per wave; 24 MFMAs of 16 cycles per 16-row stage at hidden 64 instead of 32 of 32.
template <int H, bool DROP, int NST, int WG = 1, bool SP = false>
__global__ __launch_bounds__(4 * H * WG) void comb_fwd_eff2_kernel(const float* __restrict__ xa, int64_t lda,
                                                               const float* __restrict__ xb, int64_t ldb,
                                                               const float* __restrict__ Wimg, const float* __restrict__ bias,
                                                               const uint8_t* __restrict__ mask, float zr, float omz,
                                                               float* __restrict__ out, int64_t ldo, int64_t N,
                                                               double* __restrict__ stats, int stats_exact, GnPrologue pro,
                                                               LabRows lab) {
    static_assert(H == 64 || H == 128, "H / 16 waves x 16 columns");
    static_assert(WG == 1 || (WG == 2 && H == 64), "two wave groups: 8 waves at hidden 64");
    constexpr int THREADS = 4 * H * WG, NTL = H / 16, KF4 = (2 * H) / 16;  // threads, 16-column tiles, float4 per lane of a weight slice
    constexpr int KT = 2 * H, RS = KT + 4;  // LDS row stride (floats): + 4 keeps the 16 rows of a b128 read off one bank group
    constexpr int SR = 16 * WG, NSTG = (NST + WG - 1) / WG;  // rows per stage, stages per workgroup
    static_assert(!SP || WG == 1, "split form: one wave group");
    constexpr int RSB = KT + 8;        // SP: bf16 elements per row of a piece plane (16-B aligned, rows 4 banks apart)
    constexpr int kPlane = SR * RSB;   // bf16 elements per piece plane
    __shared__ __attribute__((aligned(16))) float tile[SP ? 1 : 2][SP ? 4 : SR * RS];
    __shared__ __attribute__((aligned(16))) unsigned short tile_s[SP ? 2 : 1][SP ? 3 * kPlane : 8];  // [buffer][piece][row][k]
    __shared__ __attribute__((aligned(16))) float gn_coef_s[2 * H];
    constexpr int ROWS = 16 * NST;
    __shared__ int rows_s[ROWS];  // row of each of the workgroup's slots: -1 none; bit 30 set: computed but not stored / counted
    __shared__ double comb_s[WG > 1 ? 2 * H : 1];  // column sums of wave group 1, handed to group 0
    D_STAMP(1, 0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = wv % NTL, g = wv / NTL;  // column tile, row group
    const int j = lane & 15, q = lane >> 4;
    // ---- prologue: EVERY load it needs is issued before the first value is used (one memory round trip; the first form
    // waited three times: weights / bias / label byte — a conditional byte load forces vmcnt(0) —, then operands and
    // accumulators, then gamma / beta / alpha, which the compiler had sunk behind the accumulator wait: tools/dense_trace.py
    // slots 0 -> 6 -> 1 read 1.7 + 2.0 us).  Order of issue = order of need: the first two stages' operand rows, the
    // GraphNorm sums, the weight slice, bias, label byte.
    const bool extra = (int)blockIdx.x >= lab.n_main;
    const int base = extra ? ((int)blockIdx.x - lab.n_main) * ROWS : 0;
    const buf_rsrc r_xa = make_rsrc(xa, N * lda * 4), r_xb = make_rsrc(xb, N * ldb * 4), r_out = make_rsrc(out, N * ldo * 4);
    const buf_rsrc r_side = make_rsrc(pro.side ? pro.side : out, pro.side ? N * pro.lds * 4 : 0);
    const buf_rsrc r_mask = make_rsrc(mask, N);
    // the two float4 this thread moves per stage: 4-column group ga of the a half (GraphNorm prologue) and of the h half of
    // row rs of the stage (one buffer resource per half: wave-uniform)
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    int my_row[NSTG];  // (-1: none)
    int n_lab = 0;
    if (!extra) {
        const int64_t r0 = (int64_t)blockIdx.x * ROWS;
#pragma unroll
        for (int st = 0; st < NSTG; ++st)
            my_row[st] = (SR * st + rs < ROWS && r0 + SR * st + rs < N) ? (int)(r0 + SR * st + rs) : -1;
    } else {
        // listed rows: every thread reads its own stage rows straight from the list (the count arrives beside them)
        const buf_rsrc r_list = make_rsrc(lab.rows, (int64_t)lab.cap * 4);
        int lr[NSTG];
#pragma unroll
        for (int st = 0; st < NSTG; ++st)
            lr[st] = buf_load1i(r_list, SR * st + rs < ROWS ? (base + SR * st + rs) * 4 : kBufOOB);
        n_lab = lab.count[0];
        if (base >= n_lab) {  // extra workgroup beyond the list: an empty partial
            if (stats && !stats_exact)
                for (int c = tid; c < 2 * H; c += THREADS) stats[(size_t)blockIdx.x * 2 * H + c] = 0.0;
            return;
        }
#pragma unroll
        for (int st = 0; st < NSTG; ++st) my_row[st] = (SR * st + rs < ROWS && base + SR * st + rs < n_lab) ? lr[st] : -1;
    }
    auto issue = [&](int st, float4 (&raw)[2]) __attribute__((always_inline)) {
        const int r = my_row[st];
        raw[0] = buf_load4(r_xa, r >= 0 ? (int)((r * lda + 4 * ga) * 4) : kBufOOB);
        raw[1] = buf_load4(r_xb, r >= 0 ? (int)((r * ldb + 4 * ga) * 4) : kBufOOB);
    };
    float4 rawA[2], rawB[2];
    issue(0, rawA);
    if (1 < NSTG) issue(1, rawB);
    GnCoefRegs CR;
    const bool fold_here = pro.saved && (WG == 1 || tid < 4 * H);  // (wave-uniform)
    if (fold_here && pro.src.acc) gn_fwd_coef_issue<H, 4 * H>(pro.src, CR);
    // this wave's slice of the effective weight: 8 float4 per lane
    const float4* img = reinterpret_cast<const float4*>(Wimg + (extra ? H * KT : 0));
    float4 bw[KF4];
#pragma unroll
    for (int tt = 0; tt < KF4; ++tt) bw[tt] = img[(((tt >> 2) * NTL + w) * 4 + (tt & 3)) * 64 + lane];
    const float bias1 = bias[16 * w + j], bias0 = bias[H + 16 * w + j];
    // row of slot `tid` (threads < ROWS) and its label byte (main tiles)
    int slot_v = -1;
    unsigned slot_mask = 0;
    if (!extra) {
        const int64_t r0 = (int64_t)blockIdx.x * ROWS;
        const bool ok = tid < ROWS && r0 + tid < N;
        slot_v = ok ? (int)(r0 + tid) : -1;
        slot_mask = __builtin_amdgcn_raw_buffer_load_b8(r_mask, ok ? (int)(r0 + tid) : kBufOOB, 0, 0);
    } else {
        const buf_rsrc r_list = make_rsrc(lab.rows, (int64_t)lab.cap * 4);
        const int v = buf_load1i(r_list, tid < ROWS ? (base + tid) * 4 : kBufOOB);
        slot_v = (tid < ROWS && base + tid < n_lab) ? v : -1;
    }
    if (fold_here && pro.src.acc) {
        gn_fwd_coef_issue_params<H>(pro.src, CR);
        glass_pin(CR.gamma);
        glass_pin(CR.beta);
        glass_pin(CR.alpha);
    }
    float bias1p = bias1, bias0p = bias0;
    glass_pin(bias1p);
    glass_pin(bias0p);
    glass_pin(slot_mask);
    D_STAMP(1, 6);
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    // ---- first use of loaded values
    const float c1 = extra ? zr : omz, c0 = extra ? omz : zr;
    const float be = c1 * bias1p + c0 * bias0p;
    if (fold_here) gn_fwd_coef_finish<H, 4 * H>(pro.src, pro.saved, N, CR, gn_coef_s);
    if (tid < ROWS) rows_s[tid] = (!extra && slot_mask != 0) ? (slot_v | (1 << 30)) : slot_v;  // labeled row of a main tile: an extra workgroup stores it
    // SP: the wave's weight slice as bf16 pieces, block b = the lane's k = (KT / 4) q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    uint4 bwc[SP ? KF4 / 2 : 1][3];
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < KF4 / 2; ++b) {
            const Split4 s0 = split4(bw[2 * b]), s1 = split4(bw[2 * b + 1]);
            bwc[b][0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
            bwc[b][1] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
            bwc[b][2] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
        }
    }
    D_STAMP(1, 1);
    lds_barrier();  // coefficients + row table
    const bool pro_on = pro.saved != nullptr;
    const bool side_on = pro.side != nullptr && !extra;  // (the row's own tile writes the normalised operand)
    // GraphNorm scale / shift of this thread's four columns (the same in every stage)
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (pro_on) {
        const float4 s4 = *reinterpret_cast<const float4*>(gn_coef_s + 4 * ga);
        const float4 h4 = *reinterpret_cast<const float4*>(gn_coef_s + H + 4 * ga);
        sc[0] = s4.x, sc[1] = s4.y, sc[2] = s4.z, sc[3] = s4.w;
        sh[0] = h4.x, sh[1] = h4.y, sh[2] = h4.z, sh[3] = h4.w;
    }
    // prep: the prologue arithmetic of one float4 — branch-free, so that it can be scheduled BETWEEN the MFMAs of the
    // previous stage (the matrix core runs a 16x16x4 for 32 cycles; a wave that issues its MFMAs back to back leaves its
    // VALU idle meanwhile, and one that runs the prologue first leaves the matrix core idle)
    auto prep = [&](int st, const float4& raw) __attribute__((always_inline)) -> float4 {
        const int r = my_row[st];
        float a[4] = {raw.x, raw.y, raw.z, raw.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_scales<4>(drop, r < 0 ? 0 : r, 4 * ga, ds);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fmaf(a[k], sc[k], sh[k]) * ds[k];
        return make_float4(a[0], a[1], a[2], a[3]);
    };
    auto commit = [&](int st, const float4& v, const float4& hraw) __attribute__((always_inline)) {
        float* T = tile[SP ? 0 : (st & 1)];
        const int r = my_row[st];
        buf_store4(r_side, (pro_on && side_on && r >= 0) ? (int)((r * pro.lds + 4 * ga) * 4) : kBufOOB, v);
        if constexpr (SP) {
            const Split4 sa = split4(v), sh4 = split4(hraw);
            unsigned short* P = tile_s[st & 1] + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sa.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sa.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sa.lo;
            *reinterpret_cast<uint2*>(P + H) = sh4.hi;
            *reinterpret_cast<uint2*>(P + kPlane + H) = sh4.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane + H) = sh4.lo;
        } else {
            *reinterpret_cast<float4*>(T + rs * RS + 4 * ga) = v;
            *reinterpret_cast<float4*>(T + rs * RS + H + 4 * ga) = hraw;
        }
    };
    float ssum = 0.f, ssq = 0.f;  // this lane's column 16w + j over the rows 4q + r of every stage
    commit(0, prep(0, rawA[0]), rawA[1]);
    if (2 < NSTG) issue(2, rawA);
    lds_barrier();
#pragma unroll
    for (int st = 0; st < NSTG; ++st) {
        if (st == 1) D_STAMP(1, 5);
        const bool live_grp = SR * st + 16 * g < ROWS;  // (wave-uniform: the last stage of an odd NST has one row group only)
        const float* T = tile[SP ? 0 : (st & 1)] + (16 * g + j) * RS + (KT / 4) * q;
        float4 a4[SP ? 1 : KF4];
        uint4 af[SP ? KF4 / 2 : 1][3];
        if constexpr (SP) {
            // lane (j, q): row j of the stage, k = (KT / 4) q + 8 b .. + 7 of block b: one 16-byte read per piece
            const unsigned short* P = tile_s[st & 1] + (16 * g + j) * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[b][pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
        } else {
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(T + 4 * tt);
        }
        int rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = live_grp ? rows_s[(SR * st + 16 * g + 4 * q + r) < ROWS ? SR * st + 16 * g + 4 * q + r : 0] : -1;
        float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st + 1 < NSTG) vn = prep(st + 1, (st & 1) ? rawA[0] : rawB[0]);  // the NEXT stage's rows (even stages come in rawA)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains hide the dependent-MFMA latency
        if constexpr (SP) {
#define GLASS_SMMA16(ACC, B, pa, pb)                                                                                  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[B][pa]), __builtin_bit_cast(bf16x8, bwc[B][pb]), ACC, 0, 0, 0)
#pragma unroll
            for (int b = 0; b < KF4 / 2; b += 2) {  // two K blocks side by side (two chains), small terms first
                GLASS_SMMA16(acc0, b, 1, 1); GLASS_SMMA16(acc1, b + 1, 1, 1);
                GLASS_SMMA16(acc0, b, 2, 0); GLASS_SMMA16(acc1, b + 1, 2, 0);
                GLASS_SMMA16(acc0, b, 0, 2); GLASS_SMMA16(acc1, b + 1, 0, 2);
                GLASS_SMMA16(acc0, b, 1, 0); GLASS_SMMA16(acc1, b + 1, 1, 0);
                GLASS_SMMA16(acc0, b, 0, 1); GLASS_SMMA16(acc1, b + 1, 0, 1);
                GLASS_SMMA16(acc0, b, 0, 0); GLASS_SMMA16(acc1, b + 1, 0, 0);
            }
#undef GLASS_SMMA16
        } else if (WG == 1 || live_grp) {
#pragma unroll
            for (int tt = 0; tt < KF4; tt += 2) {
                const float x0[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w}, y0[4] = {bw[tt].x, bw[tt].y, bw[tt].z, bw[tt].w};
                const float x1[4] = {a4[tt + 1].x, a4[tt + 1].y, a4[tt + 1].z, a4[tt + 1].w};
                const float y1[4] = {bw[tt + 1].x, bw[tt + 1].y, bw[tt + 1].z, bw[tt + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], y0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], y1[e], acc1, 0, 0, 0);
                }
            }
        }
        if (st == 1) D_STAMP(1, 2);
        if (st + 1 < NSTG) {
            commit(st + 1, vn, (st & 1) ? rawA[1] : rawB[1]);
            if (st + 3 < NSTG) {
                if (st & 1) issue(st + 3, rawA); else issue(st + 3, rawB);
            }
        }
        // acc[r] = row slot SR st + 16 g + 4q + r, column 16w + j
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0 && !(rv[r] >> 30);
            const float o = acc0[r] + acc1[r] + be;
            buf_store1(r_out, live ? (int)((rv[r] * ldo + 16 * w + j) * 4) : kBufOOB, o);
            ssum += live ? o : 0.f;
            ssq += live ? o * o : 0.f;
        }
#if GLASS_STAGE_INTERLEAVE
        // one MFMA, then a few of the next stage's VALU instructions, 32 times
        if (WG == 1) {
#pragma unroll
            for (int i = 0; i < (SP ? 6 * (KF4 / 2) : 32); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SP ? GLASS_STAGE_INTERLEAVE + 2 : GLASS_STAGE_INTERLEAVE, 0);
            }
        }
#endif
        if (st + 1 < NSTG) lds_barrier();  // the next stage's rows are in LDS; every wave is done reading this stage's buffer
    }
    D_STAMP(1, 3);
    if (stats == nullptr) return;
    double s = (double)ssum, q2 = (double)ssq;
    s += __shfl_xor(s, 16);
    q2 += __shfl_xor(q2, 16);
    s += __shfl_xor(s, 32);
    q2 += __shfl_xor(q2, 32);
    if (WG > 1) {  // one add per column and workgroup, as with one wave group: group 1 hands its sums over
        if (g == 1 && q == 0) {
            comb_s[16 * w + j] = s;
            comb_s[H + 16 * w + j] = q2;
        }
        lds_barrier();
        if (g == 1) return;
        s += comb_s[16 * w + j];
        q2 += comb_s[H + 16 * w + j];
    }
    if (q == 0) {
        const int c = 16 * w + j;
        if (stats_exact) {
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 0, c, H, s, kAccScaleFwd);
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 1, c, H, q2, kAccScaleFwd);
        } else {
            stats[((size_t)blockIdx.x * 2) * H + c] = s;
            stats[((size_t)blockIdx.x * 2 + 1) * H + c] = q2;
        }
    }
    D_STAMP(1, 4);
}

#if GLASS_LAB
#include "../../tools/lab/comb_fwd_eff2p.inc"  // software-pipelined variant of the second form (measured neutral/negative): laboratory builds only
#endif

// ---- comb forward, third form: the second form with a RUN-TIME number of stages per workgroup ("tall" row tiles) ---------
// The second form fixes 64 or 80 rows per workgroup.  Beyond 80 x 256 rows a launch then runs in several ROUNDS of
// workgroups, each paying its own prologue (one memory round trip + the GraphNorm fold, ~3 us) and a drained pipeline: at
// em_user-shape (N = 50 000, hidden 128: 782 workgroups of eight waves, one per CU) the launch took 52 us for 3.4 us of
// matrix work per workgroup.  Here a main workgroup takes rows_main rows (a multiple of the stage height, chosen by the
// host so that the grid is ONE round: ~N / 256), walks them in a run-time loop unrolled by two (the two register sets of
// the double-buffered loads), and pays the prologue once.  The label byte of a row travels with its operand loads and
// reaches the epilogue through LDS next to the row id (no per-workgroup row table); the extra workgroups (listed labeled
// rows, effective labeled weight) keep a small row table and at most 80 rows.
// SP (hidden 128): the product in the split form of the tiled family (split_mma.h: every operand value cut into three bf16 pieces,
// six partial products per fp32 product on v_mfma_f32_16x16x32_bf16) — this kernel's matrix pipes were 0.45 busy on the f32-input
// MFMA (profiles/r05_step_pmc_em_user_summary.csv), 18 us of a 41 us launch; the split form needs 6/16 of those cycles.  The
// rows are cut ONCE, by the staging thread, and lie in LDS as three bf16 planes; the weight slice of a wave is cut once per
// workgroup from the SAME fp32 image (lane (j, q) owns k = 64 q .. 64 q + 63 of its column either way: block b of 32 k takes
// its k = 64 q + 8 b .. + 7).  Results as close to fp64 as the f32-input form (tests/test_gpu_kernels.py).
template <int H, bool DROP, int WG, bool SP = false>
__global__ __launch_bounds__(4 * H * WG) void comb_fwd_eff3_kernel(const float* __restrict__ xa, int64_t lda,
                                                               const float* __restrict__ xb, int64_t ldb,
                                                               const float* __restrict__ Wimg, const float* __restrict__ bias,
                                                               const uint8_t* __restrict__ mask, float zr, float omz,
                                                               float* __restrict__ out, int64_t ldo, int64_t N,
                                                               double* __restrict__ stats, int stats_exact, GnPrologue pro,
                                                               LabRows lab, int rows_main, int rows_extra) {
    static_assert(H == 64 || H == 128, "H / 16 waves x 16 columns");
    static_assert(WG == 1 || (WG == 2 && H == 64), "two wave groups: 8 waves at hidden 64");
    constexpr int THREADS = 4 * H * WG, NTL = H / 16, KF4 = (2 * H) / 16;
    constexpr int KT = 2 * H, RS = KT + 4;
    constexpr int SR = 16 * WG;
    static_assert(!SP || WG == 1, "split form: one wave group");
    constexpr int RSB = KT + 8;        // SP: bf16 elements per row of a piece plane (528 B: 16-B aligned, rows 4 banks apart)
    constexpr int kPlane = SR * RSB;   // bf16 elements per piece plane
    __shared__ __attribute__((aligned(16))) float tile[SP ? 1 : 2][SP ? 4 : SR * RS];
    __shared__ __attribute__((aligned(16))) unsigned short tile_s[SP ? 2 : 1][SP ? 3 * kPlane : 8];  // [buffer][piece][row][k]
    __shared__ __attribute__((aligned(16))) float gn_coef_s[2 * H];
    __shared__ int rowflag_s[2][SR];   // per stage buffer: row of each slot (-1 none; bit 30: computed but not stored / counted)
    __shared__ int xrows_s[80];        // extra workgroups: their listed rows
    __shared__ double comb_s[WG > 1 ? 2 * H : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = wv % NTL, g = wv / NTL;
    const int j = lane & 15, q = lane >> 4;
    const bool extra = (int)blockIdx.x >= lab.n_main;
    const int rows_wg = extra ? rows_extra : rows_main;
    const int nst = (rows_wg + SR - 1) / SR;
    const int base = extra ? ((int)blockIdx.x - lab.n_main) * rows_extra : 0;
    const int64_t r0 = (int64_t)blockIdx.x * rows_main;  // (main workgroups)
    const buf_rsrc r_xa = make_rsrc(xa, N * lda * 4), r_xb = make_rsrc(xb, N * ldb * 4), r_out = make_rsrc(out, N * ldo * 4);
    const buf_rsrc r_side = make_rsrc(pro.side ? pro.side : out, pro.side ? N * pro.lds * 4 : 0);
    const buf_rsrc r_mask = make_rsrc(mask, N);
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    int n_lab = 0;
    if (extra) {  // the listed rows through LDS (a dependent pair of round trips, on a few short workgroups)
        const buf_rsrc r_list = make_rsrc(lab.rows, (int64_t)lab.cap * 4);
        const int v = buf_load1i(r_list, tid < rows_extra ? (base + tid) * 4 : kBufOOB);
        n_lab = lab.count[0];
        if (base >= n_lab) {
            if (stats && !stats_exact)
                for (int c = tid; c < 2 * H; c += THREADS) stats[(size_t)blockIdx.x * 2 * H + c] = 0.0;
            return;
        }
        if (tid < 80) xrows_s[tid] = (tid < rows_extra && base + tid < n_lab) ? v : -1;
        lds_barrier();
    }
    struct Raw {
        float4 a, h;
        unsigned mk;
        int row;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int slot = SR * st + rs;
        int r = -1;
        if (st < nst && slot < rows_wg) r = extra ? xrows_s[slot < 80 ? slot : 0] : (r0 + slot < N ? (int)(r0 + slot) : -1);
        R.row = r;
        R.a = buf_load4(r_xa, r >= 0 ? (int)((r * lda + 4 * ga) * 4) : kBufOOB);
        R.h = buf_load4(r_xb, r >= 0 ? (int)((r * ldb + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_mask, (r >= 0 && !extra) ? r : kBufOOB, 0, 0);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    GnCoefRegs CR;
    const bool fold_here = pro.saved && (WG == 1 || tid < 4 * H);
    if (fold_here && pro.src.acc) gn_fwd_coef_issue<H, 4 * H>(pro.src, CR);
    const float4* img = reinterpret_cast<const float4*>(Wimg + (extra ? H * KT : 0));
    float4 bw[KF4];
#pragma unroll
    for (int tt = 0; tt < KF4; ++tt) bw[tt] = img[(((tt >> 2) * NTL + w) * 4 + (tt & 3)) * 64 + lane];
    float bias1 = bias[16 * w + j], bias0 = bias[H + 16 * w + j];
    if (fold_here && pro.src.acc) {
        gn_fwd_coef_issue_params<H>(pro.src, CR);
        glass_pin(CR.gamma);
        glass_pin(CR.beta);
        glass_pin(CR.alpha);
    }
    glass_pin(bias1);
    glass_pin(bias0);
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    const float c1 = extra ? zr : omz, c0 = extra ? omz : zr;
    const float be = c1 * bias1 + c0 * bias0;
    if (fold_here) gn_fwd_coef_finish<H, 4 * H>(pro.src, pro.saved, N, CR, gn_coef_s);
    // SP: the wave's weight slice as bf16 pieces, block b = the lane's k = 64 q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    uint4 bwc[SP ? KF4 / 2 : 1][3];
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < KF4 / 2; ++b) {
            const Split4 s0 = split4(bw[2 * b]), s1 = split4(bw[2 * b + 1]);
            bwc[b][0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
            bwc[b][1] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
            bwc[b][2] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
        }
    }
    lds_barrier();  // coefficients
    const bool pro_on = pro.saved != nullptr;
    const bool side_on = pro.side != nullptr && !extra;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (pro_on) {
        const float4 s4 = *reinterpret_cast<const float4*>(gn_coef_s + 4 * ga);
        const float4 h4 = *reinterpret_cast<const float4*>(gn_coef_s + H + 4 * ga);
        sc[0] = s4.x, sc[1] = s4.y, sc[2] = s4.z, sc[3] = s4.w;
        sh[0] = h4.x, sh[1] = h4.y, sh[2] = h4.z, sh[3] = h4.w;
    }
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        float* T = tile[SP ? 0 : (st & 1)];
        const int r = R.row;
        float a[4] = {R.a.x, R.a.y, R.a.z, R.a.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_scales<4>(drop, r < 0 ? 0 : r, 4 * ga, ds);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fmaf(a[k], sc[k], sh[k]) * ds[k];
        const float4 v = make_float4(a[0], a[1], a[2], a[3]);
        buf_store4(r_side, (pro_on && side_on && r >= 0) ? (int)((r * pro.lds + 4 * ga) * 4) : kBufOOB, v);
        if constexpr (SP) {
            const Split4 sa = split4(v), sh4 = split4(R.h);
            unsigned short* P = tile_s[st & 1] + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sa.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sa.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sa.lo;
            *reinterpret_cast<uint2*>(P + H) = sh4.hi;
            *reinterpret_cast<uint2*>(P + kPlane + H) = sh4.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane + H) = sh4.lo;
        } else {
            *reinterpret_cast<float4*>(T + rs * RS + 4 * ga) = v;
            *reinterpret_cast<float4*>(T + rs * RS + H + 4 * ga) = R.h;
        }
        if (ga == 0) rowflag_s[st & 1][rs] = (r >= 0 && R.mk != 0) ? (r | (1 << 30)) : r;  // labeled row of a main tile: an extra workgroup stores it
    };
    float ssum = 0.f, ssq = 0.f;
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    // one stage: the MFMAs of stage st (buffer st & 1), then stage st + 1's rows (held in Rn) go to the other buffer and the
    // loads of stage st + 3 are issued into Rn
    auto stage = [&](int st, Raw& Rn) __attribute__((always_inline)) {
        int rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = rowflag_s[st & 1][16 * g + 4 * q + r];
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
            // lane (j, q): row j of the stage, k = 64 q + 8 b .. + 7 of block b: one 16-byte read per piece
            const unsigned short* P = tile_s[st & 1] + (16 * g + j) * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b) {
                uint4 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const uint4*>(P + p * kPlane + 8 * b);
#define GLASS_SMMA16(ACC, pa, pb)                                                                                     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[pa]), __builtin_bit_cast(bf16x8, bwc[b][pb]), ACC, 0, 0, 0)
                if (b & 1) {
                    GLASS_SMMA16(acc1, 1, 1); GLASS_SMMA16(acc1, 2, 0); GLASS_SMMA16(acc1, 0, 2);
                    GLASS_SMMA16(acc1, 1, 0); GLASS_SMMA16(acc1, 0, 1); GLASS_SMMA16(acc1, 0, 0);
                } else {
                    GLASS_SMMA16(acc0, 1, 1); GLASS_SMMA16(acc0, 2, 0); GLASS_SMMA16(acc0, 0, 2);
                    GLASS_SMMA16(acc0, 1, 0); GLASS_SMMA16(acc0, 0, 1); GLASS_SMMA16(acc0, 0, 0);
                }
#undef GLASS_SMMA16
            }
        } else {
            const float* T = tile[SP ? 0 : (st & 1)] + (16 * g + j) * RS + (KT / 4) * q;
            float4 a4[KF4];
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(T + 4 * tt);
#pragma unroll
            for (int tt = 0; tt < KF4; tt += 2) {
                const float x0[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w}, y0[4] = {bw[tt].x, bw[tt].y, bw[tt].z, bw[tt].w};
                const float x1[4] = {a4[tt + 1].x, a4[tt + 1].y, a4[tt + 1].z, a4[tt + 1].w};
                const float y1[4] = {bw[tt + 1].x, bw[tt + 1].y, bw[tt + 1].z, bw[tt + 1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], y0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], y1[e], acc1, 0, 0, 0);
                }
            }
        }
        if (st + 1 < nst) {
            commit(st + 1, Rn);
            issue(st + 3, Rn);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0 && !(rv[r] >> 30);
            const float o = acc0[r] + acc1[r] + be;
            buf_store1(r_out, live ? (int)((rv[r] * ldo + 16 * w + j) * 4) : kBufOOB, o);
            ssum += live ? o : 0.f;
            ssq += live ? o * o : 0.f;
        }
        if (st + 1 < nst) lds_barrier();
    };
    for (int st = 0; st < nst; st += 2) {
        stage(st, rawB);
        if (st + 1 < nst) stage(st + 1, rawA);
    }
    if (stats == nullptr) return;
    double s = (double)ssum, q2 = (double)ssq;
    s += __shfl_xor(s, 16);
    q2 += __shfl_xor(q2, 16);
    s += __shfl_xor(s, 32);
    q2 += __shfl_xor(q2, 32);
    if (WG > 1) {
        if (g == 1 && q == 0) {
            comb_s[16 * w + j] = s;
            comb_s[H + 16 * w + j] = q2;
        }
        lds_barrier();
        if (g == 1) return;
        s += comb_s[16 * w + j];
        q2 += comb_s[H + 16 * w + j];
    }
    if (q == 0) {
        const int c = 16 * w + j;
        if (stats_exact) {
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 0, c, H, s, kAccScaleFwd);
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 1, c, H, q2, kAccScaleFwd);
        } else {
            stats[((size_t)blockIdx.x * 2) * H + c] = s;
            stats[((size_t)blockIdx.x * 2 + 1) * H + c] = q2;
        }
    }
}

// ---- trans forward in the same form (hidden 64): out = mix(act(xa W1^T + b1), act(xa W0^T + b0)), T = the two pre-activations
// A wave owns columns 16w .. 16w+15 of BOTH halves (the label mix needs f1 and f0 of a column in one lane): 2 x 4 float4 of
// weights per lane (K = 64), 32 MFMAs per 16-row stage; one float4 of the operand per thread and stage.  Image: layout
// kLayoutWave16Cols.  xa_index: the stage's rows are gathered from the embedding table (layer 0).
// SP: the products in the split form of the tiled family (split_mma.h; see comb_fwd_eff3_kernel): the staging thread cuts its
// float4 once into three bf16 pieces (planes [piece][row][k], rows 72 bf16 apart: a fragment read of 16 lanes covers all 64
// banks once), a wave cuts its two weight slices once; 24 MFMAs of 16 cycles per 16-row stage instead of 32 of 32.
template <int H, int NST, int WG = 1, bool SP = false>
__global__ __launch_bounds__(kBlock * WG) void trans_fwd2_kernel(const float* __restrict__ xa, int64_t lda, int64_t xa_rows,
                                                            const float* __restrict__ Wimg, const float* __restrict__ bias,
                                                            const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                            float* __restrict__ T, int64_t ldt, float* __restrict__ out,
                                                            int64_t ldo, int64_t N, double* __restrict__ stats, int stats_exact,
                                                            GnPrologue pro, const int64_t* __restrict__ xa_index) {
    static_assert(H == 64, "four waves x 16 columns");
    static_assert(WG == 1 || WG == 2, "one or two waves per SIMD (comb_fwd_eff2_kernel)");
    constexpr int RS = H + 4;  // LDS row stride (floats)
    constexpr int SR = 16 * WG, NSTG = (NST + WG - 1) / WG;  // rows per stage, stages per workgroup
    constexpr int RSB = H + 8;         // SP: bf16 elements per row of a piece plane (144 B)
    constexpr int kPlane = SR * RSB;   // bf16 elements per piece plane
    __shared__ __attribute__((aligned(16))) float tile[SP ? 1 : 2][SP ? 4 : SR * RS];
    __shared__ __attribute__((aligned(16))) unsigned short tile_s[SP ? 2 : 1][SP ? 3 * kPlane : 8];  // [buffer][piece][row][k]
    __shared__ __attribute__((aligned(16))) float gn_coef_s[2 * H];
    constexpr int ROWS = 16 * NST;
    __shared__ int rows_s[ROWS];  // row of each slot: -1 none; bit 30: labeled row (mix weights swapped)
    __shared__ double comb_s[WG > 1 ? 2 * H : 1];  // column sums of wave group 1, handed to group 0
    D_STAMP(2, 0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = wv & 3, g = wv >> 2;  // column tile, row group
    const int j = lane & 15, q = lane >> 4;
    // ---- prologue: every load issued before the first value is used, in order of need (see comb_fwd_eff2_kernel) ----
    const buf_rsrc r_xa = make_rsrc(xa, xa_rows * lda * 4), r_out = make_rsrc(out, N * ldo * 4);
    const buf_rsrc r_T = make_rsrc(T ? T : out, T ? N * ldt * 4 : 0);
    const buf_rsrc r_side = make_rsrc(pro.side ? pro.side : out, pro.side ? N * pro.lds * 4 : 0);
    const buf_rsrc r_mask = make_rsrc(mask, N);
    const int rs = tid >> 4, ga = tid & 15;
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    int my_row[NSTG], my_src[NSTG];  // output row of this thread's float4 per stage (-1 none) and the operand row it comes from
#pragma unroll
    for (int st = 0; st < NSTG; ++st)
        my_row[st] = (SR * st + rs < ROWS && r0 + SR * st + rs < N) ? (int)(r0 + SR * st + rs) : -1;
    if (xa_index) {  // layer 0: the operand rows are gathered from the embedding table (index -> row: a dependent pair)
        int64_t idx[NSTG];
#pragma unroll
        for (int st = 0; st < NSTG; ++st) idx[st] = my_row[st] >= 0 ? xa_index[my_row[st]] : 0;
#pragma unroll
        for (int st = 0; st < NSTG; ++st) my_src[st] = (int)(idx[st] < 0 ? 0 : (idx[st] >= xa_rows ? xa_rows - 1 : idx[st]));
    } else {
#pragma unroll
        for (int st = 0; st < NSTG; ++st) my_src[st] = my_row[st];
    }
    auto issue = [&](int st) __attribute__((always_inline)) -> float4 {
        return buf_load4(r_xa, my_row[st] >= 0 ? (int)((my_src[st] * lda + 4 * ga) * 4) : kBufOOB);
    };
    float4 rawA = issue(0), rawB = make_float4(0.f, 0.f, 0.f, 0.f);
    if (1 < NSTG) rawB = issue(1);
    GnCoefRegs CR;
    const bool fold_here = pro.saved && (WG == 1 || tid < kBlock);  // (wave-uniform)
    if (fold_here && pro.src.acc) gn_fwd_coef_issue<H, kBlock>(pro.src, CR);
    const float4* img = reinterpret_cast<const float4*>(Wimg);
    float4 bw1[4], bw0[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bw1[v] = img[(w * 4 + v) * 64 + lane];
        bw0[v] = img[((4 + w) * 4 + v) * 64 + lane];
    }
    float b1 = bias[16 * w + j], b0 = bias[H + 16 * w + j];
    const bool slot_ok = tid < ROWS && r0 + tid < N;
    const int slot_v = slot_ok ? (int)(r0 + tid) : -1;
    unsigned slot_mask = __builtin_amdgcn_raw_buffer_load_b8(r_mask, slot_ok ? (int)(r0 + tid) : kBufOOB, 0, 0);
    if (fold_here && pro.src.acc) {
        gn_fwd_coef_issue_params<H>(pro.src, CR);
        glass_pin(CR.gamma);
        glass_pin(CR.beta);
        glass_pin(CR.alpha);
    }
    glass_pin(b1);
    glass_pin(b0);
    glass_pin(slot_mask);
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    if (fold_here) gn_fwd_coef_finish<H, kBlock>(pro.src, pro.saved, N, CR, gn_coef_s);
    if (tid < ROWS) rows_s[tid] = slot_mask != 0 ? (slot_v | (1 << 30)) : slot_v;
    // SP: the wave's two weight slices as bf16 pieces, block b = the lane's k = 16 q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    uint4 bwc1[SP ? 2 : 1][3], bwc0[SP ? 2 : 1][3];
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const Split4 s0 = split4(bw1[2 * b]), s1 = split4(bw1[2 * b + 1]);
            bwc1[b][0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
            bwc1[b][1] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
            bwc1[b][2] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
            const Split4 u0 = split4(bw0[2 * b]), u1 = split4(bw0[2 * b + 1]);
            bwc0[b][0] = make_uint4(u0.hi.x, u0.hi.y, u1.hi.x, u1.hi.y);
            bwc0[b][1] = make_uint4(u0.mid.x, u0.mid.y, u1.mid.x, u1.mid.y);
            bwc0[b][2] = make_uint4(u0.lo.x, u0.lo.y, u1.lo.x, u1.lo.y);
        }
    }
    D_STAMP(2, 1);
    lds_barrier();  // coefficients + row table
    const bool pro_on = pro.saved != nullptr;
    auto stage_store = [&](int st, const float4& raw) __attribute__((always_inline)) {
        const int r = my_row[st];
        float a[4] = {raw.x, raw.y, raw.z, raw.w};
        const bool pl = pro_on && r >= 0;
        if (pl) {
            const float4 s4 = *reinterpret_cast<const float4*>(gn_coef_s + 4 * ga);
            const float4 h4 = *reinterpret_cast<const float4*>(gn_coef_s + H + 4 * ga);
            const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {h4.x, h4.y, h4.z, h4.w};
            float ds[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop.p > 0.f) drop_scales<4>(drop, r, 4 * ga, ds);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float h = fmaf(a[k], sc[k], sh[k]);
                h = act_fast(pro.act, h);
                a[k] = h * ds[k];
            }
        }
        const float4 v = make_float4(a[0], a[1], a[2], a[3]);
        buf_store4(r_side, (pl && pro.side) ? (int)((r * pro.lds + 4 * ga) * 4) : kBufOOB, v);
        if constexpr (SP) {
            const Split4 sa = split4(v);
            unsigned short* P = tile_s[st & 1] + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sa.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sa.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sa.lo;
        } else {
            *reinterpret_cast<float4*>(tile[SP ? 0 : (st & 1)] + rs * RS + 4 * ga) = v;
        }
    };
    float ssum = 0.f, ssq = 0.f;
    stage_store(0, rawA);
    if (2 < NSTG) rawA = issue(2);
    lds_barrier();
#pragma unroll
    for (int st = 0; st < NSTG; ++st) {
        const bool live_grp = SR * st + 16 * g < ROWS;  // (wave-uniform: the last stage of an odd NST has one row group only)
        int rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = live_grp ? rows_s[(SR * st + 16 * g + 4 * q + r) < ROWS ? SR * st + 16 * g + 4 * q + r : 0] : -1;
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc0 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
            if (WG == 1 || live_grp) {
                // lane (j, q): row j of the row group, k = 16 q + 8 b .. + 7 of block b: one 16-byte read per piece
                const unsigned short* P = tile_s[st & 1] + (16 * g + j) * RSB + 16 * q;
                uint4 af[2][3];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) af[b][pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
#define GLASS_SMMA16(ACC, BW, pa, pb)                                                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[b][pa]), __builtin_bit_cast(bf16x8, BW[b][pb]), ACC, 0, 0, 0)
#pragma unroll
                for (int b = 0; b < 2; ++b) {  // small terms first; the two halves' chains interleave
                    GLASS_SMMA16(acc1, bwc1, 1, 1); GLASS_SMMA16(acc0, bwc0, 1, 1);
                    GLASS_SMMA16(acc1, bwc1, 2, 0); GLASS_SMMA16(acc0, bwc0, 2, 0);
                    GLASS_SMMA16(acc1, bwc1, 0, 2); GLASS_SMMA16(acc0, bwc0, 0, 2);
                    GLASS_SMMA16(acc1, bwc1, 1, 0); GLASS_SMMA16(acc0, bwc0, 1, 0);
                    GLASS_SMMA16(acc1, bwc1, 0, 1); GLASS_SMMA16(acc0, bwc0, 0, 1);
                    GLASS_SMMA16(acc1, bwc1, 0, 0); GLASS_SMMA16(acc0, bwc0, 0, 0);
                }
#undef GLASS_SMMA16
            }
        } else if (WG == 1 || live_grp) {
            const float* A = tile[SP ? 0 : (st & 1)] + (16 * g + j) * RS + 16 * q;
            float4 a4[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) a4[v] = *reinterpret_cast<const float4*>(A + 4 * v);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float x[4] = {a4[v].x, a4[v].y, a4[v].z, a4[v].w};
                const float y1[4] = {bw1[v].x, bw1[v].y, bw1[v].z, bw1[v].w}, y0[4] = {bw0[v].x, bw0[v].y, bw0[v].z, bw0[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y1[e], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y0[e], acc0, 0, 0, 0);
                }
            }
        }
        // the next stage's rows -> the other buffer (its readers passed the barrier at the end of the previous iteration);
        // behind the MFMAs in program order, so the prologue arithmetic overlaps their execution
        if (st + 1 < NSTG) {
            if (st & 1) {
                stage_store(st + 1, rawA);
                if (st + 3 < NSTG) rawA = issue(st + 3);
            } else {
                stage_store(st + 1, rawB);
                if (st + 3 < NSTG) rawB = issue(st + 3);
            }
        }
        const int c = 16 * w + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0;
            const int row = rv[r] & ((1 << 30) - 1);
            const bool lab = (rv[r] >> 30) & 1;
            const float w1 = lab ? zr : omz, w0 = lab ? omz : zr;
            const float v1 = acc1[r] + b1, v0 = acc0[r] + b0;
            buf_store1(r_T, (live && T) ? (int)((row * ldt + c) * 4) : kBufOOB, v1);
            buf_store1(r_T, (live && T) ? (int)((row * ldt + H + c) * 4) : kBufOOB, v0);
            float a1 = v1, a0 = v0;
            a1 = act_fast(act, a1), a0 = act_fast(act, a0);
            const float o = w1 * a1 + w0 * a0;
            buf_store1(r_out, live ? (int)((row * ldo + c) * 4) : kBufOOB, o);
            ssum += live ? o : 0.f;
            ssq += live ? o * o : 0.f;
        }
        if (st + 1 < NSTG) lds_barrier();
    }
    D_STAMP(2, 3);
    if (stats == nullptr) return;
    double s = (double)ssum, q2 = (double)ssq;
    s += __shfl_xor(s, 16);
    q2 += __shfl_xor(q2, 16);
    s += __shfl_xor(s, 32);
    q2 += __shfl_xor(q2, 32);
    if (WG > 1) {  // one add per column and workgroup: group 1 hands its sums over
        if (g == 1 && q == 0) {
            comb_s[16 * w + j] = s;
            comb_s[H + 16 * w + j] = q2;
        }
        lds_barrier();
        if (g == 1) return;
        s += comb_s[16 * w + j];
        q2 += comb_s[H + 16 * w + j];
    }
    if (q == 0) {
        const int c = 16 * w + j;
        if (stats_exact) {
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 0, c, H, s, kAccScaleFwd);
            gn_acc_add(reinterpret_cast<long long*>(stats), blockIdx.x % stats_exact, 1, c, H, q2, kAccScaleFwd);
        } else {
            stats[((size_t)blockIdx.x * 2) * H + c] = s;
            stats[((size_t)blockIdx.x * 2 + 1) * H + c] = q2;
        }
    }
}

// ---- trans forward, stage-run form (hidden 128; round 6) ------------------------------------------------------------------
// The LDS-tiled kernel served this pair at hidden 128 (tiled_fwd_kernel<128, false, 128>: 391 tiles of 128 rows at em_user-shape,
// every tile re-streaming the 196 KiB cut weight image through LDS, the tiles in phase — K loops with HBM idle, then one
// chip-wide burst of 77 MB of T and m with the matrix cores idle: 2.2 + 20.6 + 14.3 us by phase stamps, DESIGN 7 R4-a').  This is
// comb_fwd_eff3_kernel's form for the trans pair: eight waves, wave w owns columns 16 w .. 16 w + 15 of BOTH halves (the label
// mix needs f1 and f0 of a column in one lane) and keeps its two weight slices in registers for the whole run (K = 128: 2 x 8
// float4, cut once into 2 x 12 uint4 in the split form); a workgroup walks rows_main rows (one round of the chip) in 16-row
// stages, double-buffered loads two stages ahead, so loads, products and stores of different stages overlap all along the
// launch.  Image: layout kLayoutWave16Cols ([256 outputs][128]).  xa_index: the rows are gathered from the embedding table
// (layer 0; the index of a stage is requested two stages before its rows).  Statistics (partials form): workgroup b writes
// entry b and zeroes entries b + grid, ... (one entry per 64 rows in the ABI; rows_main >= 64).
template <int H, bool SP>
__global__ __launch_bounds__(4 * H) void trans_fwd3_kernel(const float* __restrict__ xa, int64_t lda, int64_t xa_rows,
                                                           const float* __restrict__ Wimg, const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ mask, float zr, float omz, int act,
                                                           float* __restrict__ T, int64_t ldt, float* __restrict__ out,
                                                           int64_t ldo, int64_t N, double* __restrict__ stats, GnPrologue pro,
                                                           const int64_t* __restrict__ xa_index, int rows_main) {
    static_assert(H == 128, "eight waves x 16 columns of both halves");
    constexpr int THREADS = 4 * H, NTL = H / 16, KF4 = H / 16;  // float4 of one weight slice per lane (k = 32 q + 4 tt + e)
    constexpr int KT = H, RS = KT + 4;
    constexpr int RSB = KT + 8;        // SP: bf16 elements per row of a piece plane (272 B: 16 lanes' fragment reads cover the 64 banks once)
    constexpr int kPlane = 16 * RSB;
    __shared__ __attribute__((aligned(16))) float tile[SP ? 1 : 2][SP ? 4 : 16 * RS];
    __shared__ __attribute__((aligned(16))) unsigned short tile_s[SP ? 2 : 1][SP ? 3 * kPlane : 8];  // [buffer][piece][row][k]
    __shared__ __attribute__((aligned(16))) float gn_coef_s[2 * H];
    __shared__ int rowflag_s[2][16];   // per stage buffer: row of each slot (-1 none; bit 30: labeled row)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_main;
    const int n_rows = (int)(r0 + rows_main <= N ? rows_main : N - r0);
    const int nst = (n_rows + 15) / 16;
    const buf_rsrc r_xa = make_rsrc(xa, xa_rows * lda * 4), r_out = make_rsrc(out, N * ldo * 4);
    const buf_rsrc r_T = make_rsrc(T ? T : out, T ? N * ldt * 4 : 0);
    const buf_rsrc r_side = make_rsrc(pro.side ? pro.side : out, pro.side ? N * pro.lds * 4 : 0);
    const buf_rsrc r_mask = make_rsrc(mask, N);
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    auto row_of = [&](int st) __attribute__((always_inline)) -> int {
        const int slot = 16 * st + rs;
        return (st < nst && slot < n_rows) ? (int)(r0 + slot) : -1;
    };
    // gathered operand: the index of stage s is loaded two issues before the rows of stage s (ia = index of the next stage
    // to be issued, ib = of the one after it)
    auto index_of = [&](int st) __attribute__((always_inline)) -> int {
        const int r = row_of(st);
        if (r < 0) return 0;
        const int64_t g = xa_index[r];
        return (int)(g < 0 ? 0 : (g >= xa_rows ? xa_rows - 1 : g));
    };
    int ia = 0, ib = 0;
    if (xa_index) ia = index_of(0), ib = index_of(1);
    struct Raw {
        float4 a;
        unsigned mk;
        int row;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int r = row_of(st);
        int src = r;
        if (xa_index) {
            src = ia;
            ia = ib;
            ib = index_of(st + 2);
        }
        R.row = r;
        R.a = buf_load4(r_xa, r >= 0 ? (int)((src * lda + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_mask, r >= 0 ? r : kBufOOB, 0, 0);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    GnCoefRegs CR;
    const bool fold_here = pro.saved != nullptr;
    if (fold_here && pro.src.acc) gn_fwd_coef_issue<H, THREADS>(pro.src, CR);
    const float4* img = reinterpret_cast<const float4*>(Wimg);
    float4 bw1[KF4], bw0[KF4];
#pragma unroll
    for (int tt = 0; tt < KF4; ++tt) {
        bw1[tt] = img[(((tt >> 2) * 2 * NTL + w) * 4 + (tt & 3)) * 64 + lane];
        bw0[tt] = img[(((tt >> 2) * 2 * NTL + NTL + w) * 4 + (tt & 3)) * 64 + lane];
    }
    float b1 = bias[16 * w + j], b0 = bias[H + 16 * w + j];
    if (fold_here && pro.src.acc) {
        gn_fwd_coef_issue_params<H>(pro.src, CR);
        glass_pin(CR.gamma);
        glass_pin(CR.beta);
        glass_pin(CR.alpha);
    }
    glass_pin(b1);
    glass_pin(b0);
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    if (fold_here) gn_fwd_coef_finish<H, THREADS>(pro.src, pro.saved, N, CR, gn_coef_s);
    // SP: the wave's two weight slices as bf16 pieces, block b = the lane's k = 32 q + 8 b .. + 7 (bw[2b], bw[2b + 1])
    uint4 bwc1[SP ? KF4 / 2 : 1][3], bwc0[SP ? KF4 / 2 : 1][3];
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < KF4 / 2; ++b) {
            const Split4 s0 = split4(bw1[2 * b]), s1 = split4(bw1[2 * b + 1]);
            bwc1[b][0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
            bwc1[b][1] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
            bwc1[b][2] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
            const Split4 u0 = split4(bw0[2 * b]), u1 = split4(bw0[2 * b + 1]);
            bwc0[b][0] = make_uint4(u0.hi.x, u0.hi.y, u1.hi.x, u1.hi.y);
            bwc0[b][1] = make_uint4(u0.mid.x, u0.mid.y, u1.mid.x, u1.mid.y);
            bwc0[b][2] = make_uint4(u0.lo.x, u0.lo.y, u1.lo.x, u1.lo.y);
        }
    }
    lds_barrier();  // coefficients
    const bool pro_on = pro.saved != nullptr;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (pro_on) {
        const float4 s4 = *reinterpret_cast<const float4*>(gn_coef_s + 4 * ga);
        const float4 h4 = *reinterpret_cast<const float4*>(gn_coef_s + H + 4 * ga);
        sc[0] = s4.x, sc[1] = s4.y, sc[2] = s4.z, sc[3] = s4.w;
        sh[0] = h4.x, sh[1] = h4.y, sh[2] = h4.z, sh[3] = h4.w;
    }
    const int pact = pro_on ? pro.act : GLASS_ACT_NONE;
    const bool drop_on = pro_on && drop.p > 0.f;
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        const int r = R.row;
        float a[4] = {R.a.x, R.a.y, R.a.z, R.a.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop_on) drop_scales<4>(drop, r < 0 ? 0 : r, 4 * ga, ds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float h = fmaf(a[k], sc[k], sh[k]);
            h = act_fast(pact, h);
            a[k] = h * ds[k];
        }
        const float4 v = make_float4(a[0], a[1], a[2], a[3]);
        buf_store4(r_side, (pro_on && pro.side && r >= 0) ? (int)((r * pro.lds + 4 * ga) * 4) : kBufOOB, v);
        if constexpr (SP) {
            const Split4 sa = split4(v);
            unsigned short* P = tile_s[st & 1] + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sa.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sa.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sa.lo;
        } else {
            *reinterpret_cast<float4*>(tile[SP ? 0 : (st & 1)] + rs * RS + 4 * ga) = v;
        }
        if (ga == 0) rowflag_s[st & 1][rs] = (r >= 0 && R.mk != 0) ? (r | (1 << 30)) : r;
    };
    float ssum = 0.f, ssq = 0.f;
    double dsum = 0.0, dsq = 0.0;  // column sums: float over four stages (16 values per lane), double across
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    const int c = 16 * w + j;
    auto stage = [&](int st, Raw& Rn) __attribute__((always_inline)) {
        int rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = rowflag_s[st & 1][4 * q + r];
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc0 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
            // lane (j, q): row j of the stage, k = 32 q + 8 b .. + 7 of block b: one 16-byte read per piece
            const unsigned short* P = tile_s[st & 1] + j * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b) {
                uint4 af[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
#define GLASS_SMMA16(ACC, BW, pa, pb)                                                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[pa]), __builtin_bit_cast(bf16x8, BW[b][pb]), ACC, 0, 0, 0)
                GLASS_SMMA16(acc1, bwc1, 1, 1); GLASS_SMMA16(acc0, bwc0, 1, 1);  // small terms first; the two halves' chains interleave
                GLASS_SMMA16(acc1, bwc1, 2, 0); GLASS_SMMA16(acc0, bwc0, 2, 0);
                GLASS_SMMA16(acc1, bwc1, 0, 2); GLASS_SMMA16(acc0, bwc0, 0, 2);
                GLASS_SMMA16(acc1, bwc1, 1, 0); GLASS_SMMA16(acc0, bwc0, 1, 0);
                GLASS_SMMA16(acc1, bwc1, 0, 1); GLASS_SMMA16(acc0, bwc0, 0, 1);
                GLASS_SMMA16(acc1, bwc1, 0, 0); GLASS_SMMA16(acc0, bwc0, 0, 0);
#undef GLASS_SMMA16
            }
        } else {
            const float* A = tile[SP ? 0 : (st & 1)] + j * RS + (KT / 4) * q;
            float4 a4[KF4];
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(A + 4 * tt);
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) {
                const float x[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w};
                const float y1[4] = {bw1[tt].x, bw1[tt].y, bw1[tt].z, bw1[tt].w}, y0[4] = {bw0[tt].x, bw0[tt].y, bw0[tt].z, bw0[tt].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y1[e], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y0[e], acc0, 0, 0, 0);
                }
            }
        }
        if (st + 1 < nst) {
            commit(st + 1, Rn);
            issue(st + 3, Rn);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0;
            const int row = rv[r] & ((1 << 30) - 1);
            const bool lab = (rv[r] >> 30) & 1;
            const float w1 = lab ? zr : omz, w0 = lab ? omz : zr;
            const float v1 = acc1[r] + b1, v0 = acc0[r] + b0;
            buf_store1(r_T, (live && T) ? (int)((row * ldt + c) * 4) : kBufOOB, v1);
            buf_store1(r_T, (live && T) ? (int)((row * ldt + H + c) * 4) : kBufOOB, v0);
            float a1 = v1, a0 = v0;
            a1 = act_fast(act, a1), a0 = act_fast(act, a0);
            const float o = w1 * a1 + w0 * a0;
            buf_store1(r_out, live ? (int)((row * ldo + c) * 4) : kBufOOB, o);
            ssum += live ? o : 0.f;
            ssq += live ? o * o : 0.f;
        }
        if ((st & 3) == 3) {
            dsum += (double)ssum, dsq += (double)ssq;
            ssum = ssq = 0.f;
        }
        if (st + 1 < nst) lds_barrier();
    };
    for (int st = 0; st < nst; st += 2) {
        stage(st, rawB);
        if (st + 1 < nst) stage(st + 1, rawA);
    }
    if (stats == nullptr) return;
    dsum += (double)ssum, dsq += (double)ssq;
    dsum += __shfl_xor(dsum, 16);
    dsq += __shfl_xor(dsq, 16);
    dsum += __shfl_xor(dsum, 32);
    dsq += __shfl_xor(dsq, 32);
    if (q != 0) return;
    stats[((size_t)blockIdx.x * 2) * H + c] = dsum;
    stats[((size_t)blockIdx.x * 2 + 1) * H + c] = dsq;
    const int64_t n_entries = (N + 63) / 64;
    for (int64_t e = (int64_t)blockIdx.x + gridDim.x; e < n_entries; e += gridDim.x) {
        stats[((size_t)e * 2) * H + c] = 0.0;
        stats[((size_t)e * 2 + 1) * H + c] = 0.0;
    }
}

// ---- comb data gradient, stage-run form (hidden 128; round 6) --------------------------------------------------------------
// d[g || x_][r] = dc[r] . W_eff(label of r)  ([N, 128] x [128, 256]; no activation, so the mix folds into the weight).  The
// LDS-tiled kernel served this at hidden 128 (tiled_dgrad_kernel<128, 256, 128, 256>: 60.7 us at em_user-shape for ~100 MB of
// traffic, 1.8 + 16.3 + 20.1 us by phase with the tiles in lockstep).  Here trans_fwd3_kernel's form: eight waves, wave w owns
// columns 16 w .. + 15 of BOTH output halves (dg and dx_) and keeps the two slices of the UNLABELED effective weight in
// registers while its workgroup walks rows_main rows in 16-row stages.  Labeled rows (a batch's subgraph nodes: a few per
// workgroup) are skipped by that pass — the workgroup lists them up front (ordered compaction of its label bytes) — and done
// in a second, short pass with the LABELED effective weight loaded into the same registers and the rows gathered by the
// list.  No row list from the caller, no extra workgroups, one statistics entry per workgroup (both passes in one sum).
// The first 128 output columns are the gradient of conv.gn's output: its backward column sums ride in the epilogue as in
// trans_dgrad3_kernel (u and xhat . u prepared by the loader, through LDS).
// Image: layout kLayoutWave16EffDgradCols ([256 outputs][128], unlabeled then labeled).
constexpr int kCombDgrad3List = 4096;  // labeled rows a workgroup can list = its most rows (the host caps rows_main)
template <int H, bool SP>
__global__ __launch_bounds__(4 * H) void comb_dgrad3_kernel(const float* __restrict__ dsrc, int64_t ldd,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ WT,
                                                            const uint64_t* __restrict__ rng_state, float* __restrict__ out,
                                                            int64_t ldo, int64_t N, GnBwdStats gs, int rows_main) {
    static_assert(H == 128, "eight waves x 16 columns of both halves");
    extern __shared__ float4 lds_cd3[];
    float* lds = reinterpret_cast<float*>(lds_cd3);
    constexpr int THREADS = 4 * H, NTL = H / 16, KF4 = H / 16;
    constexpr int KT = H, RS = KT + 4, RP = H + 4;
    constexpr int RSB = KT + 8, kPlane = 16 * RSB;
    constexpr int kAFloats = SP ? (3 * kPlane) / 2 : 16 * RS;
    constexpr int kBuf = kAFloats + 2 * 16 * RP;  // A | U | XU  (floats per stage buffer)
    int* rowflag_s = reinterpret_cast<int*>(lds + 2 * kBuf);   // [2][16]: row of each slot (-1 none; bit 30: computed, not stored / counted)
    int* wave_cnt = rowflag_s + 32;                             // [8]
    int* list = wave_cnt + 8;                                   // [kCombDgrad3List]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid / (H / 4), ga = tid % (H / 4);
    const int64_t r0 = (int64_t)blockIdx.x * rows_main;
    const int n_rows = (int)(r0 + rows_main <= N ? rows_main : N - r0);
    const bool gn_on = gs.partial != nullptr;
    const buf_rsrc r_d = make_rsrc(dsrc, N * ldd * 4), r_out = make_rsrc(out, N * ldo * 4), r_m = make_rsrc(mask, N);
    const buf_rsrc r_x = make_rsrc(gn_on ? gs.x : dsrc, gn_on ? N * gs.ldx * 4 : 0);
    struct Raw {
        float4 d, x;
        unsigned mk;
        int row;
    };
    int n_cur = n_rows;   // rows of the running pass
    bool labp = false;    // second pass: the listed (labeled) rows
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int slot = 16 * st + rs;
        int r = -1;
        if (slot < n_cur) r = labp ? list[slot] : (int)(r0 + slot);
        R.row = r;
        R.d = buf_load4(r_d, r >= 0 ? (int)((r * ldd + 4 * ga) * 4) : kBufOOB);
        R.x = buf_load4(r_x, r >= 0 ? (int)((r * gs.ldx + 4 * ga) * 4) : kBufOOB);
        R.mk = __builtin_amdgcn_raw_buffer_load_b8(r_m, r >= 0 ? r : kBufOOB, 0, 0);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    // this workgroup's labeled rows, in row order
    int n_lab = 0;
    for (int p = 0; p < n_rows; p += THREADS) {
        const int slot = p + tid;
        const bool flag = slot < n_rows && mask[r0 + slot] != 0;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        int off = n_lab, total = 0;
#pragma unroll
        for (int ww = 0; ww < THREADS / 64; ++ww) {
            if (ww < w) off += wave_cnt[ww];
            total += wave_cnt[ww];
        }
        if (flag) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(r0 + slot);
        n_lab += total;
        __syncthreads();
    }
    const float4* img = reinterpret_cast<const float4*>(WT);
    float4 bw1[KF4], bw0[KF4];
    uint4 bwc1[SP ? KF4 / 2 : 1][3], bwc0[SP ? KF4 / 2 : 1][3];
    auto load_weights = [&](int image) __attribute__((always_inline)) {
        const float4* im = img + (size_t)image * (2 * H * KT / 4);
#pragma unroll
        for (int tt = 0; tt < KF4; ++tt) {
            bw1[tt] = im[(((tt >> 2) * 2 * NTL + w) * 4 + (tt & 3)) * 64 + lane];
            bw0[tt] = im[(((tt >> 2) * 2 * NTL + NTL + w) * 4 + (tt & 3)) * 64 + lane];
        }
        if constexpr (SP) {  // block b = the lane's k = 32 q + 8 b .. + 7 (bw[2b], bw[2b + 1])
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b) {
                const Split4 s0 = split4(bw1[2 * b]), s1 = split4(bw1[2 * b + 1]);
                bwc1[b][0] = make_uint4(s0.hi.x, s0.hi.y, s1.hi.x, s1.hi.y);
                bwc1[b][1] = make_uint4(s0.mid.x, s0.mid.y, s1.mid.x, s1.mid.y);
                bwc1[b][2] = make_uint4(s0.lo.x, s0.lo.y, s1.lo.x, s1.lo.y);
                const Split4 u0 = split4(bw0[2 * b]), u1 = split4(bw0[2 * b + 1]);
                bwc0[b][0] = make_uint4(u0.hi.x, u0.hi.y, u1.hi.x, u1.hi.y);
                bwc0[b][1] = make_uint4(u0.mid.x, u0.mid.y, u1.mid.x, u1.mid.y);
                bwc0[b][2] = make_uint4(u0.lo.x, u0.lo.y, u1.lo.x, u1.lo.y);
            }
        }
    };
    load_weights(0);
    const bool gdrop_on = gn_on && gs.drop.p > 0.f;
    Drop gdrop = gs.drop;
    if (gdrop_on) {
        gdrop.seed = rng_state[0];
        gdrop.step = rng_state[1];
    }
    float g_mu[4] = {0.f, 0.f, 0.f, 0.f}, g_rs[4] = {0.f, 0.f, 0.f, 0.f}, g_al[4] = {0.f, 0.f, 0.f, 0.f};
    float g_sc[4] = {0.f, 0.f, 0.f, 0.f}, g_sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (gn_on) {
        const float4 m4 = *reinterpret_cast<const float4*>(gs.saved + 4 * ga), r4 = *reinterpret_cast<const float4*>(gs.saved + H + 4 * ga);
        const float4 s4 = *reinterpret_cast<const float4*>(gs.saved + 2 * H + 4 * ga), h4 = *reinterpret_cast<const float4*>(gs.saved + 3 * H + 4 * ga);
        const float4 a4 = *reinterpret_cast<const float4*>(gs.alpha + 4 * ga);
        g_mu[0] = m4.x, g_mu[1] = m4.y, g_mu[2] = m4.z, g_mu[3] = m4.w;
        g_rs[0] = r4.x, g_rs[1] = r4.y, g_rs[2] = r4.z, g_rs[3] = r4.w;
        g_sc[0] = s4.x, g_sc[1] = s4.y, g_sc[2] = s4.z, g_sc[3] = s4.w;
        g_sh[0] = h4.x, g_sh[1] = h4.y, g_sh[2] = h4.z, g_sh[3] = h4.w;
        g_al[0] = a4.x, g_al[1] = a4.y, g_al[2] = a4.z, g_al[3] = a4.w;
    }
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        float* At = lds + (st & 1) * kBuf;
        float* U = At + kAFloats;
        float* XU = U + 16 * RP;
        const int r = R.row < 0 ? 0 : R.row;
        const float xv[4] = {R.x.x, R.x.y, R.x.z, R.x.w};
        float gds[4] = {1.f, 1.f, 1.f, 1.f}, u[4], xu[4];
        if (gdrop_on) drop_scales<4>(gdrop, r, 4 * ga, gds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float uk = gds[k];
            uk *= act_grad(gs.act, fmaf(xv[k], g_sc[k], g_sh[k]));
            u[k] = uk;
            xu[k] = (xv[k] - g_al[k] * g_mu[k]) * g_rs[k] * uk;
        }
        if constexpr (SP) {
            const Split4 sd = split4(R.d);
            unsigned short* P = reinterpret_cast<unsigned short*>(At) + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sd.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sd.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sd.lo;
        } else {
            *reinterpret_cast<float4*>(At + rs * RS + 4 * ga) = R.d;
        }
        *reinterpret_cast<float4*>(U + rs * RP + 4 * ga) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4*>(XU + rs * RP + 4 * ga) = make_float4(xu[0], xu[1], xu[2], xu[3]);
        // first pass: a labeled row is computed with the wrong weight and dropped (the second pass owns it)
        if (ga == 0) rowflag_s[(st & 1) * 16 + rs] = (R.row >= 0 && !labp && R.mk != 0) ? (R.row | (1 << 30)) : R.row;
    };
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;  // column sums: float over four stages, double across
    const int c = 16 * w + j;
    int nst = (n_cur + 15) / 16;
    auto stage = [&](int st, Raw& Rn) __attribute__((always_inline)) {
        const float* At = lds + (st & 1) * kBuf;
        const float* U = At + kAFloats;
        const float* XU = U + 16 * RP;
        int rv[4];
        float uu[4], xx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rv[r] = rowflag_s[(st & 1) * 16 + 4 * q + r];
            uu[r] = U[(4 * q + r) * RP + c];
            xx[r] = XU[(4 * q + r) * RP + c];
        }
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc0 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
            const unsigned short* P = reinterpret_cast<const unsigned short*>(At) + j * RSB + (KT / 4) * q;
#pragma unroll
            for (int b = 0; b < KF4 / 2; ++b) {
                uint4 af[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
#define GLASS_SMMA16(ACC, BW, pa, pb)                                                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[pa]), __builtin_bit_cast(bf16x8, BW[b][pb]), ACC, 0, 0, 0)
                GLASS_SMMA16(acc1, bwc1, 1, 1); GLASS_SMMA16(acc0, bwc0, 1, 1);
                GLASS_SMMA16(acc1, bwc1, 2, 0); GLASS_SMMA16(acc0, bwc0, 2, 0);
                GLASS_SMMA16(acc1, bwc1, 0, 2); GLASS_SMMA16(acc0, bwc0, 0, 2);
                GLASS_SMMA16(acc1, bwc1, 1, 0); GLASS_SMMA16(acc0, bwc0, 1, 0);
                GLASS_SMMA16(acc1, bwc1, 0, 1); GLASS_SMMA16(acc0, bwc0, 0, 1);
                GLASS_SMMA16(acc1, bwc1, 0, 0); GLASS_SMMA16(acc0, bwc0, 0, 0);
#undef GLASS_SMMA16
            }
        } else {
            const float* A = At + j * RS + (KT / 4) * q;
            float4 a4[KF4];
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) a4[tt] = *reinterpret_cast<const float4*>(A + 4 * tt);
#pragma unroll
            for (int tt = 0; tt < KF4; ++tt) {
                const float x[4] = {a4[tt].x, a4[tt].y, a4[tt].z, a4[tt].w};
                const float y1[4] = {bw1[tt].x, bw1[tt].y, bw1[tt].z, bw1[tt].w}, y0[4] = {bw0[tt].x, bw0[tt].y, bw0[tt].z, bw0[tt].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y1[e], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y0[e], acc0, 0, 0, 0);
                }
            }
        }
        if (st + 1 < nst) {
            commit(st + 1, Rn);
            issue(st + 3, Rn);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0 && !(rv[r] >> 30);
            buf_store1(r_out, live ? (int)((rv[r] * ldo + c) * 4) : kBufOOB, acc1[r]);
            buf_store1(r_out, live ? (int)((rv[r] * ldo + H + c) * 4) : kBufOOB, acc0[r]);
            const float vl = live ? acc1[r] : 0.f;
            s1 = fmaf(vl, uu[r], s1);
            s2 = fmaf(vl, xx[r], s2);
        }
        if ((st & 3) == 3) {
            d1 += (double)s1, d2 += (double)s2;
            s1 = s2 = 0.f;
        }
        if (st + 1 < nst) lds_barrier();
    };
    auto run = [&]() __attribute__((always_inline)) {  // (stages 0 and 1 of the pass are requested)
        commit(0, rawA);
        issue(2, rawA);
        lds_barrier();
        for (int st = 0; st < nst; st += 2) {
            stage(st, rawB);
            if (st + 1 < nst) stage(st + 1, rawA);
        }
        d1 += (double)s1, d2 += (double)s2;
        s1 = s2 = 0.f;
    };
    run();
    if (n_lab > 0) {  // (workgroup-uniform)
        lds_barrier();  // the first pass's last stage is read; the list is complete since the prologue
        labp = true;
        n_cur = n_lab;
        nst = (n_cur + 15) / 16;
        issue(0, rawA);
        issue(1, rawB);
        load_weights(1);
        run();
    }
    if (!gn_on) return;
    d1 += __shfl_xor(d1, 16);
    d2 += __shfl_xor(d2, 16);
    d1 += __shfl_xor(d1, 32);
    d2 += __shfl_xor(d2, 32);
    if (q != 0) return;
    gs.partial[((size_t)blockIdx.x * 2) * H + c] = d1;
    gs.partial[((size_t)blockIdx.x * 2 + 1) * H + c] = d2;
    const int64_t n_entries = (N + 63) / 64;
    for (int64_t e = (int64_t)blockIdx.x + gridDim.x; e < n_entries; e += gridDim.x) {
        gs.partial[((size_t)e * 2) * H + c] = 0.0;
        gs.partial[((size_t)e * 2 + 1) * H + c] = 0.0;
    }
}
template <bool SP>
constexpr size_t comb_dgrad3_lds() {
    constexpr int H = 128, kA = SP ? (3 * 16 * (H + 8)) / 2 : 16 * (H + 4);
    return (size_t)(2 * (kA + 2 * 16 * (H + 4)) + 32 + 8 + kCombDgrad3List) * sizeof(float);
}

// Data gradient of the comb pair in the same form:  d[g || x_][r] = dc[r] . (w1 W1 + w0 W0): K = H instead of 2H (one
// K pass), the mix coefficient folded into the weight.  The first H output columns are the gradient of conv.gn's output:
// its backward column sums are accumulated by the epilogue as in dual_dgrad_body (one partial per workgroup, the extra
// ones included).
struct DgradEffArgs {
    const float* dsrc; int64_t ldd;
    const uint8_t* mask;
    const float* WT;  // unl | lab images
    const uint64_t* rng_state;
    float* out; int64_t ldo;
    int64_t N;
    GnBwdStats gs;
    LabRows lab;
    GnBwdSrc src;  // src.acc != nullptr: dsrc is derived on load (staged bodies only)
};

struct DcRaw {
    float d[kKC];
};

template <int H, int RW>
__device__ __forceinline__ void comb_dgrad_eff_body(const float* __restrict__ dsrc, int64_t ldd,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ WT,
                                                    const uint64_t* __restrict__ rng_state, float* __restrict__ out,
                                                    int64_t ldo, int64_t N, GnBwdStats gs, LabRows lab, int block,
                                                    float4* lds_w) {
    constexpr int KT = H, KQ = KT / 4, NT = 2 * H, NGL = NT / 64, NLOC = 4 * NGL;
    constexpr int THREADS = kWave * RW;
    static_assert(H == 64 && KQ == kKC, "one K pass of 64");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    EffRows R;
    if (!eff_rows<RW>(R, block, mask, N, lab, w, i, q)) {  // extra workgroup beyond the list: an empty partial
        if (gs.partial && !gs.exact)
            for (int c = threadIdx.x; c < 2 * H; c += THREADS) gs.partial[(size_t)block * 2 * H + c] = 0.0;
        return;
    }
    const bool row_ok = R.row >= 0;
    const float* drow = dsrc + (row_ok ? R.row : 0) * ldd + q * KQ;
    const float* W = WT + (R.extra ? NT * KT : 0);
    f32x4 acc[NLOC];
#pragma unroll
    for (int t = 0; t < NLOC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 g_mu, g_rstd, g_scale, g_shift, g_al;
    if (gs.partial) {  // before the product: the epilogue then waits for the GraphNorm input rows only
        if (gs.drop.p > 0.f) {
            gs.drop.seed = rng_state[0];
            gs.drop.step = rng_state[1];
        }
        g_mu = *reinterpret_cast<const float4*>(gs.saved + 4 * i);
        g_rstd = *reinterpret_cast<const float4*>(gs.saved + H + 4 * i);
        g_scale = *reinterpret_cast<const float4*>(gs.saved + 2 * H + 4 * i);
        g_shift = *reinterpret_cast<const float4*>(gs.saved + 3 * H + 4 * i);
        g_al = *reinterpret_cast<const float4*>(gs.alpha + 4 * i);
    }
    staged_product<NT, KT, NLOC, NLOC, THREADS, DcRaw>(
        acc, W, lds_w, lane, 0, 0,
        [&](int kc, DcRaw& raw) __attribute__((always_inline)) { load16(raw.d, drow + kc * kKC, row_ok); },
        [&](int, const DcRaw& raw, float (&a)[kKC]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < kKC; ++s) a[s] = raw.d[s];
        });
    D_STAMP(3, 2);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        if (R.erow[reg] < 0) continue;
        const int64_t r = R.erow[reg];
#pragma unroll
        for (int gl = 0; gl < NGL; ++gl) {
            const int c = 64 * gl + 4 * i;
            const float4 v = make_float4(acc[4 * gl][reg], acc[4 * gl + 1][reg], acc[4 * gl + 2][reg], acc[4 * gl + 3][reg]);
            *reinterpret_cast<float4*>(out + r * ldo + c) = v;
            if (gs.partial && gl == 0) {
                const float4 x4 = *reinterpret_cast<const float4*>(gs.x + r * gs.ldx + c);
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, dy[4] = {v.x, v.y, v.z, v.w};
                const float mu[4] = {g_mu.x, g_mu.y, g_mu.z, g_mu.w}, rs[4] = {g_rstd.x, g_rstd.y, g_rstd.z, g_rstd.w};
                const float sc[4] = {g_scale.x, g_scale.y, g_scale.z, g_scale.w};
                const float sh[4] = {g_shift.x, g_shift.y, g_shift.z, g_shift.w};
                const float al[4] = {g_al.x, g_al.y, g_al.z, g_al.w};
                float ds[4] = {1.f, 1.f, 1.f, 1.f};
                if (gs.drop.p > 0.f) drop_scales<4>(gs.drop, r, c, ds);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float gp = dy[k] * ds[k];
                    gp *= act_grad(gs.act, fmaf(xv[k], sc[k], sh[k]));
                    const float xhat = (xv[k] - al[k] * mu[k]) * rs[k];
                    s1[k] += gp;
                    s2[k] = fmaf(gp, xhat, s2[k]);
                }
            }
        }
    }
    D_STAMP(3, 3);
    if (gs.partial == nullptr) return;
    __syncthreads();  // every wave is done with the weight image in LDS
    double* red = reinterpret_cast<double*>(lds_w);  // [RW row waves][H][2]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double a = (double)s1[k], b2 = (double)s2[k];
        a += __shfl_xor(a, 16);
        b2 += __shfl_xor(b2, 16);
        a += __shfl_xor(a, 32);
        b2 += __shfl_xor(b2, 32);
        if (q == 0) {
            red[(w * H + 4 * i + k) * 2] = a;
            red[(w * H + 4 * i + k) * 2 + 1] = b2;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += THREADS) {
        double a = 0.0, b2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < RW; ++ww) {
            a += red[(ww * H + c) * 2];
            b2 += red[(ww * H + c) * 2 + 1];
        }
        if (gs.exact) {  // exact accumulators (gn_acc.h): no finalize launch behind this kernel
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 0, c, H, a, kAccScaleBwd);
            gn_acc_add(reinterpret_cast<long long*>(gs.partial), block % gs.exact, 1, c, H, b2, kAccScaleBwd);
        } else {
            gs.partial[((size_t)block * 2) * H + c] = a;
            gs.partial[((size_t)block * 2 + 1) * H + c] = b2;
        }
    }
}

template <int H>
__global__ __launch_bounds__(kBlock) void comb_dgrad_eff_kernel(DgradEffArgs A) {
    extern __shared__ float4 lds_w[];
    comb_dgrad_eff_body<H, 4>(A.dsrc, A.ldd, A.mask, A.WT, A.rng_state, A.out, A.ldo, A.N, A.gs, A.lab, blockIdx.x, lds_w);
}

// ---- comb backward in the staged form (hidden 64): data gradient AND weight-gradient partial from ONE pass over the rows
// The first form runs the data gradient's row tiles and the weight gradient's row slabs as two kinds of workgroup of one
// launch: both read dc, each is latency-bound, and the launch lasts as long as the slower kind (phase stamps: 11.1 / 12.1 us
// lives inside a 14.9 us launch).  Here a workgroup takes 64 rows through LDS in four 16-row stages and uses every stage
// twice: (a) data gradient d[g || x_] = dc . W_eff (a wave owns 16 columns of each half; the effective weight in 32
// registers), with the GraphNorm backward sums of the g half from two tiles the LOADER threads prepare (keep-scale u and
// xhat * u per element: one dropout hash per float4, none in the epilogue); (b) the stage's contribution to this
// workgroup's own weight-gradient partial  S_wg[o][i] = sum_rows dc[r][o] * [g || x_][r][i]  (a wave owns 16 outputs o and all
// 128 inputs: 8 accumulator tiles; both operands read from TRANSPOSED tiles, four rows per b128).  The row tiles'
// partials are the S tiles of the S / L form (every row, labeled or not); the extra workgroups, which hold the listed
// labeled rows, produce the L tiles — the batched reduce is the same (plain [o][i] order, header[2] = 1).  No separate
// weight-gradient workgroups: one round of 280 workgroups instead of 505.
// Image: layout kLayoutWave16EffDgradCols (tile t = columns 64 (t >> 2) + 16 (t & 3) .. + 15 of [dg || dx_]).
// The data-gradient half alone, as a device function for the workgroups [0, n_dgrad) of comb_bwd_eff_kernel (the weight
// gradient stays with its own workgroups there): image in layout kLayoutWave16EffDgradCols, `lds` >= kCombDgrad2Lds bytes.
// SP: the two products in the split form (split_mma.h; see trans_dgrad2_body): dc cut once by the loader into three bf16
// planes ([piece][row][k], rows 72 bf16 apart), the two weight slices cut once per wave; 24 MFMAs of 16 cycles per stage.
template <int H, bool DROP, bool SP = false>
__device__ __forceinline__ void comb_dgrad2_body(const DgradEffArgs& A, int blk, float* lds) {
    static_assert(H == 64, "four waves x 16 columns");
    constexpr int RS = H + 4;   // plain tiles [16 rows][H]: row stride (floats)
    constexpr int RSB = H + 8, kPlane = 16 * RSB;  // SP: bf16 per row of a piece plane, per plane
    constexpr int kDcFloats = SP ? (3 * kPlane) / 2 : 16 * RS;
    struct __attribute__((aligned(16))) Stage {
        float dcP[kDcFloats];     // dc rows (A operand; SP: the three bf16 planes)
        float U[16 * RS];         // keep-scale of the GraphNorm's dropout per element of the g half
        float XU[16 * RS];        // xhat * keep-scale
    };
    Stage* stg = reinterpret_cast<Stage*>(lds);
    int* rows_s = reinterpret_cast<int*>(lds + 2 * (kDcFloats + 2 * 16 * RS));
    float* coef_s = lds + 2 * (kDcFloats + 2 * 16 * RS) + 64;  // [5][64] A | Bx | K | scale | shift of the GraphNorm dc comes from (src)
    const GnBwdSrc& src = A.src;
    const bool src_on = src.acc != nullptr;
    D_STAMP(3, 0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int n_main = A.lab.n_main;
    const bool extra = (int)blk >= n_main;
    const GnBwdStats& gs = A.gs;
    const int64_t N = A.N;
    int n_lab = 0, base = 0;
    if (extra) {
        n_lab = A.lab.count[0];
        base = ((int)blk - n_main) * 64;
        if (base >= n_lab) {  // beyond the list: an empty L tile, empty sums
            if (gs.partial && !gs.exact)
                for (int c = tid; c < 2 * H; c += kBlock) gs.partial[(size_t)blk * 2 * H + c] = 0.0;
            return;
        }
    }
    const buf_rsrc r_dc = src_on ? make_rsrc(src.dy, N * src.lddy * 4) : make_rsrc(A.dsrc, N * A.ldd * 4);
    const int64_t ld_dc = src_on ? src.lddy : A.ldd;
    const buf_rsrc r_sx = make_rsrc(src_on ? src.x : A.out, src_on ? N * src.ldx * 4 : 0);
    const buf_rsrc r_sad = make_rsrc((src_on && src.addend) ? src.addend : A.out, (src_on && src.addend) ? N * src.ldadd * 4 : 0);
    const buf_rsrc r_out = make_rsrc(A.out, N * A.ldo * 4);
    const buf_rsrc r_a = make_rsrc(gs.partial ? gs.x : A.dsrc, gs.partial ? N * gs.ldx * 4 : 0);
    // this wave's slices of the effective weight (transposed operand): columns 16w .. of the dg half and of the dx_ half
    const float4* img = reinterpret_cast<const float4*>(A.WT + (extra ? 2 * H * H : 0));
    float4 bwg[4], bwx[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bwg[v] = img[(w * 4 + v) * 64 + lane];
        bwx[v] = img[((4 + w) * 4 + v) * 64 + lane];
    }
    uint4 bwgc[SP ? 2 : 1][3], bwxc[SP ? 2 : 1][3];  // SP: block b = the lane's k = 16 q + 8 b .. + 7
    if constexpr (SP) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const Split4 c0 = split4(bwg[2 * b]), c1 = split4(bwg[2 * b + 1]);
            bwgc[b][0] = make_uint4(c0.hi.x, c0.hi.y, c1.hi.x, c1.hi.y);
            bwgc[b][1] = make_uint4(c0.mid.x, c0.mid.y, c1.mid.x, c1.mid.y);
            bwgc[b][2] = make_uint4(c0.lo.x, c0.lo.y, c1.lo.x, c1.lo.y);
            const Split4 d0 = split4(bwx[2 * b]), d1 = split4(bwx[2 * b + 1]);
            bwxc[b][0] = make_uint4(d0.hi.x, d0.hi.y, d1.hi.x, d1.hi.y);
            bwxc[b][1] = make_uint4(d0.mid.x, d0.mid.y, d1.mid.x, d1.mid.y);
            bwxc[b][2] = make_uint4(d0.lo.x, d0.lo.y, d1.lo.x, d1.lo.y);
        }
    }
    const int rs = tid >> 4, ga = tid & 15;  // loader role: row rs of the stage, columns 4 ga .. 4 ga + 3
    int my_row[4];
    int slot_v = -1;
    unsigned char slot_mask = 0;
    if (!extra) {
        const int64_t r0 = (int64_t)blk * 64;
#pragma unroll
        for (int st = 0; st < 4; ++st) my_row[st] = r0 + 16 * st + rs < N ? (int)(r0 + 16 * st + rs) : -1;
        if (tid < 64 && r0 + tid < N) {
            slot_v = (int)(r0 + tid);
            slot_mask = A.mask[r0 + tid];
        }
    } else {
        if (tid < 64) {
            slot_v = base + tid < n_lab ? A.lab.rows[base + tid] : -1;
            rows_s[tid] = slot_v;
        }
        lds_barrier();
#pragma unroll
        for (int st = 0; st < 4; ++st) my_row[st] = rows_s[16 * st + rs];
    }
    struct Raw {
        float4 dc, a, sx, sad;  // (src: dc holds dy; sx / sad the GraphNorm's input and the addend)
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int r = my_row[st];
        R.dc = buf_load4(r_dc, r >= 0 ? (int)((r * ld_dc + 4 * ga) * 4) : kBufOOB);
        R.sx = buf_load4(r_sx, r >= 0 ? (int)((r * src.ldx + 4 * ga) * 4) : kBufOOB);
        R.sad = buf_load4(r_sad, r >= 0 ? (int)((r * src.ldadd + 4 * ga) * 4) : kBufOOB);
        R.a = buf_load4(r_a, r >= 0 ? (int)((r * gs.ldx + 4 * ga) * 4) : kBufOOB);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    // GraphNorm coefficients of this loader thread's four columns
    float g_mu[4] = {0.f, 0.f, 0.f, 0.f}, g_rs[4] = {0.f, 0.f, 0.f, 0.f}, g_al[4] = {0.f, 0.f, 0.f, 0.f};
    Drop drop = gs.drop;
    if (gs.partial) {
        if (DROP) {
            drop.seed = A.rng_state[0];
            drop.step = A.rng_state[1];
        }
        const float4 m4 = *reinterpret_cast<const float4*>(gs.saved + 4 * ga);
        const float4 r4 = *reinterpret_cast<const float4*>(gs.saved + H + 4 * ga);
        const float4 a4 = *reinterpret_cast<const float4*>(gs.alpha + 4 * ga);
        g_mu[0] = m4.x, g_mu[1] = m4.y, g_mu[2] = m4.z, g_mu[3] = m4.w;
        g_rs[0] = r4.x, g_rs[1] = r4.y, g_rs[2] = r4.z, g_rs[3] = r4.w;
        g_al[0] = a4.x, g_al[1] = a4.y, g_al[2] = a4.z, g_al[3] = a4.w;
    }
    if (tid < 64) rows_s[tid] = (!extra && slot_mask != 0) ? (slot_v | (1 << 30)) : slot_v;
    Drop sdrop = src.drop;
    if (src_on) {
        if (sdrop.p > 0.f) {
            sdrop.seed = A.rng_state[0];
            sdrop.step = A.rng_state[1];
        }
        gn_bwd_coef_nobarrier(src.acc, src.n_rep, N, src.saved, src.gamma, src.alpha, src.dgamma, src.dbeta, src.dalpha,
                              src.accumulate, blk == 0, coef_s);
        lds_barrier();  // coefficients before the first stage is prepared
    }
    // stage -> LDS: the loader's four float4 in the layouts their readers want
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        Stage& S = stg[st & 1];
        const int r = my_row[st];
        float4 dcv = R.dc;
        if (src_on) {
            float sds[4] = {1.f, 1.f, 1.f, 1.f};
            if (sdrop.p > 0.f) drop_scales<4>(sdrop, r < 0 ? 0 : r, 4 * ga, sds);
            dcv = gn_bwd_apply4(R.dc, R.sx, R.sad, coef_s, 4 * ga, src.act, sds);
            if (r < 0) dcv = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (SP) {
            const Split4 sc = split4(dcv);
            unsigned short* P = reinterpret_cast<unsigned short*>(S.dcP) + rs * RSB + 4 * ga;
            *reinterpret_cast<uint2*>(P) = sc.hi;
            *reinterpret_cast<uint2*>(P + kPlane) = sc.mid;
            *reinterpret_cast<uint2*>(P + 2 * kPlane) = sc.lo;
        } else {
            *reinterpret_cast<float4*>(S.dcP + rs * RS + 4 * ga) = dcv;
        }
        const float av[4] = {R.a.x, R.a.y, R.a.z, R.a.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_scales<4>(drop, r < 0 ? 0 : r, 4 * ga, ds);
        float u[4], xu[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u[k] = ds[k];
            xu[k] = (av[k] - g_al[k] * g_mu[k]) * g_rs[k] * ds[k];
        }
        *reinterpret_cast<float4*>(S.U + rs * RS + 4 * ga) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4*>(S.XU + rs * RS + 4 * ga) = make_float4(xu[0], xu[1], xu[2], xu[3]);
    };
    commit(0, rawA);
    issue(2, rawA);
    D_STAMP(3, 1);
    lds_barrier();
    float s1 = 0.f, s2 = 0.f;     // GraphNorm backward sums of column 16w + j over this lane's rows
    const int cg = 16 * w + j;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const Stage& S = stg[st & 1];
        float4 a4[SP ? 1 : 4];
        uint4 af[SP ? 2 : 1][3];
        if constexpr (SP) {
            const unsigned short* P = reinterpret_cast<const unsigned short*>(S.dcP) + j * RSB + 16 * q;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) af[b][pc] = *reinterpret_cast<const uint4*>(P + pc * kPlane + 8 * b);
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) a4[v] = *reinterpret_cast<const float4*>(S.dcP + j * RS + 16 * q + 4 * v);
        }
        int rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = rows_s[16 * st + 4 * q + r];
        float uu[4], xx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uu[r] = S.U[(4 * q + r) * RS + cg];
            xx[r] = S.XU[(4 * q + r) * RS + cg];
        }
        f32x4 accg = {0.f, 0.f, 0.f, 0.f}, accx = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SP) {
#define GLASS_SMMA16(ACC, BW, pa, pb)                                                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[b][pa]), __builtin_bit_cast(bf16x8, BW[b][pb]), ACC, 0, 0, 0)
#pragma unroll
            for (int b = 0; b < 2; ++b) {  // small terms first; the two halves' chains interleave
                GLASS_SMMA16(accg, bwgc, 1, 1); GLASS_SMMA16(accx, bwxc, 1, 1);
                GLASS_SMMA16(accg, bwgc, 2, 0); GLASS_SMMA16(accx, bwxc, 2, 0);
                GLASS_SMMA16(accg, bwgc, 0, 2); GLASS_SMMA16(accx, bwxc, 0, 2);
                GLASS_SMMA16(accg, bwgc, 1, 0); GLASS_SMMA16(accx, bwxc, 1, 0);
                GLASS_SMMA16(accg, bwgc, 0, 1); GLASS_SMMA16(accx, bwxc, 0, 1);
                GLASS_SMMA16(accg, bwgc, 0, 0); GLASS_SMMA16(accx, bwxc, 0, 0);
            }
#undef GLASS_SMMA16
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float x[4] = {a4[v].x, a4[v].y, a4[v].z, a4[v].w};
                const float yg[4] = {bwg[v].x, bwg[v].y, bwg[v].z, bwg[v].w}, yx[4] = {bwx[v].x, bwx[v].y, bwx[v].z, bwx[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!GLASS_MFMA_KEEP(v * 4 + e)) continue;
                    accg = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], yg[e], accg, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], yx[e], accx, 0, 0, 0);
                }
            }
        }
        // next stage -> LDS (the other buffer: its last readers passed the barrier at the end of the previous iteration)
        if (st + 1 < 4) {
            commit(st + 1, (st & 1) ? rawA : rawB);
            if (st + 3 < 4) {
                if (st & 1) issue(st + 3, rawA); else issue(st + 3, rawB);
            }
        }
        // data-gradient epilogue: rows 4q + r, columns cg (dg) and H + cg (dx_)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = rv[r] >= 0 && !(rv[r] >> 30);
            const int row = rv[r] & ((1 << 30) - 1);
            buf_store1(r_out, live ? (int)((row * A.ldo + cg) * 4) : kBufOOB, accg[r]);
            buf_store1(r_out, live ? (int)((row * A.ldo + H + cg) * 4) : kBufOOB, accx[r]);
            const float gp = live ? accg[r] * uu[r] : 0.f;
            s1 += gp;
            s2 = fmaf(live ? accg[r] : 0.f, xx[r], s2);
        }
        if (st + 1 < 4) lds_barrier();
    }
    D_STAMP(3, 3);
    if (gs.partial) {
        double a = (double)s1, b2 = (double)s2;
        a += __shfl_xor(a, 16);
        b2 += __shfl_xor(b2, 16);
        a += __shfl_xor(a, 32);
        b2 += __shfl_xor(b2, 32);
        if (q == 0) {
            if (gs.exact) {
                gn_acc_add(reinterpret_cast<long long*>(gs.partial), blk % gs.exact, 0, cg, H, a, kAccScaleBwd);
                gn_acc_add(reinterpret_cast<long long*>(gs.partial), blk % gs.exact, 1, cg, H, b2, kAccScaleBwd);
            } else {
                gs.partial[((size_t)blk * 2) * H + cg] = a;
                gs.partial[((size_t)blk * 2 + 1) * H + cg] = b2;
            }
        }
    }
    D_STAMP(3, 4);
}
constexpr size_t kCombDgrad2Lds = (size_t)(2 * 3 * 16 * (64 + 4) + 64 + 5 * 64) * sizeof(float);
constexpr size_t kCombDgrad2SplitLds = (size_t)(2 * ((3 * 16 * (64 + 8)) / 2 + 2 * 16 * (64 + 4)) + 64 + 5 * 64) * sizeof(float);  // SP

#if GLASS_LAB
#include "../../tools/lab/comb_bwd_eff2.inc"  // one-pass staged comb backward (GLASS_COMB_BWD_V2; measured slower): laboratory builds only
#endif

// Fused backward launch of the comb pair in effective-weight form: data-gradient row tiles (main + extra), then the
// weight-gradient blocks in S / L form (wgrad_sl_body: row slabs, then the labeled-row tiles).
template <int H, bool SP = false>
__global__ __launch_bounds__(kBlock, 2) void comb_bwd_eff_kernel(DgradEffArgs A, int n_dgrad_blocks, WgradSL sl, float zr,
                                                                float* __restrict__ part_w, float* __restrict__ part_b) {
    extern __shared__ float4 lds_w[];
    // the weight-gradient workgroups are the long ones (phase stamps: 12.6 / 16.2 us of life against 7.8 / 11.1): they take the
    // FIRST workgroup ids, so the dispatcher starts them first; b = the id in the old order (data-gradient tiles first)
    const int n_wg_blocks = (int)gridDim.x - n_dgrad_blocks;
    const int b = (int)blockIdx.x < n_wg_blocks ? n_dgrad_blocks + (int)blockIdx.x : (int)blockIdx.x - n_wg_blocks;
    if (b == 0 && threadIdx.x == 0) {  // mode header behind the bias partials, read by the (deferred) reduce launch
        float* header = part_b + (int64_t)(sl.n_s + sl.n_l) * kSLOut;
        header[0] = 2.f;
        header[1] = zr;
        header[2] = GLASS_SL_STAGED2 ? 1.f : 0.f;  // tiles in plain [o][i] order (staged body) / in the permuted accumulator order
    }
    if (b < n_dgrad_blocks) {
        D_STAMP(3, 0);
        if (GLASS_COMB_DGRAD_V2) {  // staged form; its images (layout 10) lie behind the layout-7 pair
            DgradEffArgs A2 = A;
            A2.WT = A.WT + 2 * (2 * H * H);
            if (A.gs.partial && A.gs.drop.p > 0.f)
                comb_dgrad2_body<H, true, SP>(A2, b, reinterpret_cast<float*>(lds_w));
            else
                comb_dgrad2_body<H, false, SP>(A2, b, reinterpret_cast<float*>(lds_w));
        } else {
            comb_dgrad_eff_body<H, 4>(A.dsrc, A.ldd, A.mask, A.WT, A.rng_state, A.out, A.ldo, A.N, A.gs, A.lab, b, lds_w);
        }
        D_STAMP(3, 4);
        return;
    }
    D_STAMP(3, 5);
    float* lds = reinterpret_cast<float*>(lds_w);
    if (GLASS_SL_STAGED2) {
        if constexpr (SP)
            wgrad_sl_staged2s_body(sl, A.N, b - n_dgrad_blocks, part_w, part_b, lds);
        else
            wgrad_sl_staged2_body(sl, A.N, b - n_dgrad_blocks, part_w, part_b, lds);
        D_STAMP(3, 6);
        return;
    }
    if (GLASS_WGRAD_STAGED && sl.rows_per_slab <= kStageRows)  // small graph: the whole slab through LDS, one memory round trip
        wgrad_sl_staged_body(sl, A.N, b - n_dgrad_blocks, part_w, part_b, lds, lds + 2 * kTile);
    else
        wgrad_sl_body<GLASS_FUSED_WGRAD_STAGES>(sl, A.N, b - n_dgrad_blocks, part_w, part_b, lds, lds + 2 * kTile);
    D_STAMP(3, 6);
}

// ---- packing of the stacked weights into MFMA images (one launch for the whole model, once per step) ------
// job: logical operand B[NT][KT] (row = output column of the product, col = k).  transposed == 0: B = src
// ([NT][KT] row-major, the forward weight [2H][K]); transposed == 1: B[n][k] = src[k][n] with src [KT][NT]
// row-major (the data-gradient operand W^T of a weight stored [2H][n_out]).
// dst[((kc*NTILES + t)*4 + v)*64 + lane] (float4) = B[tile_col(t, j)][q*KT/4 + kc*16 + 4v .. +3], lane = j + 16q.
struct PackJob {
    const float* src;
    float* dst;
    int NT, KT, transposed, layout;
    float zr;  // kLayoutTiledPlainEff: the pair's z_ratio
    int cut;   // tiled layouts: also write the image cut into bf16 pieces behind the fp32 one (split_mma.h, pack_cut_put)
};
constexpr int kMaxPackJobs = 16;
struct PackBatch {
    PackJob job[kMaxPackJobs];
};

__device__ __forceinline__ float4 pack_fetch(const PackJob& j, int n, int k) {
    if (!j.transposed) return *reinterpret_cast<const float4*>(j.src + (int64_t)n * j.KT + k);
    return make_float4(j.src[(int64_t)k * j.NT + n], j.src[(int64_t)(k + 1) * j.NT + n], j.src[(int64_t)(k + 2) * j.NT + n],
                       j.src[(int64_t)(k + 3) * j.NT + n]);
}

// The cut image of a tiled layout (behind the fp32 image and its appendix, same tile numbering): per tile of 16 k x 256 slots
// [piece 3][h 2][slot 256] 16-byte units = the LDS image of split_mma.h's SplitImg<256>, so that a K step's B operand is 24 KiB
// copied as it lies (global_load_lds in the tiled kernels).  Element (slot nl, k-quad q): 8 bytes per piece.
__device__ __forceinline__ void pack_cut_put(float* cut, int tile, int nl, int q, const float4& v) {
    const Split4 s = split4(v);
    uint2* p = reinterpret_cast<uint2*>(cut) + (((size_t)tile * 6 + (q >> 1)) * 256 + nl) * 2 + (q & 1);
    p[0] = s.hi;
    p[1024] = s.mid;
    p[2048] = s.lo;
}

// The other once-per-step prologue job rides in the same launch (grid row n_jobs): emb_gn's statistics through the
// embedding table (embnorm.hip, emb_table.h) — it only depends on the parameters, like the packing.
struct TableJob {
    const float* W;  // nullptr: none
    int V, H;
    const int32_t* rowptr;
    const float *gamma, *beta, *alpha;
    float eps;
    float *saved, *table;
};

// ... and so does the zero-fill of the step's exact GraphNorm accumulators (gn_acc.h), spread over every workgroup
struct ZeroJob {
    long long* p;  // nullptr: none
    int64_t n;     // int64 words
};

// (a device function: the stand-alone prologue launch below, and the first grid rows of the step's head launch; `rows` = the
// grid rows that run it — the zero-fill is spread over exactly those; kBlock threads per workgroup take part)
__device__ __forceinline__ void pack_batch_body(const PackBatch& batch, uint64_t* rng_state, const TableJob& tab, int n_jobs,
                                                const ZeroJob& zero, int rows) {
    if (rng_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) rng_state[1] += 1;  // see glass_rng_advance
    if (zero.p) {
        const int64_t nthreads = (int64_t)gridDim.x * rows * kBlock;
        for (int64_t k = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kBlock + threadIdx.x; k < zero.n; k += nthreads) zero.p[k] = 0;
    }
    if ((int)blockIdx.y >= n_jobs) {
        __shared__ double tab_lds[kBlock * 2];
        __shared__ float tab_coef[2 * kTabCols];
        if (tab.W && (int)blockIdx.x < (tab.H + kTabCols - 1) / kTabCols)
            emb_table_fwd_block(blockIdx.x, tab.W, tab.V, tab.H, tab.rowptr, tab.gamma, tab.beta, tab.alpha, tab.eps, tab.saved,
                                tab.table, tab_lds, tab_coef);
        return;
    }
    const PackJob j = batch.job[blockIdx.y];
    const int total = j.NT * j.KT / 4;  // float4 elements
    if (j.layout == kLayoutWave16 || j.layout == kLayoutWave16Cols) {
        const int KQ = j.KT / 4, NTILES = j.NT / 16;
        const bool cols = j.layout == kLayoutWave16Cols;
        for (int l = blockIdx.x * kBlock + threadIdx.x; l < total; l += gridDim.x * kBlock) {
            const int lane = l & 63, v = (l >> 6) & 3, t = (l >> 8) % NTILES, kc = (l >> 8) / NTILES;
            const int jj = lane & 15, q = lane >> 4;
            const int n = cols ? 64 * (t >> 2) + 16 * (t & 3) + jj : tile_col(t, jj);
            reinterpret_cast<float4*>(j.dst)[l] = pack_fetch(j, n, q * KQ + kc * kKC + 4 * v);
        }
        return;
    }
    if (j.layout == kLayoutWave16EffFwd || j.layout == kLayoutWave16EffDgrad || j.layout == kLayoutWave16EffFwdCols ||
        j.layout == kLayoutWave16EffDgradCols) {
        // comb pair, hidden 64: two wave16 images of the effective weights c1 * (f1 half) + c0 * (f0 half) — unlabeled rows
        // (c1, c0) = (1-z, z), then labeled rows (z, 1-z).  Forward: the halves are the two row blocks of B (NT/2 outputs
        // each); data gradient (transposed source): the two halves of k (the stacked output index of the pair).
        const bool fwd = j.layout == kLayoutWave16EffFwd || j.layout == kLayoutWave16EffFwdCols;
        // "Cols" forms: tile t = columns 64 (t >> 2) + 16 (t & 3) .. + 15 (the staged kernels: a wave owns 16 consecutive columns)
        const bool cols = j.layout == kLayoutWave16EffFwdCols || j.layout == kLayoutWave16EffDgradCols;
        const int NTe = fwd ? j.NT / 2 : j.NT, KTe = fwd ? j.KT : j.KT / 2;
        const int KQ = KTe / 4, NTILES = NTe / 16, per = NTe * KTe / 4;
        const float zr = j.zr, omz = 1.f - j.zr;
        for (int l = blockIdx.x * kBlock + threadIdx.x; l < 2 * per; l += gridDim.x * kBlock) {
            const int img = l >= per, ll = img ? l - per : l;
            const int lane = ll & 63, v = (ll >> 6) & 3, t = (ll >> 8) % NTILES, kc = (ll >> 8) / NTILES;
            const int jj = lane & 15, q = lane >> 4;
            const int n = cols ? 64 * (t >> 2) + 16 * (t & 3) + jj : tile_col(t, jj), k = q * KQ + kc * kKC + 4 * v;
            const float4 a = pack_fetch(j, n, k);
            const float4 b = fwd ? pack_fetch(j, j.NT / 2 + n, k) : pack_fetch(j, n, j.KT / 2 + k);
            const float c1 = img ? zr : omz, c0 = img ? omz : zr;
            reinterpret_cast<float4*>(j.dst)[l] = make_float4(c1 * a.x + c0 * b.x, c1 * a.y + c0 * b.y, c1 * a.z + c0 * b.z,
                                                              c1 * a.w + c0 * b.w);
        }
        return;
    }
    if (j.layout == kLayoutTiledSplit) {
        // split data-gradient operand (hidden 128, trans pair: NT = 128 outputs, KT = 256 = the two stacked halves): one
        // 256-slot column tile holds BOTH halves side by side over K = KT / 2 — slot v < NT: B[v][k], else B[v - NT][KT/2 + k]
        const int NKS = j.KT / 32;
        for (int l = blockIdx.x * kBlock + threadIdx.x; l < total; l += gridDim.x * kBlock) {
            const int nl = l & 255, q = (l >> 8) & 3, ks = l >> 10;
            const int v = tiled_col(kLayoutTiledPlain, 0, nl, 0);
            if (ks < NKS) {
                const float4 e = pack_fetch(j, v % j.NT, (v / j.NT) * (j.KT / 2) + 16 * ks + 4 * q);
                reinterpret_cast<float4*>(j.dst)[l] = e;
                if (j.cut) pack_cut_put(j.dst + 4 * (size_t)total, ks, nl, q, e);
            }
        }
        return;
    }
    // tiled layouts (dense_tiled.hip): dst[((ct * NKS + ks) * 4 + q) * 256 + nl] (float4) = B[tiled_col(ct, nl)][16 ks + 4 q ..+3]
    const int NKS = j.KT / 16, H = j.NT / 2;  // H only meaningful for the paired layout (NT = 2H)
    const int lay = j.layout == kLayoutTiledPlainEff ? kLayoutTiledPlain : j.layout == kLayoutTiledPairedEff ? kLayoutTiledPaired : j.layout;
    const bool has_app = j.layout == kLayoutTiledPlainEff || j.layout == kLayoutTiledPairedEff;
    float* cut = j.dst + 4 * (size_t)total + (has_app ? 2 * (size_t)total : 0);  // behind the fp32 image and its appendix
    for (int l = blockIdx.x * kBlock + threadIdx.x; l < total; l += gridDim.x * kBlock) {
        const int nl = l & 255, q = (l >> 8) & 3, tile = l >> 10;
        const int ct = tile / NKS, ks = tile % NKS;
        const float4 e = pack_fetch(j, tiled_col(lay, ct, nl, H), 16 * ks + 4 * q);
        reinterpret_cast<float4*>(j.dst)[l] = e;
        if (j.cut) pack_cut_put(cut, tile, nl, q, e);
    }
    if (j.layout == kLayoutTiledPairedEff) {
        // appendix of the forward operand: W_unl[n][k] = (1 - z) * B[n][k] + z * B[NT/2 + n][k], n < NT/2 (an unlabeled
        // row weighs the f1 half with 1 - z and the f0 half with z), plain tiling: 256 output columns per column tile
        float4* app = reinterpret_cast<float4*>(j.dst) + total;
        const float zr = j.zr, omz = 1.f - j.zr;
        for (int l = blockIdx.x * kBlock + threadIdx.x; l < total / 2; l += gridDim.x * kBlock) {
            const int nl = l & 255, q = (l >> 8) & 3, tile = l >> 10;
            const int ct = tile / NKS, ks = tile % NKS;
            const int n = tiled_col(kLayoutTiledPlain, ct, nl, H), k = 16 * ks + 4 * q;
            const float4 a = pack_fetch(j, n, k), b = pack_fetch(j, H + n, k);
            const float4 e = make_float4(omz * a.x + zr * b.x, omz * a.y + zr * b.y, omz * a.z + zr * b.z, omz * a.w + zr * b.w);
            app[l] = e;
            if (j.cut) pack_cut_put(cut, total / 1024 + tile, nl, q, e);
        }
        return;
    }
    if (j.layout != kLayoutTiledPlainEff) return;
    // appendix: B_unl[n][k] = (1 - z) * B[n][k] + z * B[n][KT/2 + k], k < KT/2 (k <-> the f1 / f0 halves of the stacked
    // output index: an unlabeled row weighs f1 with 1 - z and f0 with z), same plain tiling over K' = KT / 2
    const int NKS2 = NKS / 2;
    float4* app = reinterpret_cast<float4*>(j.dst) + total;
    const float zr = j.zr, omz = 1.f - j.zr;
    for (int l = blockIdx.x * kBlock + threadIdx.x; l < total / 2; l += gridDim.x * kBlock) {
        const int nl = l & 255, q = (l >> 8) & 3, tile = l >> 10;
        const int ct = tile / NKS2, ks = tile % NKS2;
        const int n = tiled_col(kLayoutTiledPlain, ct, nl, H), k = 16 * ks + 4 * q;
        const float4 a = pack_fetch(j, n, k), b = pack_fetch(j, n, j.KT / 2 + k);
        const float4 e = make_float4(omz * a.x + zr * b.x, omz * a.y + zr * b.y, omz * a.z + zr * b.z, omz * a.w + zr * b.w);
        app[l] = e;
        if (j.cut) pack_cut_put(cut, total / 1024 + tile, nl, q, e);
    }
}

__global__ __launch_bounds__(kBlock) void pack_batch_kernel(PackBatch batch, uint64_t* rng_state, TableJob tab, int n_jobs,
                                                            ZeroJob zero) {
    pack_batch_body(batch, rng_state, tab, n_jobs, zero, (int)gridDim.y);
}

// ---- the head of a replayed step: prologue || labels as ONE launch (glass_step_head_f32) ------------------------------
// The prologue (weight images, table statistics, accumulator zero-fill: depends on the parameters only) and the label
// launch (depends on the batch only) are the two independent kernels at the head of every step, ~4.5 and ~6.5 us each at
// ppi_bp-shape, and each is a chain of dependent round trips that leaves the chip idle.  Side by side in one grid the
// shorter one disappears from the step's chain.  Grid rows [0, rows): the prologue's (its kBlock-thread roles on the first
// kBlock threads of a kLabThreads-wide workgroup; the other waves end at once — a barrier never counts ended waves); row
// `rows`, workgroup 0: the label workgroup (labels_body.h).  Which batch: a device-resident cursor (glass_batch_cursor) —
// a captured launch cannot take a new pointer per replay.
struct LabelJob {
    glass_batch_cursor* cur;
    int n_idx, smax, y_row_words;
    int64_t* pos_dst;
    uint32_t* y_dst;
    uint8_t* mask;
    int32_t *lab_rows, *lab_count, *owner;
    int64_t N;
};

__global__ __launch_bounds__(kLabThreads) void step_head_kernel(PackBatch batch, uint64_t* rng_state, TableJob tab, int n_jobs,
                                                                ZeroJob zero, int rows, LabelJob lj) {
    if ((int)blockIdx.y < rows) {
        if (threadIdx.x >= kBlock) return;
        pack_batch_body(batch, rng_state, tab, n_jobs, zero, rows);
        return;
    }
    if (blockIdx.x != 0) return;
    // (wave-uniform scalar loads: one dependent round trip in front of the body's own)
    const int64_t n_b = lj.cur->n_batches, c = lj.cur->cursor;
    const int64_t b = c < 0 ? 0 : (lj.cur->wrap ? c % n_b : (c >= n_b ? n_b - 1 : c));
    const BatchSrc src{lj.cur->pos_all, reinterpret_cast<const uint32_t*>(lj.cur->y_all), lj.cur->idx + b * lj.n_idx, lj.smax,
                       lj.y_row_words > 0 ? lj.y_row_words : 1, lj.cur->n_all};
    batch_labels_body<true>(src, lj.n_idx * lj.smax, lj.pos_dst, lj.y_dst, (int64_t)lj.n_idx * lj.y_row_words, lj.mask, lj.lab_rows,
                            lj.lab_count, lj.owner, lj.N, 1);
    if (threadIdx.x == 0) lj.cur->cursor = c + 1;  // (every read of the cursor lies before the body's barriers)
}

}  // namespace glass

using namespace glass;

// Dynamic LDS above 64 KiB must be allowed per kernel once.
template <typename K>
static void allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static bool wave16_shape_ok(int64_t H) { return H == 64; }
// Wave groups of the staged hidden-64 forward kernels (trans_fwd2_kernel / comb_fwd_eff2_kernel): 2 = two waves per SIMD.
// Measured at ppi_bp-shape (round 4, same box, alternating runs): trans forward 13.2 -> 11.8 us per launch with two groups
// (three 32-row stages instead of five 16-row ones); the comb forward does not move (12.1 vs 12.0: its stage is 17 KB of
// LDS traffic for the same 32 MFMAs per wave, and eight waves fetch the weight slices twice) — so trans takes 2, comb 1.
// GLASS_FWD_WG=1|2 forces both (laboratory A/B; read once per process).
// hidden 64: the staged kernels' products in the split form of the tiled family (split_mma.h) — per call, like there
static bool h64_split_products() { return tiled_split_products() && lab_knob("GLASS_H64_SPLIT", 1) != 0; }
static int fwd_wave_groups(bool comb) {
    const int forced = lab_knob("GLASS_FWD_WG", 0);
    if (forced == 1 || forced == 2) return forced;
    return comb ? 1 : 2;
}
// Above this many rows the two halves of the backward of a pair fill the chip on their own (one launch each, the
// weight gradient with its 4-stage pipeline); below, they run as two branches of one launch (dual_bwd_kernel).
static bool tiled_here(int64_t H) { return tiled_shape_ok(H) && !wave16_shape_ok(H); }
static bool dense_shape_ok(int64_t H) { return wave16_shape_ok(H) || tiled_shape_ok(H) || narrow_shape_ok(H); }
static size_t lds_bytes(int64_t NT, int n_pass) {  // weight images resident at once: two when K needs more than one pass
    const size_t image = (size_t)NT * 256;
    return n_pass > 1 ? 2 * image : image;
}

// Policy: hidden 64 on the wave-owns-16-rows kernels of this file (one wave group per 64 rows); hidden 128 / 256 / 512 on
// the LDS-tiled kernels of dense_tiled.hip (hidden 128, N = 50 000 against round 1's two-wave-group kernels of this file:
// forward 63 -> 53 / 95 -> 85 us, data gradient 79 -> 56 (trans, split tile) / 98 -> 80 us (comb), em_user-shape step
// 0.82 -> 0.70 ms).  The CS > 1 form of the kernels below (wave groups splitting the output columns) is no longer
// instantiated.
// (A/B switches live in the Python layer, glass_amd/ops.py: the library keeps no state.)
// exact cross-workgroup GraphNorm sums (gn_acc.h) instead of per-workgroup partials + a finalize launch: the hidden-64 kernels
#ifdef GLASS_DENSE_TRACE
extern "C" int glass_dense_trace_set(unsigned long long* p, int sel) {
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dense_trace), &p, sizeof(p));
    return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dense_trace_sel), &sel, sizeof(sel));
}
#endif
extern "C" int glass_gn_exact_supported(int64_t H) { return wave16_shape_ok(H) ? 1 : 0; }
// ... the FORWARD sums alone (statistics kernel -> staged comb forward -> readout): also hidden 128
extern "C" int glass_gn_exact_fwd_supported(int64_t H) { return (wave16_shape_ok(H) || (GLASS_COMB_FWD_V2 && H == 128)) ? 1 : 0; }
// gn_src (the C-ABI struct) -> the kernels' form; saved = the [4C] buffer workgroup 0 writes.  NULL gn_src: final statistics.
static bool rep_ok(int64_t n_rep) { return n_rep == 2 || n_rep == 4 || n_rep == 8 || n_rep == 16; }  // <= kAccRep, even
static bool make_exact_src(const glass_gn_src* g, const float* saved, GnExactSrc& out) {
    out = GnExactSrc{nullptr, 1, kAccRep, nullptr, nullptr, nullptr, 0.f, nullptr};
    if (!g) return true;
    if (!g->acc || g->n_src != 1 || !rep_ok(g->n_rep) || !g->gamma || !g->beta || !g->alpha || !saved) return false;
    out = GnExactSrc{reinterpret_cast<const long long*>(g->acc), 1, (int)g->n_rep, g->gamma, g->beta, g->alpha, g->eps,
                     const_cast<float*>(saved)};
    return true;
}
extern "C" int64_t glass_gn_exact_words(int64_t C) { return gn_acc_words(C); }  // int64 words of one accumulator block

extern "C" int glass_dual_linear_supported(int64_t H) { return dense_shape_ok(H) ? 1 : 0; }

extern "C" int glass_pair_head_supported(int64_t hidden);
extern "C" int glass_dense_caps_query(int64_t H, glass_dense_caps* out) {
    GLASS_REQUIRE(H > 0 && out, "dense_caps: H > 0 and a record to fill");
    glass_dense_caps c{};
    c.family = narrow_shape_ok(H) ? 1 : wave16_shape_ok(H) ? 2 : tiled_here(H) ? 3 : 0;
    if (c.family) {
        c.weight_layout = glass_dual_linear_layout(H);
        c.fwd_layout_trans = glass_dual_linear_fwd_layout(H, H);
        c.fwd_layout_comb = glass_dual_linear_fwd_layout(H, 2 * H);
        c.dgrad_layout_trans = glass_dual_linear_dgrad_layout(H, H);
        c.dgrad_layout_comb = glass_dual_linear_dgrad_layout(H, 2 * H);
        c.stat_rows = (int32_t)glass_dual_linear_stat_rows(H);
        c.fwd_gather = glass_dual_linear_fwd_gather_supported(H);
        c.act_codes = (1 << GLASS_ACT_ELU) | (1 << GLASS_ACT_RELU);
        // the family's default; GLASS_DENSE_F32_PRODUCTS in a call's act word opts out (round 6: the staged hidden-64 kernels too)
        c.product_form = (c.family == 3 || (c.family == 2 && lab_knob("GLASS_H64_SPLIT", 1) != 0)) ? 1 : 0;
    }
    c.serve_width = c.family ? (int32_t)H : (H <= 64 ? 64 : H <= 128 ? 128 : H <= 256 ? 256 : H <= 512 ? 512 : 0);
    c.gn_exact = glass_gn_exact_supported(H);
    c.gn_exact_fwd = glass_gn_exact_fwd_supported(H);
    c.comb_eff = glass_comb_eff_supported(H);
    c.comb_eff_fwd = glass_comb_eff_fwd_supported(H);
    c.comb_eff_fwd_layout = c.comb_eff_fwd ? glass_comb_eff_fwd_layout(H) : 0;
    c.comb_eff_dgrad_layout2 = c.comb_eff ? glass_comb_eff_dgrad_layout2(H) : 0;
    c.pair_head = glass_pair_head_supported(H);
    *out = c;
    return 0;
}
// How the LDS-tiled family (hidden 128 / 256 / 512) forms its fp32 products: 1 = six bf16 partial products of 3-way split
// operands (split_mma.h: as accurate as the f32-input instruction, 6/16 of its matrix-core cycles), 0 = v_mfma_f32_32x32x2_f32.
// glass_dual_linear_fwd_f32 with xa_index (the trans pair of layer 0 gathers its operand rows from the embedding table)
extern "C" int glass_dual_linear_fwd_gather_supported(int64_t H) { return (wave16_shape_ok(H) || narrow_shape_ok(H) || tiled_here(H)) ? 1 : 0; }

// Operand-image layout glass_dense_pack_batch_f32 must produce for hidden size H: 0 = wave16 images (forward and data
// gradient alike), 1 = tiled (forward operand: paired layout; data-gradient operand: plain layout)
extern "C" int glass_dual_linear_layout(int64_t H) { return narrow_shape_ok(H) ? 2 : tiled_here(H) ? 1 : 0; }

// Layout code (flags >> 1 of glass_dense_pack_batch_f32) of the DATA-GRADIENT operand image glass_dual_linear_dgrad_f32 /
// _bwd_f32 read for (H, n_out): 0 wave16, 2 plain, 3 split (hidden 128, 128-wide output), 4 plain + effective-weight
// appendix (comb pair: the image holds 1.5 x the weight's floats).
// ... and of the FORWARD operand image for (H, K = input width): 0 wave16, 1 paired, 5 paired + effective-weight appendix
// (comb pair, K = 2H, at hidden 256 / 512: 1.5 x the weight's floats)
extern "C" int glass_dual_linear_fwd_layout(int64_t H, int64_t K) {
    if (GLASS_TRANS_FWD_V2 && (wave16_shape_ok(H) || H == 128) && K == H) return kLayoutWave16Cols;  // trans pair at hidden 64 / 128: trans_fwd2_kernel / trans_fwd3_kernel
    if (!tiled_here(H)) return kLayoutWave16;
    return tiled_eff_fwd_shape(H, K) ? kLayoutTiledPairedEff : kLayoutTiledPaired;
}
extern "C" int glass_dual_linear_dgrad_layout(int64_t H, int64_t n_out) {
    if (GLASS_TRANS_DGRAD_V2 && (wave16_shape_ok(H) || H == 128) && n_out == H) return kLayoutWave16Cols;  // trans pair at hidden 64 / 128: trans_dgrad2_body
    if (!tiled_here(H)) return kLayoutWave16;
    if (GLASS_COMB_DGRAD_V2 && H == 128 && n_out == 2 * H) return kLayoutWave16EffDgradCols;  // comb pair at hidden 128: comb_dgrad3_kernel
    if (tiled_eff_dgrad_shape(H, n_out)) return kLayoutTiledPlainEff;
    return n_out % 256 == 0 ? kLayoutTiledPlain : kLayoutTiledSplit;
}

// rows per workgroup = rows per epilogue statistics partial
extern "C" int64_t glass_dual_linear_stat_rows(int64_t H) { return narrow_shape_ok(H) ? narrow_rows() : tiled_here(H) ? tiled_rows(H) : 64; }

extern "C" int glass_dual_linear_fwd_f32(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* W,
                                         const float* bias, const uint8_t* mask, double z_ratio, int act, float* T,
                                         int64_t ldt, float* out, int64_t ldo, int64_t n_nodes, int64_t H,
                                         double* stats, int stats_exact, const float* gn_saved, const glass_gn_src* gn_src,
                                         int gn_act, float p_drop, const uint64_t* rng_state, uint64_t call_id, float* xa_out,
                                         int64_t ldxo, const int64_t* xa_index, int64_t xa_rows, void* stream) {
    const CallOptions call_options(act);  // act word -> activation code + this call's options
    GLASS_REQUIRE(xa && W && bias && mask && out && n_nodes > 0, "dual_linear_fwd: null pointer");
    GLASS_REQUIRE((!stats_exact || (stats && wave16_shape_ok(H) && rep_ok(stats_exact))) && (!gn_src || (gn_saved && wave16_shape_ok(H))),
                  "dual_linear_fwd: exact GraphNorm accumulators are served at hidden 64 only (glass_gn_exact_supported)");
    GnExactSrc esrc;
    GLASS_REQUIRE(make_exact_src(gn_src, gn_saved, esrc), "dual_linear_fwd: bad gn_src (one accumulator block, all pointers set)");
    GLASS_REQUIRE(!xa_index || (gn_saved && !xb && xa_rows > 0 && xa_rows < (1ll << 31)),
                  "dual_linear_fwd: a gathered operand needs the GraphNorm prologue (its side output is the gathered, "
                  "normalised [N,H] input) and is the trans pair's");
    if (xa_index && !wave16_shape_ok(H) && !narrow_shape_ok(H) && !tiled_here(H)) {
        set_error("dual_linear_fwd: gathered operand not served at this hidden size (glass_dual_linear_fwd_gather_supported)");
        return GLASS_E_UNSUPPORTED;
    }
    GLASS_REQUIRE(!gn_saved || (xa_out && ldxo >= H && (narrow_shape_ok(H) || (ldxo % 4 == 0 && aligned16(xa_out) && aligned16(gn_saved))) &&
                                p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state) &&
                                act_code_ok(gn_act)),
                  "dual_linear_fwd: bad GraphNorm prologue arguments");
    if (!dense_shape_ok(H)) {
        set_error("dual_linear_fwd: hidden size %lld not supported (<= 32, 64, 128, 256, 512)", (long long)H);
        return GLASS_E_UNSUPPORTED;
    }
    if (narrow_shape_ok(H)) {  // thread-per-row kernels: any alignment, the weight as it is (no packed image)
        GLASS_REQUIRE(lda >= H && (!xb || ldb >= H) && ldo >= H && (!T || ldt >= 2 * H) && (!gn_saved || ldxo >= H),
                      "dual_linear_fwd: leading dimensions");
        const GnPrologue npro{gn_saved, (int)H, gn_act, make_drop(gn_saved ? p_drop : 0.f, call_id, H), rng_state, xa_out, ldxo, esrc};
        return launch_narrow_fwd(xa, lda, xb, ldb, W, bias, mask, (float)z_ratio, (float)(1.0 - z_ratio), act, T, ldt, out, ldo,
                                 n_nodes, H, stats, npro, xa_index, xa_rows, (hipStream_t)stream);
    }
    const bool comb = xb != nullptr;
    GLASS_REQUIRE(lda >= H && lda % 4 == 0 && aligned16(xa) && (!comb || (ldb >= H && ldb % 4 == 0 && aligned16(xb))) &&
                      aligned16(W) && aligned16(bias) && ldo >= H && ldo % 4 == 0 && aligned16(out) &&
                      (!T || (ldt >= 2 * H && ldt % 4 == 0 && aligned16(T))),
                  "dual_linear_fwd: operands must be 16-B aligned with ld %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(n_nodes, glass_dual_linear_stat_rows(H)));
    const float zr = (float)z_ratio, omz = (float)(1.0 - z_ratio);
    // dynamic LDS: one weight image per K-pass in flight (NT*256 bytes each; two when K needs > 1 pass and both fit)
    const size_t lds_comb = lds_bytes(2 * H, (int)(2 * H / 64)), lds_trans = lds_bytes(2 * H, (int)(H / 64));
    const GnPrologue pro{gn_saved, (int)H, gn_act, make_drop(gn_saved ? p_drop : 0.f, call_id, H), rng_state, xa_out, ldxo, esrc};
    if (GLASS_TRANS_FWD_V2 && !comb && H == 128) {  // stage-run form (image in layout kLayoutWave16Cols: glass_dual_linear_fwd_layout)
        const int64_t src_rows = xa_index ? xa_rows : n_nodes;
        const int64_t ld_max = std::max(std::max(ldo, T ? ldt : (int64_t)0), gn_saved ? ldxo : (int64_t)0);
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31) && src_rows * lda * 4 < (1ll << 31),
                      "dual_linear_fwd: rows * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
        // one round of the chip; never fewer than 64 rows per workgroup (the statistics have one entry per 64 rows)
        int64_t rows_main = ceil_div(ceil_div(n_nodes, (int64_t)256), (int64_t)16) * 16;
        if (rows_main < 64) rows_main = 64;
        const dim3 grid3((unsigned)ceil_div(n_nodes, rows_main));
        if (tiled_split_products())
            hipLaunchKernelGGL((trans_fwd3_kernel<128, true>), grid3, dim3(512), 0, st, xa, lda, src_rows, W, bias, mask, zr, omz, act, T,
                               ldt, out, ldo, n_nodes, stats, pro, xa_index, (int)rows_main);
        else
            hipLaunchKernelGGL((trans_fwd3_kernel<128, false>), grid3, dim3(512), 0, st, xa, lda, src_rows, W, bias, mask, zr, omz, act, T,
                               ldt, out, ldo, n_nodes, stats, pro, xa_index, (int)rows_main);
        return launch_status("glass_dual_linear_fwd_f32 (stage run, hidden 128)");
    }
    if (tiled_here(H)) {
        GnPrologue tpro = pro;
        tpro.gather = xa_index;
        tpro.gather_rows = xa_rows;
        return launch_tiled_fwd(xa, lda, xb, ldb, W, bias, mask, zr, omz, act, T, ldt, out, ldo, n_nodes, H, stats, tpro, st);
    }
#define GLASS_FWD(HH, CS, RW)                                                                                      \
    if (H == HH) {                                                                                                 \
        allow_lds(dual_fwd_kernel<HH, true, CS, RW>, lds_comb);                                                    \
        allow_lds(dual_fwd_kernel<HH, false, CS, RW>, lds_trans);                                                  \
        if (comb)                                                                                                  \
            hipLaunchKernelGGL((dual_fwd_kernel<HH, true, CS, RW>), grid, dim3(kWave * RW * CS), lds_comb, st, xa, lda,  \
                               xb, ldb, W, bias, mask, zr, omz, act, T, ldt, out, ldo, n_nodes, stats, stats_exact, pro, \
                               nullptr, 0);                                                                        \
        else                                                                                                       \
            hipLaunchKernelGGL((dual_fwd_kernel<HH, false, CS, RW>), grid, dim3(kWave * RW * CS), lds_trans, st, xa, lda, \
                               xb, ldb, W, bias, mask, zr, omz, act, T, ldt, out, ldo, n_nodes, stats, stats_exact, pro, \
                               xa_index, (int)xa_rows);                                                            \
    }
    if (GLASS_TRANS_FWD_V2 && !comb && H == 64) {  // (image in layout kLayoutWave16Cols: glass_dual_linear_fwd_layout)
        const int64_t src_rows = xa_index ? xa_rows : n_nodes;
        const int64_t ld_max = std::max(std::max(ldo, T ? ldt : (int64_t)0), gn_saved ? ldxo : (int64_t)0);
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31) && src_rows * lda * 4 < (1ll << 31),
                      "dual_linear_fwd: rows * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
        // 80-row tiles when they bring the launch down to one workgroup per CU (no per-workgroup partials then: their count
        // is the 64-row geometry of glass_dual_linear_stat_rows)
        const int64_t wg80 = ceil_div(n_nodes, 80);
        const bool tall = (stats == nullptr || stats_exact) && grid.x > 256 && wg80 <= 256;
#define GLASS_TF2(NS, WGN)                                                                                                  \
    hipLaunchKernelGGL((trans_fwd2_kernel<64, NS, WGN>), tall ? dim3((unsigned)wg80) : grid, dim3(kBlock * WGN), 0, st, xa, lda,     \
                       src_rows, W, bias, mask, zr, omz, act, T, ldt, out, ldo, n_nodes, stats, stats_exact, pro, xa_index)
#if GLASS_LAB
#include "../../tools/lab/dispatch_trans_fwd_wg1.inc"
#endif
        if (h64_split_products()) {
#define GLASS_TF2S(NS)                                                                                                      \
    hipLaunchKernelGGL((trans_fwd2_kernel<64, NS, 2, true>), tall ? dim3((unsigned)wg80) : grid, dim3(kBlock * 2), 0, st, xa, lda,  \
                       src_rows, W, bias, mask, zr, omz, act, T, ldt, out, ldo, n_nodes, stats, stats_exact, pro, xa_index)
            if (tall) GLASS_TF2S(5); else GLASS_TF2S(4);
#undef GLASS_TF2S
            return launch_status("glass_dual_linear_fwd_f32");
        }
        if (tall) GLASS_TF2(5, 2); else GLASS_TF2(4, 2);
#undef GLASS_TF2
        return launch_status("glass_dual_linear_fwd_f32");
    }
    GLASS_FWD(64, 1, 4)
#undef GLASS_FWD
    return launch_status("glass_dual_linear_fwd_f32");
}

// The weight-gradient half of glass_dual_linear_bwd_f32 (inputs of the pair + scratch for the partial sums)
struct BwdWgrad {
    const float* X; int64_t ldx;
    const float* X2; int64_t ldx2;
    void* ws;
};

static int dgrad_launch(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, double z_ratio,
                        int act, const float* WT, int64_t n_out, const float* addend, int64_t ldadd, float p_drop,
                        const uint64_t* rng_state, uint64_t call_id, float* out, int64_t ldo, int64_t n_nodes, int64_t H,
                        double* gn_partial, const float* gn_x, int64_t gn_ldx, const float* gn_saved,
                        const float* gn_alpha, int gn_act, float gn_p_drop, uint64_t gn_call_id, int gn_exact,
                        const BwdWgrad* wg, void* stream) {
    GLASS_REQUIRE(dsrc && mask && WT && out && n_nodes > 0, "dual_linear_dgrad: null pointer");
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || (rng_state && n_out == H)),
                  "dual_linear_dgrad: bad dropout args (the masked output must be the [N,H] layer input)");
    if (!dense_shape_ok(H) || (n_out != H && n_out != 2 * H)) {
        set_error("dual_linear_dgrad: unsupported shape H=%lld n_out=%lld", (long long)H, (long long)n_out);
        return GLASS_E_UNSUPPORTED;
    }
    if (narrow_shape_ok(H)) {  // thread-per-row kernels: any alignment; WT = the row-major weight itself ([2H][n_out])
        GLASS_REQUIRE(!gn_exact, "dual_linear_dgrad: exact GraphNorm accumulators are served at hidden 64 only");
        GLASS_REQUIRE(ldd >= H && ldo >= n_out && (act == GLASS_ACT_NONE || (T && ldt >= 2 * H)) && (!addend || ldadd >= n_out) &&
                          (!gn_partial || (gn_x && gn_saved && gn_alpha && gn_ldx >= H && (gn_p_drop == 0.f || rng_state))),
                      "dual_linear_dgrad: bad arguments");
        const GnBwdStats ngs{gn_partial, 0, gn_x, gn_ldx, gn_saved, gn_alpha, gn_act,
                             make_drop(gn_partial ? gn_p_drop : 0.f, gn_call_id, H)};
        const int rc = launch_narrow_dgrad(dsrc, ldd, act != GLASS_ACT_NONE ? T : nullptr, ldt, mask, (float)z_ratio,
                                           (float)(1.0 - z_ratio), act, WT, n_out, addend, ldadd, make_drop(p_drop, call_id, n_out),
                                           rng_state, out, ldo, n_nodes, H, ngs, (hipStream_t)stream);
        if (rc || !wg) return rc;
        return glass_dual_linear_wgrad_f32(dsrc, ldd, T, ldt, mask, z_ratio, act, wg->X, wg->ldx, wg->X2, wg->ldx2, n_nodes, H,
                                           nullptr, 0, nullptr, 0, wg->ws, stream);
    }
    GLASS_REQUIRE(ldd >= H && ldd % 4 == 0 && aligned16(dsrc) && aligned16(WT) && ldo >= n_out && ldo % 4 == 0 &&
                      aligned16(out) && (act == GLASS_ACT_NONE || (T && ldt >= 2 * H && ldt % 4 == 0 && aligned16(T))) &&
                      (!addend || (ldadd >= n_out && ldadd % 4 == 0 && aligned16(addend))),
                  "dual_linear_dgrad: operands must be 16-B aligned with ld %% 4 == 0");
    GLASS_REQUIRE(!gn_exact || (gn_partial && wave16_shape_ok(H) && aligned16(gn_partial) && rep_ok(gn_exact)),
                  "dual_linear_dgrad: exact GraphNorm accumulators are served at hidden 64 only (glass_gn_exact_supported)");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(n_nodes, glass_dual_linear_stat_rows(H)));
    const float zr = (float)z_ratio, omz = (float)(1.0 - z_ratio);
    const float* Tp = act != GLASS_ACT_NONE ? T : nullptr;
    const bool dg2 = GLASS_TRANS_DGRAD_V2 && H == 64 && n_out == H;  // (image in layout kLayoutWave16Cols: glass_dual_linear_dgrad_layout)
    size_t lds_dg = lds_bytes(n_out, 2);  // K = 2H always needs >= 2 passes
    if (dg2) {
        lds_dg = kTransDgrad2Lds;
        const int64_t ld_max = std::max(std::max(ldd, ldo), std::max(std::max(act != GLASS_ACT_NONE ? ldt : (int64_t)0, addend ? ldadd : (int64_t)0),
                                                                     gn_partial ? gn_ldx : (int64_t)0));
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31), "dual_linear_dgrad: n_nodes * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
    }
    const Drop drop = make_drop(p_drop, call_id, n_out);
    GLASS_REQUIRE(!gn_partial || (gn_x && gn_saved && gn_alpha && gn_ldx >= H && gn_ldx % 4 == 0 && aligned16(gn_x) &&
                                  aligned16(gn_saved) && aligned16(gn_alpha) && gn_p_drop >= 0.f && gn_p_drop < 1.f &&
                                  (gn_p_drop == 0.f || rng_state) && act_code_ok(gn_act)),
                  "dual_linear_dgrad: bad GraphNorm statistics arguments");
    const GnBwdStats gs{gn_partial, gn_exact, gn_x, gn_ldx, gn_saved, gn_alpha, gn_act,
                        make_drop(gn_partial ? gn_p_drop : 0.f, gn_call_id, H)};
    // the weight-gradient partials of the same pair, as a second launch (large graphs / wide layers) ...
    auto wgrad_after = [&]() -> int {
        if (!wg) return 0;
        return glass_dual_linear_wgrad_f32(dsrc, ldd, T, ldt, mask, z_ratio, act, wg->X, wg->ldx, wg->X2, wg->ldx2, n_nodes, H,
                                           nullptr, 0, nullptr, 0, wg->ws, stream);
    };
    if (GLASS_TRANS_DGRAD_V2 && H == 128 && n_out == H) {  // staged form with 8 waves (image in layout kLayoutWave16Cols)
        const int64_t ld_max = std::max(std::max(ldd, ldo), std::max(std::max(act != GLASS_ACT_NONE ? ldt : (int64_t)0, addend ? ldadd : (int64_t)0),
                                                                     gn_partial ? gn_ldx : (int64_t)0));
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31), "dual_linear_dgrad: n_nodes * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
        GLASS_REQUIRE(!gn_exact, "dual_linear_dgrad: exact GraphNorm accumulators are served at hidden 64 only");
        const DgradArgs d128{dsrc, ldd, Tp, ldt, mask, zr, omz, act, WT, addend, ldadd, drop, rng_state, out, ldo, n_nodes, gs};
        const int64_t n_tiles = ceil_div(n_nodes, 64);
        const bool split3 = tiled_split_products() && lab_knob("GLASS_TRANS_DGRAD3_SPLIT", 1) != 0;  // the call's product form (GLASS_DENSE_F32_PRODUCTS opts out)
        // more than one round of 64-row tiles: runs of stages, one workgroup per CU (trans_dgrad3_kernel); in the split form that
        // kernel serves the smaller graphs too, one 64-row tile (four stages) per workgroup
        if (lab_knob("GLASS_TRANS_DGRAD3", 1) && (n_tiles > 256 || split3)) {
            const int64_t n_stages = ceil_div(n_nodes, 16);
            int stages_per_wg = (int)ceil_div(n_stages, 256);  // never more workgroups than 64-row tiles (the partials' entries): >= 4 stages each
            if (stages_per_wg < 4) stages_per_wg = 4;
            if (split3) {
                const size_t lds3s = trans_dgrad3_split_lds(128);
                allow_lds((trans_dgrad3_kernel<128, true>), lds3s);
                hipLaunchKernelGGL((trans_dgrad3_kernel<128, true>), dim3((unsigned)ceil_div(n_stages, stages_per_wg)), dim3(512), lds3s, st, d128, stages_per_wg);
                const int rc3 = launch_status("glass_dual_linear_dgrad_f32 (staged, tall, hidden 128, split products)");
                return rc3 ? rc3 : wgrad_after();
            }
            const size_t lds3 = trans_dgrad3_lds(128);
            allow_lds(trans_dgrad3_kernel<128>, lds3);
            hipLaunchKernelGGL((trans_dgrad3_kernel<128>), dim3((unsigned)ceil_div(n_stages, stages_per_wg)), dim3(512), lds3, st, d128, stages_per_wg);
            const int rc3 = launch_status("glass_dual_linear_dgrad_f32 (staged, tall, hidden 128)");
            return rc3 ? rc3 : wgrad_after();
        }
        const size_t lds128 = trans_dgrad2_lds(128);
        allow_lds(trans_dgrad2_kernel<128>, lds128);
        hipLaunchKernelGGL((trans_dgrad2_kernel<128>), dim3((unsigned)ceil_div(n_nodes, 64)), dim3(512), lds128, st, d128);
        const int rc = launch_status("glass_dual_linear_dgrad_f32 (staged, hidden 128)");
        return rc ? rc : wgrad_after();
    }
    if (GLASS_COMB_DGRAD_V2 && H == 128 && n_out == 2 * H) {  // comb pair: stage-run form (image in layout kLayoutWave16EffDgradCols)
        if (act != GLASS_ACT_NONE || addend || p_drop > 0.f) {
            set_error("dual_linear_dgrad: at hidden 128 the 256-wide data gradient is the comb pair's (no activation, addend or dropout)");
            return GLASS_E_UNSUPPORTED;
        }
        GLASS_REQUIRE(!gn_exact, "dual_linear_dgrad: exact GraphNorm accumulators are served at hidden 64 only");
        const int64_t ld_max = std::max(std::max(ldd, ldo), gn_partial ? gn_ldx : (int64_t)0);
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31), "dual_linear_dgrad: n_nodes * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
        // one round of the chip; >= 64 rows per workgroup (one statistics entry per 64 rows), <= the list a workgroup can hold
        int64_t rows_main = ceil_div(ceil_div(n_nodes, (int64_t)256), (int64_t)16) * 16;
        if (rows_main < 64) rows_main = 64;
        if (rows_main > kCombDgrad3List) rows_main = kCombDgrad3List;
        const dim3 grid3((unsigned)ceil_div(n_nodes, rows_main));
        if (tiled_split_products()) {
            allow_lds((comb_dgrad3_kernel<128, true>), comb_dgrad3_lds<true>());
            hipLaunchKernelGGL((comb_dgrad3_kernel<128, true>), grid3, dim3(512), comb_dgrad3_lds<true>(), st, dsrc, ldd, mask, WT, rng_state,
                               out, ldo, n_nodes, gs, (int)rows_main);
        } else {
            allow_lds((comb_dgrad3_kernel<128, false>), comb_dgrad3_lds<false>());
            hipLaunchKernelGGL((comb_dgrad3_kernel<128, false>), grid3, dim3(512), comb_dgrad3_lds<false>(), st, dsrc, ldd, mask, WT, rng_state,
                               out, ldo, n_nodes, gs, (int)rows_main);
        }
        const int rc = launch_status("glass_dual_linear_dgrad_f32 (stage run, hidden 128, comb pair)");
        return rc ? rc : wgrad_after();
    }
    if (tiled_here(H)) {
        const int rc = launch_tiled_dgrad(dsrc, ldd, Tp, ldt, mask, zr, omz, act, WT, n_out, addend, ldadd, drop, rng_state,
                                          out, ldo, n_nodes, H, gs, st);
        return rc ? rc : wgrad_after();
    }
    const DgradArgs dargs{dsrc, ldd, Tp, ldt, mask, zr, omz, act, WT, addend, ldadd, drop, rng_state, out, ldo, n_nodes, gs};
    // ... or, at hidden 64 on a graph small enough to be latency-bound, as the second branch of the SAME launch
    const int64_t O = 2 * H, I = wg && wg->X2 ? 2 * H : H;
    if (wg && H == 64 && n_nodes <= kFusedBwdMaxRows && !wgrad_tiled_shape(n_nodes, O, I)) {
        GLASS_REQUIRE(wg->X && wg->ws && wg->ldx >= H && wg->ldx % 2 == 0 && (reinterpret_cast<uintptr_t>(wg->X) & 7u) == 0 &&
                          (!wg->X2 || (wg->ldx2 >= H && wg->ldx2 % 2 == 0 && (reinterpret_cast<uintptr_t>(wg->X2) & 7u) == 0)),
                      "dual_linear_bwd: the pair's inputs must be 8-B aligned with even leading dimensions");
        const WgradGeom g = wgrad_geom(n_nodes, O, I);
        float* part_w = (float*)wg->ws;
        const WgradSynth sy{dsrc, ldd, Tp, ldt, mask, zr, omz, act, (int)H, wg->X2, wg->ldx2};
        const size_t lds_wg = (size_t)(2 * kTile + 8 * kOT) * sizeof(float);
        const size_t lds_fused = lds_dg > lds_wg ? lds_dg : lds_wg;
        const unsigned blocks = grid.x + (unsigned)(g.n_slabs * g.ny * g.nz);
        // split products (the call's option, as in the forward): both bodies of the trans pair's fused launch, when they are the
        // staged ones (slabs of whole 32-row stages: wgrad_geom rounds them so for every fused launch)
        const bool sp = n_out == H && h64_split_products() && GLASS_TRANS_DGRAD_V2 && GLASS_TRANS_WGRAD_STAGED2 && O == 128 && I == 64 &&
                        g.ny == 1 && g.rows_per_slab % 32 == 0;
        if (sp) {
            const size_t lds_sp = std::max(kTransDgrad2SplitLds, kStg2sLdsBytes);
            allow_lds(dual_bwd_kernel<64, 64, true>, lds_sp);
            hipLaunchKernelGGL((dual_bwd_kernel<64, 64, true>), dim3(blocks), dim3(kBlock), lds_sp, st, dargs, (int)grid.x, wg->X,
                               wg->ldx, (int)O, (int)I, g.rows_per_slab, g.n_slabs, g.ny, part_w, part_w + g.part_w_floats,
                               part_w + g.part_w_floats + g.part_b_floats - kWgradHeaderFloats, sy);
        } else if (n_out == H) {
            allow_lds(dual_bwd_kernel<64, 64>, lds_fused);
            hipLaunchKernelGGL((dual_bwd_kernel<64, 64>), dim3(blocks), dim3(kBlock), lds_fused, st, dargs, (int)grid.x, wg->X,
                               wg->ldx, (int)O, (int)I, g.rows_per_slab, g.n_slabs, g.ny, part_w, part_w + g.part_w_floats,
                               part_w + g.part_w_floats + g.part_b_floats - kWgradHeaderFloats, sy);
        } else {
            allow_lds(dual_bwd_kernel<64, 128>, lds_fused);
            hipLaunchKernelGGL((dual_bwd_kernel<64, 128>), dim3(blocks), dim3(kBlock), lds_fused, st, dargs, (int)grid.x, wg->X,
                               wg->ldx, (int)O, (int)I, g.rows_per_slab, g.n_slabs, g.ny, part_w, part_w + g.part_w_floats,
                               part_w + g.part_w_floats + g.part_b_floats - kWgradHeaderFloats, sy);
        }
        return launch_status("glass_dual_linear_bwd_f32");
    }
#define GLASS_DG(HH, CS, RW)                                                                                       \
    if (H == HH) {                                                                                                 \
        allow_lds(dual_dgrad_kernel<HH, HH, CS, RW>, lds_dg);                                                      \
        allow_lds(dual_dgrad_kernel<HH, 2 * HH, CS, RW>, lds_dg);                                                  \
        if (n_out == H)                                                                                            \
            hipLaunchKernelGGL((dual_dgrad_kernel<HH, HH, CS, RW>), grid, dim3(kWave * RW * CS), lds_dg, st, dargs); \
        else                                                                                                       \
            hipLaunchKernelGGL((dual_dgrad_kernel<HH, 2 * HH, CS, RW>), grid, dim3(kWave * RW * CS), lds_dg, st, dargs); \
    }
    GLASS_DG(64, 1, 4)
#undef GLASS_DG
    const int rc = launch_status("glass_dual_linear_dgrad_f32");
    return rc ? rc : wgrad_after();
}

extern "C" int glass_dual_linear_dgrad_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt,
                                           const uint8_t* mask, double z_ratio, int act, const float* WT,
                                           int64_t n_out, const float* addend, int64_t ldadd, float p_drop,
                                           const uint64_t* rng_state, uint64_t call_id, float* out, int64_t ldo,
                                           int64_t n_nodes, int64_t H, double* gn_partial, const float* gn_x,
                                           int64_t gn_ldx, const float* gn_saved, const float* gn_alpha, int gn_act,
                                           float gn_p_drop, uint64_t gn_call_id, int gn_exact, void* stream) {
    const CallOptions call_options(act);
    return dgrad_launch(dsrc, ldd, T, ldt, mask, z_ratio, act, WT, n_out, addend, ldadd, p_drop, rng_state, call_id, out, ldo,
                        n_nodes, H, gn_partial, gn_x, gn_ldx, gn_saved, gn_alpha, gn_act, gn_p_drop, gn_call_id, gn_exact,
                        nullptr, stream);
}

extern "C" int glass_dual_linear_bwd_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask,
                                         double z_ratio, int act, const float* WT, int64_t n_out, const float* addend,
                                         int64_t ldadd, float p_drop, const uint64_t* rng_state, uint64_t call_id,
                                         float* out, int64_t ldo, int64_t n_nodes, int64_t H, double* gn_partial,
                                         const float* gn_x, int64_t gn_ldx, const float* gn_saved, const float* gn_alpha,
                                         int gn_act, float gn_p_drop, uint64_t gn_call_id, int gn_exact, const float* X,
                                         int64_t ldx, const float* X2, int64_t ldx2, void* ws, void* stream) {
    const CallOptions call_options(act);
    GLASS_REQUIRE(X && ws, "dual_linear_bwd: null pointer");
    const BwdWgrad wg{X, ldx, X2, ldx2, ws};
    return dgrad_launch(dsrc, ldd, T, ldt, mask, z_ratio, act, WT, n_out, addend, ldadd, p_drop, rng_state, call_id, out, ldo,
                        n_nodes, H, gn_partial, gn_x, gn_ldx, gn_saved, gn_alpha, gn_act, gn_p_drop, gn_call_id, gn_exact, &wg,
                        stream);
}

// ---- comb pair in effective-weight form (hidden 64; see comb_fwd_eff_kernel) -----------------------------------------
extern "C" int glass_comb_eff_supported(int64_t H) { return H == 64 ? 1 : 0; }
// ... the FORWARD alone also at hidden 128 (staged kernel with 8 waves; the backward of that width stays on the tiled kernels)
extern "C" int glass_comb_eff_fwd_supported(int64_t H) { return (H == 64 || (GLASS_COMB_FWD_V2 && H == 128)) ? 1 : 0; }

// workgroups of the launch = entries of `stats` / `gn_partial`: row tiles + extra workgroups for up to lab_cap listed rows
// Geometry of the comb forward launch in its third form (comb_fwd_eff3_kernel): main workgroups of rows_main rows — one
// ROUND of the chip (256 workgroups) once the graph is larger than 64 rows x 256, never fewer than 64 rows —, extra
// workgroups of 64 listed rows.  GLASS_COMB_FWD3=0: the second form's fixed 64 / 80-row tiles (laboratory A/B).
struct CombFwdGeom {
    int rows_main, rows_extra, n_main, n_extra;
};
static bool comb_fwd_pf_on() { return GLASS_LAB && lab_knob("GLASS_COMB_FWD_PF", 0) == 1; }
static bool comb_fwd3_on() { return lab_knob("GLASS_COMB_FWD3", 1) != 0 && GLASS_COMB_FWD_V2; }
static CombFwdGeom comb_fwd_geom(int64_t n_nodes, int64_t lab_cap, int64_t H);
// the third form is taken where its tiles are taller than the second form's 80 rows (below that the unrolled second form
// measured 0.7 us faster per launch at ppi_bp-shape: 24.6 vs 25.3 us for the two launches)
static bool comb_fwd3_tall(int64_t n_nodes, int64_t lab_cap, int64_t H);
static CombFwdGeom comb_fwd_geom(int64_t n_nodes, int64_t lab_cap, int64_t H) {
    CombFwdGeom g;
    const int64_t sr = (H == 64 && fwd_wave_groups(true) == 2) ? 32 : 16;  // stage height of the kernel that will run
    int64_t rows = ceil_div(ceil_div(n_nodes, (int64_t)256), sr) * sr;
    if (rows < 64) rows = 64;
    g.rows_main = (int)rows;
    g.rows_extra = 64;
    g.n_main = (int)ceil_div(n_nodes, rows);
    g.n_extra = (int)ceil_div(lab_cap, (int64_t)64);
    return g;
}

static bool comb_fwd3_tall(int64_t n_nodes, int64_t lab_cap, int64_t H) {
    return comb_fwd3_on() && comb_fwd_geom(n_nodes, lab_cap, H).rows_main > 80;
}

// partial-sum entries of the BACKWARD's gn_partial (64-row tiles + the extra workgroups) ...
extern "C" int64_t glass_comb_eff_blocks(int64_t n_nodes, int64_t H, int64_t lab_cap) {
    if (!glass_comb_eff_fwd_supported(H) || n_nodes <= 0 || lab_cap < 0) return GLASS_E_ARG;
    return ceil_div(n_nodes, 64) + ceil_div(lab_cap, 64);
}

// ... and of the FORWARD's `stats` in the partials form: one per workgroup of ITS geometry (tall row tiles on larger graphs)
extern "C" int64_t glass_comb_eff_fwd_blocks(int64_t n_nodes, int64_t H, int64_t lab_cap) {
    if (!glass_comb_eff_fwd_supported(H) || n_nodes <= 0 || lab_cap < 0) return GLASS_E_ARG;
    if (comb_fwd3_tall(n_nodes, lab_cap, H)) {
        const CombFwdGeom g = comb_fwd_geom(n_nodes, lab_cap, H);
        return g.n_main + g.n_extra;
    }
    return ceil_div(n_nodes, 64) + ceil_div(lab_cap, 64);
}

// dsrc_gn of glass_comb_eff_bwd_f32: both staged bodies inside the fused launch
extern "C" int glass_comb_eff_bwd_gn_src_supported(int64_t n_nodes, int64_t H) {
    return (GLASS_COMB_DGRAD_V2 && GLASS_SL_STAGED2 && !GLASS_COMB_BWD_V2 && H == 64 && n_nodes > 0 && n_nodes <= kFusedBwdMaxRows) ? 1 : 0;
}

// Layout code(s) of the data-gradient image glass_comb_eff_bwd_f32 reads: 7, and when the staged kernel is compiled in the
// buffer holds a SECOND pair of images in layout 10 behind the first (returns 10 then; 0 otherwise)
extern "C" int glass_comb_eff_dgrad_layout2(int64_t H) {
    (void)H;
    return (GLASS_COMB_BWD_V2 || GLASS_COMB_DGRAD_V2) ? kLayoutWave16EffDgradCols : 0;
}

// Most rows glass_comb_eff_fwd_f32 serves when every operand's row stride is <= ld floats (32-bit buffer offsets)
extern "C" int64_t glass_comb_eff_max_rows(int64_t ld) {
    if (ld <= 0) return GLASS_E_ARG;
    return GLASS_COMB_FWD_V2 ? ((1ll << 31) - 1) / (4 * ld) : (1ll << 31) - 1;
}

// Layout code (flags >> 1 of glass_dense_pack_batch_f32) of the forward image glass_comb_eff_fwd_f32 reads
extern "C" int glass_comb_eff_fwd_layout(int64_t H) {
    (void)H;
    return GLASS_COMB_FWD_V2 ? kLayoutWave16EffFwdCols : kLayoutWave16EffFwd;
}

extern "C" int glass_comb_eff_fwd_f32(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg_eff,
                                      const float* bias, const uint8_t* mask, double z_ratio, float* out, int64_t ldo,
                                      int64_t n_nodes, int64_t H, double* stats, int stats_exact, const float* gn_saved,
                                      const glass_gn_src* gn_src, int gn_act, float p_drop, const uint64_t* rng_state,
                                      uint64_t call_id, float* xa_out, int64_t ldxo, const int32_t* lab_rows,
                                      const int32_t* lab_count, int64_t lab_cap, void* stream) {
    const CallOptions call_options(gn_act);  // gn_act word -> activation code of the GraphNorm prologue + this call's options
    GLASS_REQUIRE(xa && xb && Wimg_eff && bias && mask && out && lab_rows && lab_count && n_nodes > 0 && lab_cap >= 0,
                  "comb_eff_fwd: null pointer");
    if (!glass_comb_eff_fwd_supported(H)) {
        set_error("comb_eff_fwd: hidden size %lld not supported (64, 128)", (long long)H);
        return GLASS_E_UNSUPPORTED;
    }
    GLASS_REQUIRE((!stats_exact && !gn_src) || glass_gn_exact_fwd_supported(H), "comb_eff_fwd: exact GraphNorm accumulators not served at this hidden size (glass_gn_exact_fwd_supported)");
    GLASS_REQUIRE(!gn_saved || (xa_out && ldxo >= H && ldxo % 4 == 0 && aligned16(xa_out) && aligned16(gn_saved) &&
                                p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state) &&
                                act_code_ok(gn_act)),
                  "comb_eff_fwd: bad GraphNorm prologue arguments");
    GLASS_REQUIRE(lda >= H && lda % 4 == 0 && aligned16(xa) && ldb >= H && ldb % 4 == 0 && aligned16(xb) &&
                      aligned16(Wimg_eff) && aligned16(bias) && ldo >= H && ldo % 4 == 0 && aligned16(out),
                  "comb_eff_fwd: operands must be 16-B aligned with ld %% 4 == 0");
    int n_main = (int)ceil_div(n_nodes, 64);
    dim3 grid((unsigned)(n_main + ceil_div(lab_cap, 64)));
    // 80-row tiles when they bring the launch down to one workgroup per CU (and no per-workgroup partials are written:
    // their count is glass_comb_eff_blocks = the 64-row geometry)
    const int64_t wg80 = ceil_div(n_nodes, 80) + ceil_div(lab_cap, 80);
    const bool tall = GLASS_COMB_FWD_V2 && (stats == nullptr || stats_exact) && grid.x > 256 && wg80 <= 256;
    if (tall) {
        n_main = (int)ceil_div(n_nodes, 80);
        grid = dim3((unsigned)wg80);
    }
    const float zr = (float)z_ratio, omz = (float)(1.0 - z_ratio);
    GnExactSrc esrc;
    GLASS_REQUIRE(make_exact_src(gn_src, gn_saved, esrc) && (!stats_exact || (stats && rep_ok(stats_exact))),
                  "comb_eff_fwd: bad gn_src (one accumulator block, all pointers set)");
    const GnPrologue pro{gn_saved, (int)H, gn_act, make_drop(gn_saved ? p_drop : 0.f, call_id, H), rng_state, xa_out, ldxo, esrc};
    const LabRows lab{lab_rows, lab_count, n_main, (int)lab_cap};
#if GLASS_LAB
#include "../../tools/lab/dispatch_comb_fwd_v1_lds.inc"
#endif
    const int64_t ld_max = std::max(std::max(lda, ldb), std::max(ldo, gn_saved ? ldxo : (int64_t)0));
    GLASS_REQUIRE(!GLASS_COMB_FWD_V2 || n_nodes * ld_max * 4 < (1ll << 31),
                  "comb_eff_fwd: n_nodes * ld * 4 must stay below 2^31 (32-bit buffer offsets; glass_comb_eff_max_rows)");
    GLASS_REQUIRE(!GLASS_COMB_FWD_V2 || !gn_saved || gn_act == GLASS_ACT_NONE,
                  "comb_eff_fwd: the GraphNorm in front of the comb pair has no activation (impl/models.py:165-166)");
#define GLASS_CF2(HH, DR, NS)                                                                                             \
    hipLaunchKernelGGL((comb_fwd_eff2_kernel<HH, DR, NS>), grid, dim3(4 * HH), 0, (hipStream_t)stream, xa, lda, xb, ldb, Wimg_eff, \
                       bias, mask, zr, omz, out, ldo, n_nodes, stats, stats_exact, pro, lab)
#define GLASS_CF2W(DR, NS)                                                                                                \
    hipLaunchKernelGGL((comb_fwd_eff2_kernel<64, DR, NS, 2>), grid, dim3(512), 0, (hipStream_t)stream, xa, lda, xb, ldb, Wimg_eff, \
                       bias, mask, zr, omz, out, ldo, n_nodes, stats, stats_exact, pro, lab)
    const bool dr = gn_saved && p_drop > 0.f;
    if (comb_fwd3_tall(n_nodes, lab_cap, H)) {
        const CombFwdGeom cg = comb_fwd_geom(n_nodes, lab_cap, H);
        const LabRows lab3{lab_rows, lab_count, cg.n_main, (int)lab_cap};
        const dim3 grid3((unsigned)(cg.n_main + cg.n_extra));
#define GLASS_CF3(HH, DR, WGN)                                                                                              \
    hipLaunchKernelGGL((comb_fwd_eff3_kernel<HH, DR, WGN>), grid3, dim3(4 * HH * WGN), 0, (hipStream_t)stream, xa, lda, xb, ldb,   \
                       Wimg_eff, bias, mask, zr, omz, out, ldo, n_nodes, stats, stats_exact, pro, lab3, cg.rows_main, cg.rows_extra)
#define GLASS_CF3S(DR)                                                                                                      \
    hipLaunchKernelGGL((comb_fwd_eff3_kernel<128, DR, 1, true>), grid3, dim3(512), 0, (hipStream_t)stream, xa, lda, xb, ldb,       \
                       Wimg_eff, bias, mask, zr, omz, out, ldo, n_nodes, stats, stats_exact, pro, lab3, cg.rows_main, cg.rows_extra)
        if (H == 128 && tiled_split_products()) {  // (the split form of the tiled family; GLASS_DENSE_F32_PRODUCTS in gn_act opts out)
            if (dr) GLASS_CF3S(true); else GLASS_CF3S(false);
        } else if (H == 128) {
            if (dr) GLASS_CF3(128, true, 1); else GLASS_CF3(128, false, 1);
#if GLASS_LAB
#include "../../tools/lab/dispatch_comb_fwd3_wg2.inc"
#endif
        } else {
            if (dr) GLASS_CF3(64, true, 1); else GLASS_CF3(64, false, 1);
        }
#undef GLASS_CF3
#undef GLASS_CF3S
        return launch_status("glass_comb_eff_fwd_f32");
    }
    if (GLASS_COMB_FWD_V2 && H == 128) {
        if (dr && tall) GLASS_CF2(128, true, 5);
        else if (dr) GLASS_CF2(128, true, 4);
        else if (tall) GLASS_CF2(128, false, 5);
        else GLASS_CF2(128, false, 4);
#if GLASS_LAB
#include "../../tools/lab/dispatch_comb_fwd2_wg2_pf.inc"
#endif
    } else if (GLASS_COMB_FWD_V2 && h64_split_products()) {
#define GLASS_CF2S(DR, NS)                                                                                                \
    hipLaunchKernelGGL((comb_fwd_eff2_kernel<64, DR, NS, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, xa, lda, xb, ldb,    \
                       Wimg_eff, bias, mask, zr, omz, out, ldo, n_nodes, stats, stats_exact, pro, lab)
        if (dr && tall) GLASS_CF2S(true, 5);
        else if (dr) GLASS_CF2S(true, 4);
        else if (tall) GLASS_CF2S(false, 5);
        else GLASS_CF2S(false, 4);
#undef GLASS_CF2S
    } else if (GLASS_COMB_FWD_V2) {
        if (dr && tall) GLASS_CF2(64, true, 5);
        else if (dr) GLASS_CF2(64, true, 4);
        else if (tall) GLASS_CF2(64, false, 5);
        else GLASS_CF2(64, false, 4);
    }
#undef GLASS_CF2
#undef GLASS_CF2W
#if GLASS_LAB
#include "../../tools/lab/dispatch_comb_fwd_v1.inc"
#endif
    return launch_status("glass_comb_eff_fwd_f32");
}

extern "C" int glass_comb_eff_bwd_f32(const float* dsrc, int64_t ldd, const uint8_t* mask, double z_ratio,
                                      const float* WTimg_eff, float* out, int64_t ldo, int64_t n_nodes, int64_t H,
                                      double* gn_partial, const float* gn_x, int64_t gn_ldx, const float* gn_saved,
                                      const float* gn_alpha, int gn_act, float gn_p_drop, const uint64_t* rng_state,
                                      uint64_t gn_call_id, int gn_exact, const float* X, int64_t ldx, const float* X2,
                                      int64_t ldx2, void* ws, const int32_t* lab_rows, const int32_t* lab_count,
                                      int64_t lab_cap, const glass_gn_bwd_src* dsrc_gn, void* stream) {
    const CallOptions call_options(gn_act);  // gn_act word -> activation code of conv.gn + this call's options (product form)
    GLASS_REQUIRE((dsrc || dsrc_gn) && mask && WTimg_eff && out && lab_rows && lab_count && n_nodes > 0 && lab_cap >= 0,
                  "comb_eff_bwd: null pointer");
    GnBwdSrc gsrc{};
    if (dsrc_gn) {
        const glass_gn_bwd_src& g = *dsrc_gn;
        if (!glass_comb_eff_bwd_gn_src_supported(n_nodes, H) || !X) {
            set_error("comb_eff_bwd: dsrc_gn is served by the staged fused launch only (hidden 64, small graphs, with the weight gradient)");
            return GLASS_E_UNSUPPORTED;
        }
        GLASS_REQUIRE(g.acc && rep_ok(g.n_rep) && g.dy && g.x && g.saved && g.gamma && g.alpha && g.lddy >= H && g.lddy % 4 == 0 &&
                          g.ldx >= H && g.ldx % 4 == 0 && aligned16(g.dy) && aligned16(g.x) && aligned16(g.saved) &&
                          (!g.addend || (g.ldadd >= H && g.ldadd % 4 == 0 && aligned16(g.addend))) && g.p_drop >= 0.f && g.p_drop < 1.f &&
                          (g.p_drop == 0.f || rng_state) && act_code_ok(g.act) &&
                          n_nodes * std::max(std::max(g.lddy, g.ldx), g.addend ? g.ldadd : (int64_t)0) * 4 < (1ll << 31),
                      "comb_eff_bwd: bad dsrc_gn");
        gsrc = GnBwdSrc{reinterpret_cast<const long long*>(g.acc), (int)g.n_rep, g.dy, g.lddy, g.x, g.ldx, g.addend, g.ldadd, g.saved,
                        g.gamma, g.alpha, g.dgamma, g.dbeta, g.dalpha, g.accumulate, g.act, make_drop(g.p_drop, g.call_id, H)};
        if (!dsrc) {
            dsrc = g.dy;  // (only its alignment is looked at below)
            ldd = g.lddy;
        }
    }
    if (H != 64) {
        set_error("comb_eff_bwd: hidden size %lld not supported (64)", (long long)H);
        return GLASS_E_UNSUPPORTED;
    }
    GLASS_REQUIRE(ldd >= H && ldd % 4 == 0 && aligned16(dsrc) && aligned16(WTimg_eff) && ldo >= 2 * H && ldo % 4 == 0 &&
                      aligned16(out),
                  "comb_eff_bwd: operands must be 16-B aligned with ld %% 4 == 0");
    GLASS_REQUIRE(!gn_partial || (gn_x && gn_saved && gn_alpha && gn_ldx >= H && gn_ldx % 4 == 0 && aligned16(gn_x) &&
                                  aligned16(gn_saved) && aligned16(gn_alpha) && gn_p_drop >= 0.f && gn_p_drop < 1.f &&
                                  (gn_p_drop == 0.f || rng_state) && act_code_ok(gn_act)),
                  "comb_eff_bwd: bad GraphNorm statistics arguments");
    GLASS_REQUIRE(!gn_exact || (gn_partial && rep_ok(gn_exact)), "comb_eff_bwd: gn_exact = replicas of the accumulators (2, 4, 8, 16)");
    hipStream_t st = (hipStream_t)stream;
    const int n_main = (int)ceil_div(n_nodes, 64);
    const unsigned n_dg = (unsigned)(n_main + ceil_div(lab_cap, 64));
    const float zr = (float)z_ratio;
    const GnBwdStats gs{gn_partial, gn_exact, gn_x, gn_ldx, gn_saved, gn_alpha, gn_act, make_drop(gn_partial ? gn_p_drop : 0.f, gn_call_id, H)};
    const DgradEffArgs dargs{dsrc, ldd, mask, WTimg_eff, rng_state, out, ldo, n_nodes, gs, LabRows{lab_rows, lab_count, n_main, (int)lab_cap}, gsrc};
    const size_t lds_dg = lds_bytes(2 * H, 1);  // one K pass of the [2H][H] effective weight
    if (!X) {  // data gradient only
        hipLaunchKernelGGL((comb_dgrad_eff_kernel<64>), dim3(n_dg), dim3(kBlock), lds_dg, st, dargs);
        return launch_status("glass_comb_eff_bwd_f32 (dgrad)");
    }
    GLASS_REQUIRE(X2 && ws && ldx >= H && ldx % 4 == 0 && aligned16(X) && ldx2 >= H && ldx2 % 4 == 0 && aligned16(X2) &&
                      aligned16(ws),
                  "comb_eff_bwd: the pair's inputs must be 16-B aligned with ld %% 4 == 0");
    // weight-gradient partials in S / L form (wgrad_common.h): reduced later by glass_linear_wgrad_reduce_batch_f32 with
    // lab_cap[job] = this call's lab_cap
    const WgradSLGeom g = wgrad_sl_geom(n_nodes, lab_cap);
    float* part_w = (float*)ws;
    float* part_b = part_w + g.part_w_floats;
    const WgradSL sl{dsrc, ldd, X, ldx, X2, ldx2, lab_rows, lab_count, g.n_s, g.rows_per_slab, g.n_l, gsrc, rng_state};
    if (n_nodes > kFusedBwdMaxRows) {  // large graph: the two halves fill the chip on their own
        hipLaunchKernelGGL((comb_dgrad_eff_kernel<64>), dim3(n_dg), dim3(kBlock), lds_dg, st, dargs);
        launch_wgrad_sl(sl, n_nodes, zr, part_w, part_b, st);
        return launch_status("glass_comb_eff_bwd_f32 (two launches)");
    }
#if GLASS_LAB && GLASS_COMB_BWD_V2
#include "../../tools/lab/dispatch_comb_bwd_v2.inc"
#endif
    if (GLASS_COMB_DGRAD_V2) {
        GLASS_REQUIRE(gn_act == GLASS_ACT_NONE, "comb_eff_bwd: the GraphNorm in front of the comb pair has no activation");
        const int64_t ld_max = std::max(std::max(ldd, ldo), gn_partial ? gn_ldx : (int64_t)0);
        GLASS_REQUIRE(n_nodes * ld_max * 4 < (1ll << 31), "comb_eff_bwd: n_nodes * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
    }
    const size_t lds_wg = (size_t)(2 * kTile + 8 * kSLOut) * sizeof(float);
    static_assert(kCombDgrad2Lds <= (size_t)(2 * kTile + 8 * kSLOut) * sizeof(float), "the staged data gradient fits the fused launch's LDS");
    const size_t lds_fused = lds_dg > lds_wg ? lds_dg : lds_wg;
    if (h64_split_products() && GLASS_COMB_DGRAD_V2 && GLASS_SL_STAGED2 && g.rows_per_slab % 32 == 0) {
        // split products (the call's option: GLASS_DENSE_F32_PRODUCTS in gn_act opts out) in both staged bodies
        const size_t lds_sp = std::max(kCombDgrad2SplitLds, kStg2sLdsBytes);
        allow_lds(comb_bwd_eff_kernel<64, true>, lds_sp);
        hipLaunchKernelGGL((comb_bwd_eff_kernel<64, true>), dim3(n_dg + (unsigned)(g.n_s + g.n_l)), dim3(kBlock), lds_sp, st, dargs,
                           (int)n_dg, sl, zr, part_w, part_b);
        return launch_status("glass_comb_eff_bwd_f32");
    }
    allow_lds(comb_bwd_eff_kernel<64>, lds_fused);
    hipLaunchKernelGGL((comb_bwd_eff_kernel<64>), dim3(n_dg + (unsigned)(g.n_s + g.n_l)), dim3(kBlock), lds_fused, st, dargs,
                       (int)n_dg, sl, zr, part_w, part_b);
    return launch_status("glass_comb_eff_bwd_f32");
}

// scratch bytes of glass_comb_eff_bwd_f32's weight-gradient partials
extern "C" int64_t glass_comb_eff_ws_bytes(int64_t n_nodes, int64_t H, int64_t lab_cap) {
    if (H != 64 || n_nodes <= 0 || lab_cap < 0) return GLASS_E_ARG;
    const WgradSLGeom g = wgrad_sl_geom(n_nodes, lab_cap);
    return (g.part_w_floats + g.part_b_floats) * (int64_t)sizeof(float);
}

static int pack_launch(const float* const* src, float* const* dst, const int64_t* dst_floats, const int64_t* NT, const int64_t* KT,
                       const int32_t* transposed, const float* z_ratio, int64_t n_jobs, uint64_t* rng_state,
                       const TableJob& tab, void* stream, const char* what, ZeroJob zero = ZeroJob{nullptr, 0},
                       const LabelJob* lj = nullptr) {
    GLASS_REQUIRE(n_jobs >= 0 && n_jobs <= kMaxPackJobs && (n_jobs == 0 || (src && dst && dst_floats && NT && KT && transposed)),
                  "%s: bad arguments (at most %d matrices per call)", what, kMaxPackJobs);
    PackBatch b;
    for (int k = 0; k < kMaxPackJobs; ++k) b.job[k] = PackJob{nullptr, nullptr, 0, 0, 0, 0, 0.f, 0};
    for (int k = 0; k < n_jobs; ++k) {
        GLASS_REQUIRE(src[k] && dst[k] && NT[k] > 0 && NT[k] % 64 == 0 && KT[k] > 0 && KT[k] % 64 == 0 && aligned16(src[k]) &&
                          aligned16(dst[k]),
                      "%s: job %d needs NT, KT multiples of 64 and 16-B aligned buffers", what, k);
        const int layout = transposed[k] >> 1;
        // the image sizes differ by layout (appendix, cut image): the caller states what dst[k] holds, nothing is assumed
        GLASS_REQUIRE(dst_floats[k] >= glass_dense_image_floats(NT[k], KT[k], transposed[k]),
                      "%s: job %d: dst holds %lld floats, the image of this layout needs %lld (glass_dense_image_floats)", what, k,
                      (long long)dst_floats[k], (long long)glass_dense_image_floats(NT[k], KT[k], transposed[k]));
        GLASS_REQUIRE(layout == kLayoutWave16 ||
                          (layout == kLayoutWave16Cols && ((NT[k] == 128 && KT[k] == 64 && !(transposed[k] & 1)) ||
                                                           (NT[k] == 256 && KT[k] == 128 && !(transposed[k] & 1)) ||
                                                           (NT[k] == 64 && KT[k] == 128 && (transposed[k] & 1)) ||
                                                           (NT[k] == 128 && KT[k] == 256 && (transposed[k] & 1)))) ||
                          ((layout == kLayoutTiledPaired || layout == kLayoutTiledPlain) && NT[k] % 256 == 0) ||
                          (layout == kLayoutTiledSplit && NT[k] == 128 && KT[k] == 256 && (transposed[k] & 1)) ||
                          (layout == kLayoutTiledPlainEff && NT[k] % 256 == 0 && KT[k] % 32 == 0 && (transposed[k] & 1) && z_ratio) ||
                          (layout == kLayoutTiledPairedEff && NT[k] % 512 == 0 && !(transposed[k] & 1) && z_ratio) ||
                          ((layout == kLayoutWave16EffFwd || layout == kLayoutWave16EffFwdCols) && NT[k] % 128 == 0 &&
                           !(transposed[k] & 1) && z_ratio) ||
                          ((layout == kLayoutWave16EffDgrad || layout == kLayoutWave16EffDgradCols) && KT[k] % 128 == 0 &&
                           (transposed[k] & 1) && z_ratio),
                      "%s: job %d: unknown layout %d, NT not a multiple of 256 for a tiled layout, or a split "
                      "layout that is not the transposed 128 x 256 operand", what, k, layout);
        const bool tiled_layout = layout == kLayoutTiledPaired || layout == kLayoutTiledPlain || layout == kLayoutTiledSplit ||
                                  layout == kLayoutTiledPlainEff || layout == kLayoutTiledPairedEff;
        b.job[k] = PackJob{src[k], dst[k], (int)NT[k], (int)KT[k], transposed[k] & 1, layout, z_ratio ? z_ratio[k] : 0.f,
                           tiled_layout ? 1 : 0};  // (always: either product form may read this image, call by call)
    }
    unsigned gx = 32;
    if (tab.W && (unsigned)ceil_div(tab.H, kTabCols) > gx) gx = (unsigned)ceil_div(tab.H, kTabCols);
    unsigned gy = (unsigned)n_jobs + (tab.W ? 1u : 0u);
    if (gy == 0) gy = 1;  // (only the zero-fill / the dropout stream to serve)
    if (lj)
        hipLaunchKernelGGL(step_head_kernel, dim3(gx, gy + 1), dim3(kLabThreads), 0, (hipStream_t)stream, b, rng_state, tab, (int)n_jobs,
                           zero, (int)gy, *lj);
    else
        hipLaunchKernelGGL(pack_batch_kernel, dim3(gx, gy), dim3(kBlock), 0, (hipStream_t)stream, b, rng_state, tab, (int)n_jobs, zero);
    return launch_status(what);
}

// Floats an operand image of (NT, KT, flags = transposed | layout << 1) occupies: NT*KT, + half of it for the effective-weight
// appendix of layouts 4 / 5, and for the tiled layouts (1..5) the same again x 3/2 behind it — the image cut into bf16 pieces
// that the split product form reads (always written, whatever the product form at pack time).
extern "C" int64_t glass_dense_image_floats(int64_t NT, int64_t KT, int32_t flags) {
    if (NT <= 0 || KT <= 0) return GLASS_E_ARG;
    const int layout = flags >> 1;
    int64_t base = NT * KT;
    if (layout == kLayoutTiledPlainEff || layout == kLayoutTiledPairedEff) base += base / 2;
    const bool tiled_layout = layout == kLayoutTiledPaired || layout == kLayoutTiledPlain || layout == kLayoutTiledSplit ||
                              layout == kLayoutTiledPlainEff || layout == kLayoutTiledPairedEff;
    return tiled_layout ? base + base * 3 / 2 : base;
}

extern "C" int glass_dense_pack_batch_f32(const float* const* src, float* const* dst, const int64_t* dst_floats,
                                          const int64_t* NT, const int64_t* KT, const int32_t* transposed,
                                          const float* z_ratio, int64_t n_jobs, uint64_t* rng_state, void* stream) {
    GLASS_REQUIRE(src && dst && dst_floats && NT && KT && transposed && n_jobs >= 0 && n_jobs <= kMaxPackJobs,
                  "dense_pack_batch: bad arguments (at most %d matrices per call)", kMaxPackJobs);
    if (n_jobs == 0) return rng_state ? glass_rng_advance(rng_state, stream) : 0;
    return pack_launch(src, dst, dst_floats, NT, KT, transposed, z_ratio, n_jobs, rng_state, TableJob{}, stream, "glass_dense_pack_batch_f32");
}

// The once-per-step prologue as ONE launch: the weight packing above + emb_gn's statistics through the embedding table
// (the first half of glass_embed_norm_fwd_f32: saved[4H]; table may be NULL when the consumer gathers from W itself).
extern "C" int glass_step_prologue_f32(const float* const* src, float* const* dst, const int64_t* dst_floats,
                                       const int64_t* NT, const int64_t* KT, const int32_t* transposed, const float* z_ratio, int64_t n_jobs, uint64_t* rng_state,
                                       const float* W, int64_t V, const int32_t* class_rowptr, const float* gamma,
                                       const float* beta, const float* alpha, float eps, float* saved, float* table,
                                       int64_t H, int64_t* zero_words, int64_t n_zero_words, void* stream) {
    GLASS_REQUIRE(!W || (class_rowptr && gamma && beta && alpha && saved && H > 0 && V > 0 && V <= GLASS_EMBED_NORM_MAX_ROWS),
                  "step_prologue: bad embedding-table arguments (at most %d rows)", GLASS_EMBED_NORM_MAX_ROWS);
    GLASS_REQUIRE(n_zero_words >= 0 && (n_zero_words == 0 || zero_words), "step_prologue: bad zero-fill arguments");
    TableJob tab{};
    if (W) tab = TableJob{W, (int)V, (int)H, class_rowptr, gamma, beta, alpha, eps, saved, table};
    return pack_launch(src, dst, dst_floats, NT, KT, transposed, z_ratio, n_jobs, rng_state, tab, stream, "glass_step_prologue_f32",
                       ZeroJob{n_zero_words > 0 ? (long long*)zero_words : nullptr, n_zero_words});
}

// The head of a replayed step: the prologue above and the label launch of the step's batch (labels.hip) in one grid.
extern "C" int glass_step_head_f32(const float* const* src, float* const* dst, const int64_t* dst_floats, const int64_t* NT,
                                   const int64_t* KT, const int32_t* transposed, const float* z_ratio, int64_t n_jobs,
                                   uint64_t* rng_state, const float* W, int64_t V, const int32_t* class_rowptr, const float* gamma,
                                   const float* beta, const float* alpha, float eps, float* saved, float* table, int64_t H,
                                   int64_t* zero_words, int64_t n_zero_words, glass_batch_cursor* cur, int64_t n_idx,
                                   int64_t smax, int64_t y_row_bytes, int64_t* pos_dst, void* y_dst, uint8_t* mask,
                                   int32_t* lab_rows, int32_t* lab_count, void* ws, int64_t n_nodes, void* stream) {
    GLASS_REQUIRE(!W || (class_rowptr && gamma && beta && alpha && saved && H > 0 && V > 0 && V <= GLASS_EMBED_NORM_MAX_ROWS),
                  "step_head: bad embedding-table arguments (at most %d rows)", GLASS_EMBED_NORM_MAX_ROWS);
    GLASS_REQUIRE(n_zero_words >= 0 && (n_zero_words == 0 || zero_words), "step_head: bad zero-fill arguments");
    GLASS_REQUIRE(cur && pos_dst && mask && lab_rows && lab_count && ws, "step_head: null pointer");
    GLASS_REQUIRE(smax > 0 && n_idx > 0 && n_idx * smax < (1ll << 30) && n_nodes > 0 && n_nodes < (1ll << 31) &&
                      (reinterpret_cast<uintptr_t>(cur) & 7u) == 0,
                  "step_head: bad sizes");
    GLASS_REQUIRE(y_row_bytes == 0 || (y_dst && y_row_bytes > 0 && y_row_bytes % 4 == 0 && y_row_bytes < (1ll << 30) &&
                                       (reinterpret_cast<uintptr_t>(y_dst) & 3u) == 0),
                  "step_head: the target rows have 4-byte granularity");
    TableJob tab{};
    if (W) tab = TableJob{W, (int)V, (int)H, class_rowptr, gamma, beta, alpha, eps, saved, table};
    const LabelJob lj{cur, (int)n_idx, (int)smax, (int)(y_row_bytes / 4), pos_dst, (uint32_t*)y_dst, mask, lab_rows, lab_count,
                      (int32_t*)ws, n_nodes};
    return pack_launch(src, dst, dst_floats, NT, KT, transposed, z_ratio, n_jobs, rng_state, tab, stream, "glass_step_head_f32",
                       ZeroJob{n_zero_words > 0 ? (long long*)zero_words : nullptr, n_zero_words}, &lj);
}
