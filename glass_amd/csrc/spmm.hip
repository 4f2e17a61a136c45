// K1: CSR aggregation Y = A @ X for gfx950.   Replaces `self.adj @ x` (reference
// impl/models.py:164) and, on the CSR of A^T, its autograd backward.
//
// Shape of the work: per edge one gather of a 4H-byte feature row (256 B at H=64) — HBM/L2
// bound, no matrix-core work.  Layout per wave64: a feature row is covered by LPR lanes x 16 B
// (LPR=16 at H=64), so one wave-instruction gathers G = 64/LPR neighbour rows = 1 KiB, fully
// coalesced per row.  A wave first reads up to 64 (col,val) pairs of its row coalesced, then
// hands them to the lane groups with ds_bpermute (__shfl) — one dependent latency per 64 edges —
// and keeps U gathers in flight.  Partial sums of the G groups are combined with __shfl_xor;
// rows long enough to need a whole workgroup are combined through LDS; rows longer than one
// chunk go through per-chunk partial rows summed in fixed order (no float atomics).
#include "common.h"
#include "spmm_common.h"

#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace glass {

// stand-alone form: matrices that are ALL long rows (the selection matrix of the embedding backward)
template <int VW, int LPR, int U, bool NT>
__global__ __launch_bounds__(kBlock) void spmm_long_kernel(const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float* __restrict__ X, int64_t ldx,
                                                           float* __restrict__ Y, int64_t ldy,
                                                           float* __restrict__ partials, int H,
                                                           const int32_t* __restrict__ items) {
    __shared__ float lds[(kBlock / kWave) * LPR * VW];
    long_item_body<VW, LPR, U, NT>(col, val, X, ldx, Y, ldy, partials, H, items + 4 * (int64_t)blockIdx.x, lds);
}

#ifdef GLASS_K1_TRACE  // laboratory build only (tools/Makefile `lab`): per-wave wall-clock stamps of the sweep kernel
__device__ unsigned long long* g_k1_trace;
#define K1_STAMP(slot) do { if (g_k1_trace && lane == 0 && blockIdx.y == 0) g_k1_trace[(size_t)wave * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define K1_STAMP(slot) do { } while (0)
#endif

// ---- sweep kernel: one wave per plan item (r0, r1, e0, e1): <= 64 consecutive short rows holding <= 256 edges ----
// The item carries its edge range, so the index loads (row pointers, col, val) depend on ONE scalar load of the plan
// and not on a chain plan -> rowptr -> col: plan -> {rowptr chunk, col/val} -> X rows -> store.
//
// Two modes, chosen per wave from the item alone (wave-uniform, a pure function of the plan and H, so results are
// bitwise repeatable):
//  * row mode (mean degree > rp_factor * G): one row at a time, its edges split over the G lane groups, partial sums
//    combined with __shfl_xor;
//  * flat mode (short rows): the item's col/val/rowptr are staged once in LDS with coalesced loads; each lane group
//    owns an edge-balanced run of whole rows and walks ITS edges as one flat stream, U gathers in flight whatever
//    the row lengths (a row boundary only flushes the accumulator), rows summed in plain edge order, no cross-group
//    reduction.  On a degree-1 pattern that is U*G independent 4H-byte gathers in flight per wave behind a single
//    index round trip (the previous form had 2 per group behind three dependent round trips: 0.55 of the HBM roofline).
constexpr int64_t kStreamNtBytes = 256ll << 20;  // Infinity Cache size
constexpr int kFlatMinItems = 8192;   // the flat-capable sweep kernel only on throughput-bound sweeps ...
constexpr int kFlatMinShare15 = 4;    // ... with >= 4/15 of the sweep cost in flat-eligible items
// Tunables (overridable only by the laboratory build, tools/Makefile `lab`): gathers in flight per lane group and the
// sweep item caps
#ifndef GLASS_K1_U
#define GLASS_K1_U 8
#endif
#ifndef GLASS_K1_ITEM_EDGES
#define GLASS_K1_ITEM_EDGES 256
#endif
constexpr int kGathers = GLASS_K1_U;
constexpr int kItemRows = 64;                    // rows per sweep item (one coalesced rowptr load per wave)
constexpr int kItemEdges = GLASS_K1_ITEM_EDGES;  // edges per sweep item (LDS staging: 2 KiB of (col,val) per wave at 256)

// FLAT = false: a specialisation without the flat branch, its LDS staging and its registers (64 instead of 76 VGPRs: 8
// instead of 6 waves per SIMD); every item then runs in row mode, which is correct for any item.  Chosen on the host
// (launch_spmm_u) unless flat mode pays: the sweep must be throughput-bound (>= 8 192 items; on small graphs the
// serial per-group walk of flat mode lengthens the critical path: density-shape 5.9 vs 4.8 us) AND a real share of the
// work must sit in flat-eligible items (header word H_FLAT_SHARE; ppi_bp-shape: a few low-degree rows out of 17 080
// qualify, and carrying the flat branch for them cost 13.4 vs 12.2 us per launch).
template <int VW, int LPR, int U, bool NT, bool FLAT>
// (row-only build at H >= 64: held to 64 VGPRs = 8 waves per SIMD; the long-row branch would otherwise cost a 65th)
__global__ __launch_bounds__(kBlock, (!FLAT && LPR >= 16) ? 8 : 1) void spmm_sweep_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float* __restrict__ X, int64_t ldx,
                                                            float* __restrict__ Y, int64_t ldy, int H,
                                                            const int32_t* __restrict__ items, int n_waves,
                                                            int rp_factor, const int32_t* __restrict__ long_items,
                                                            int n_sweep_blocks, float* __restrict__ partials) {
    constexpr int G = kWave / LPR;
    constexpr int kWaves = kBlock / kWave;
    constexpr bool kFlat = FLAT && G > 1;
    __shared__ int32_t s_rp[kFlat ? kWaves : 1][kFlat ? kItemRows + 1 : 1];
    __shared__ int32_t s_col[kFlat ? kWaves : 1][kFlat ? kItemEdges : 1];
    __shared__ float s_val[kFlat ? kWaves : 1][kFlat ? kItemEdges : 1];
    __shared__ float s_long[kWaves * LPR * VW];
    if ((int)blockIdx.x >= n_sweep_blocks) {
        // The workgroups past the sweep's own run the plan's long-row items in the same launch: on graphs that have both
        // kinds of rows a second dependent launch for a handful of long rows cost more than the rows themselves (the
        // shipped density graph: 4 955 sweep waves + 4 long rows, 12.8 us as two launches).
        long_item_body<VW, LPR, U, NT>(col, val, X, ldx, Y, ldy, partials, H,
                                       long_items + 4 * (int64_t)((int)blockIdx.x - n_sweep_blocks), s_long);
        return;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * kWaves + w;
    if (wave >= n_waves) return;
    const int grp = lane / LPR, sub = lane % LPR;
    const int coff = (blockIdx.y * LPR + sub) * VW;
    const bool col_ok = coff < H;
    const float* Xc = X + coff;
    K1_STAMP(0);  // wave start | item known | row pointers known | done
    const int4 it = reinterpret_cast<const int4*>(items)[wave];
    const int r0 = __builtin_amdgcn_readfirstlane(it.x), r1 = __builtin_amdgcn_readfirstlane(it.y);
    const int e0 = __builtin_amdgcn_readfirstlane(it.z), e1 = __builtin_amdgcn_readfirstlane(it.w);
    const int nrows = r1 - r0, ne = e1 - e0;
    K1_STAMP(1);
    // start edge of row r0 + lane (lanes past the item hold e1, so "end of row i" is always lane i + 1's value or e1)
    const int rp_reg = (lane < nrows) ? ld_stream<NT>(rowptr + r0 + lane) : e1;
    if (kFlat && ne <= rp_factor * G * nrows) {
        // ---- flat mode ----
        int32_t* rp_s = s_rp[kFlat ? w : 0];
        int32_t* col_s = s_col[kFlat ? w : 0];
        float* val_s = s_val[kFlat ? w : 0];
        int cr[kItemEdges / kWave];
        float vr[kItemEdges / kWave];
#pragma unroll
        for (int k = 0; k < kItemEdges / kWave; ++k) {
            const int idx = k * kWave + lane;
            cr[k] = 0;
            vr[k] = 0.f;
            if (idx < ne) {
                cr[k] = ld_stream<NT>(col + e0 + idx);
                vr[k] = ld_stream<NT>(val + e0 + idx);
            }
        }
        rp_s[lane] = rp_reg - e0;
        if (lane == 0) rp_s[kItemRows] = ne;
#pragma unroll
        for (int k = 0; k < kItemEdges / kWave; ++k) {
            col_s[k * kWave + lane] = cr[k];
            val_s[k * kWave + lane] = vr[k];
        }
        // this lane group's rows [rbeg, rend): edge-balanced cut points (row granularity) when G is small, equal row
        // counts otherwise
        int rbeg, rend;
        if (G <= 8) {
            rbeg = 0;
            rend = nrows;
#pragma unroll
            for (int g = 1; g < G; ++g) {
                const int tgt = e0 + (int)(((int64_t)ne * g) / G);
                const int cnt = __popcll(__ballot(lane < nrows && rp_reg < tgt));
                if (grp == g) rbeg = cnt;
                if (grp == g - 1) rend = cnt;
            }
        } else {
            const int per = (nrows + G - 1) / G;
            rbeg = min(grp * per, nrows);
            rend = min(rbeg + per, nrows);
        }
        // (the LDS image was written by this wave only; ds operations of one wave complete in order)
        int r = rbeg;
        int e = rbeg < nrows ? rp_s[rbeg] : ne;
        const int ge = rend < nrows ? rp_s[rend] : ne;
        int row_end = rp_s[r + 1];  // entries past the item's rows hold ne
        Vec<VW> acc;
        acc.zero();
        while (e < ge) {
            Vec<VW> x[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = e + u < ge;
                const int c = ok ? col_s[e + u] : 0;
                v[u] = ok ? val_s[e + u] : 0.f;
                x[u].zero();
                if (ok && col_ok) x[u].load(Xc + (int64_t)c * ldx);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e + u < ge) {
                    while (e + u >= row_end) {  // row r is complete (possibly empty): flush
                        if (col_ok) acc.template store_out<NT>(Y + (int64_t)(r0 + r) * ldy + coff);
                        acc.zero();
                        ++r;
                        row_end = rp_s[r + 1];
                    }
                    acc.fma(v[u], x[u]);
                }
            }
            e += U;
        }
        for (; r < rend; ++r) {  // the row in progress, then trailing empty rows
            if (col_ok) acc.template store_out<NT>(Y + (int64_t)(r0 + r) * ldy + coff);
            acc.zero();
        }
        return;
    }
    // ---- row mode ----
#ifdef GLASS_K1_TRACE
    if (__builtin_amdgcn_readfirstlane(rp_reg) >= 0) K1_STAMP(2);
#endif
    int es = e0;
    for (int i = 0; i < nrows; ++i) {
        const int ee = (i + 1 < nrows) ? __builtin_amdgcn_readlane(rp_reg, i + 1) : e1;
        Vec<VW> acc;
        acc.zero();
        gather_edges<VW, LPR, U, NT>(acc, col, val, Xc, ldx, es, ee, lane, grp, col_ok);
        reduce_groups<VW, LPR>(acc);
        if (grp == 0 && col_ok) acc.template store_out<NT>(Y + (int64_t)(r0 + i) * ldy + coff);
        es = ee;
    }
    K1_STAMP(3);
}

// ---- reduce kernel: rows cut into several chunks: sum their partial rows in slot order --------
__global__ __launch_bounds__(kBlock) void spmm_reduce_kernel(const float* __restrict__ partials, float* __restrict__ Y,
                                                             int64_t ldy, int H, const int32_t* __restrict__ rrows) {
    const int32_t* rr = rrows + 3 * (int64_t)blockIdx.x;
    const int row = rr[0], first = rr[1], n = rr[2];
    for (int c = threadIdx.x; c < H; c += kBlock) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += partials[(int64_t)(first + k) * H + c];
        Y[(int64_t)row * ldy + c] = s;
    }
}

template <int VW, int LPR, int U, bool NT>
static int launch_spmm_u(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t ldx,
                       float* Y, int64_t ldy, int64_t H, const int32_t* hdr, const int32_t* plan, float* ws,
                       hipStream_t st) {
    const int n_ctiles = (int)ceil_div(H, (int64_t)LPR * VW);
    const int n_waves = hdr[H_NSWEEP];
    if (n_waves > 0) {
        const int n_sweep_blocks = (int)ceil_div(n_waves, kBlock / kWave);
        dim3 grid((unsigned)(n_sweep_blocks + hdr[H_NLONG]), n_ctiles);  // sweep workgroups, then one per long-row item
        constexpr int G = kWave / LPR;
        int g_log2 = 0;
        while ((1 << g_log2) < G) ++g_log2;
        const int share15 = (hdr[H_FLAT_SHARE] >> (4 * g_log2)) & 15;  // fifteenths of the sweep cost in flat-eligible items
        if (G > 1 && n_waves >= kFlatMinItems && share15 >= kFlatMinShare15)
            hipLaunchKernelGGL((spmm_sweep_kernel<VW, LPR, U, NT, true>), grid, dim3(kBlock), 0, st, rowptr, col, val, X, ldx,
                               Y, ldy, (int)H, plan + hdr[H_OFF_SWEEP], n_waves, hdr[H_RP_FACTOR], plan + hdr[H_OFF_LONG],
                               n_sweep_blocks, ws);
        else
            hipLaunchKernelGGL((spmm_sweep_kernel<VW, LPR, U, NT, false>), grid, dim3(kBlock), 0, st, rowptr, col, val, X, ldx,
                               Y, ldy, (int)H, plan + hdr[H_OFF_SWEEP], n_waves, 0, plan + hdr[H_OFF_LONG], n_sweep_blocks, ws);
    } else if (hdr[H_NLONG] > 0) {
        dim3 grid((unsigned)hdr[H_NLONG], n_ctiles);
        hipLaunchKernelGGL((spmm_long_kernel<VW, LPR, U, NT>), grid, dim3(kBlock), 0, st, col, val, X, ldx, Y, ldy, ws,
                           (int)H, plan + hdr[H_OFF_LONG]);
    }
    if (hdr[H_NREDUCE] > 0) {
        hipLaunchKernelGGL(spmm_reduce_kernel, dim3((unsigned)hdr[H_NREDUCE]), dim3(kBlock), 0, st, ws, Y, ldy, (int)H,
                           plan + hdr[H_OFF_REDUCE]);
    }
    return launch_status("glass_spmm_csr_f32");
}

template <int VW, int LPR>
static int launch_spmm(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t ldx,
                       float* Y, int64_t ldy, int64_t H, const int32_t* hdr, const int32_t* plan, float* ws,
                       hipStream_t st) {
    // gathers in flight per lane group: measured us/launch for U = 2 / 4 / 8 on MI355X — ppi_bp-shape
    // 14.9 / 13.2 / 12.7, hpo_neuro-shape 87.8 / 83.4 / 81.0, power-law H=256 2638 / 2600 / 2597.
    // streamed bytes of this launch (indices in, Y out) beyond what the Infinity Cache holds -> non-temporal streams
    const int64_t streamed = (int64_t)hdr[H_NNZ] * 8 + (int64_t)hdr[H_NROWS] * (4 * H + 4);
    if (streamed > kStreamNtBytes) return launch_spmm_u<VW, LPR, kGathers, true>(rowptr, col, val, X, ldx, Y, ldy, H, hdr, plan, ws, st);
    return launch_spmm_u<VW, LPR, kGathers, false>(rowptr, col, val, X, ldx, Y, ldy, H, hdr, plan, ws, st);
}

}  // namespace glass

using namespace glass;

#ifdef GLASS_K1_TRACE
extern "C" int glass_k1_trace_set(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k1_trace), &p, sizeof(p));
}
#endif

// Plan policy (host).  Rows with >= kLongThr edges are given to whole workgroups in chunks of
// kLongChunk edges; the rest are swept by waves holding ~edges_per_wave edges each.
// Matrices with few, long rows (the transposed one-hot selection matrix of the embedding backward:
// V ~ 50 rows holding N entries) would keep most of the chip idle under that policy (47 workgroups,
// 26 us at ppi_bp-shape), so when the whole matrix yields fewer than ~1024 chunks the threshold and
// the chunk shrink until it does (never below one 64-edge batch per wave).  Such matrices (<= 1024 rows averaging
// >= 64 entries) also skip the sweep kernel altogether: EVERY row becomes workgroup items (an empty row one empty
// item that stores zeros), so the product is two launches (items + reduce) instead of three.
#ifndef GLASS_K1_SMALL_COST
#define GLASS_K1_SMALL_COST (2ll << 20)   // edges + 4 per row: below this a product is latency-bound (plan policy)
#endif
#ifndef GLASS_K1_LONG_THR
#define GLASS_K1_LONG_THR 256
#endif
static constexpr int kLongThrMax = GLASS_K1_LONG_THR;
static constexpr int kLongChunkMax = 2048;
static constexpr int kRowCost = 4;        // per-row overhead in edge units (rowptr read, reduce, store)
#ifndef GLASS_K1_TARGET_WAVES
#define GLASS_K1_TARGET_WAVES 32768
#endif
static constexpr int kTargetWaves = GLASS_K1_TARGET_WAVES;  // ~4 rounds of 256 CUs x 32 waves
static constexpr int kMinBudget = 16;        // smallest item budget: on small graphs one degree-12 row per wave (density-shape: 5.0 -> 4.0 us; 32 packed two)
static constexpr int kFlatFactor = 4;       // flat mode up to a mean degree of 4 G (uniform degree 12 at H = 64: 988 -> 931 us; degree 37: row mode)

extern "C" int glass_spmm_plan_build(const int32_t* rowptr, int64_t n_rows, int32_t* plan, int64_t* plan_words) {
    GLASS_REQUIRE(rowptr && plan_words && n_rows >= 0 && n_rows < (1ll << 31), "plan_build: bad arguments");
    const int64_t nnz = rowptr[n_rows];
    GLASS_REQUIRE(rowptr[0] == 0 && nnz >= 0, "plan_build: rowptr[0] must be 0 and nnz < 2^31");
    std::vector<int32_t> sweep, longs, reduces;
    int kLongThr = kLongThrMax, kLongChunk = kLongChunkMax;
    while (kLongChunk > 256) {
        int64_t items = 0;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t d = (int64_t)rowptr[r + 1] - rowptr[r];
            if (d >= kLongThr) items += ceil_div(d, kLongChunk);
        }
        if (items + n_rows >= 1024) break;  // enough independent work items
        kLongChunk /= 2;
        if (kLongThr > 64) kLongThr /= 2;
    }
    // Small products (a few generations of resident waves at most) are latency-bound: the launch lasts as long as its
    // longest wave, and a wave walks d edges in d/32 dependent rounds.  There a row of >= 64 edges is worth a workgroup
    // (4 waves share it; the long-row items ride in the sweep launch, so they cost no launch of their own) and a chunk
    // is 512 edges (2 batches per wave; longer rows pay the reduce launch instead of a 16-round chain).  Measured,
    // us per product at H = 64 (2048-edge chunks / threshold 256 -> this): shipped density graph 12.8 -> 6.1; N = 4 998
    // Zipf 0.8 19.4 -> 8.1; N = 17 080, 634 k edges, Zipf 0.5 31.6 -> 16.3; N = 14 587, 1.2 M edges, Zipf 0.7 35.2 -> 24.7.
    // Throughput-bound products keep the large chunks (hpo_neuro-shape, 6.5 M edges: 75.8 vs 77.8 with 512).
    if (nnz + 4 * n_rows <= GLASS_K1_SMALL_COST) {
        if (kLongThr > 64) kLongThr = 64;
        if (kLongChunk > 512) kLongChunk = 512;
    }
    const bool all_long = n_rows > 0 && n_rows <= 1024 && nnz >= 64 * n_rows;
    if (all_long) kLongThr = 0;
    int64_t cost_total = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t d = (int64_t)rowptr[r + 1] - rowptr[r];
        GLASS_REQUIRE(d >= 0, "plan_build: rowptr not monotone at row %lld", (long long)r);
        cost_total += kRowCost + (d < kLongThr ? d : 0);
    }
    int64_t budget = cost_total / kTargetWaves;
    if (budget < kMinBudget) budget = kMinBudget;
    if (budget > 1024) budget = 1024;
    // Sweep items (r0, r1, e0, e1): maximal runs of consecutive SHORT rows under three caps — the cost budget (balance),
    // kItemRows rows and kItemEdges edges (what the kernel stages per wave).  Long rows belong to the workgroup kernel
    // and end the current item, so an item never contains one.
    int64_t acc = 0, it_r0 = 0, it_edges = 0;
    int32_t n_slots = 0;
    bool open = false;
    int64_t min_item_deg = INT32_MAX;  // min over sweep items of ceil(edges / rows)
    int64_t flat_cost[7] = {0, 0, 0, 0, 0, 0, 0}, sweep_cost = 0;  // cost in items flat-eligible at G = 1, 2, 4, .. 64 lane groups
    auto close_item = [&](int64_t r_end) {
        if (!open) return;
        const int64_t e = (int64_t)rowptr[r_end] - rowptr[it_r0], nr = r_end - it_r0;
        const int64_t md = ceil_div(e, nr);
        if (md < min_item_deg) min_item_deg = md;
        const int64_t cost = e + kRowCost * nr;
        sweep_cost += cost;
        for (int k = 0; k < 7; ++k)
            if (e <= (int64_t)kFlatFactor * (1 << k) * nr) flat_cost[k] += cost;  // the kernel's flat-mode test
        sweep.push_back((int32_t)it_r0);
        sweep.push_back((int32_t)r_end);
        sweep.push_back(rowptr[it_r0]);
        sweep.push_back(rowptr[r_end]);
        open = false;
    };
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t d = (int64_t)rowptr[r + 1] - rowptr[r];
        if (d < kLongThr) {
            const int64_t c = kRowCost + d;
            if (open && (acc + c > budget || r - it_r0 >= kItemRows || it_edges + d > kItemEdges)) close_item(r);
            if (!open) {
                open = true;
                it_r0 = r;
                acc = it_edges = 0;
            }
            acc += c;
            it_edges += d;
        } else {
            close_item(r);
            const int64_t n_chunks = d > 0 ? ceil_div(d, kLongChunk) : 1;
            const int64_t per = ceil_div(ceil_div(d, n_chunks), kWave) * kWave;  // even chunks, whole batches
            if (n_chunks > 1) {
                reduces.push_back((int32_t)r);
                reduces.push_back(n_slots);
                reduces.push_back((int32_t)n_chunks);
            }
            for (int64_t k = 0; k < n_chunks; ++k) {
                const int64_t b = rowptr[r] + k * per;
                const int64_t e = (b + per < rowptr[r + 1]) ? b + per : rowptr[r + 1];
                longs.push_back((int32_t)r);
                longs.push_back((int32_t)b);
                longs.push_back((int32_t)e);
                longs.push_back(n_chunks > 1 ? n_slots++ : -1);
            }
        }
    }
    close_item(n_rows);
    const int64_t n_sweep = (int64_t)sweep.size() / 4;
    const int64_t off_sweep = GLASS_PLAN_HEADER_WORDS;
    const int64_t off_long = off_sweep + (int64_t)sweep.size();
    const int64_t off_reduce = off_long + (int64_t)longs.size();
    const int64_t total = off_reduce + (int64_t)reduces.size();
    *plan_words = total;
    if (!plan) return 0;
    for (int i = 0; i < GLASS_PLAN_HEADER_WORDS; ++i) plan[i] = 0;
    plan[H_MAGIC] = kPlanMagic;
    plan[H_VER] = kPlanVersion;
    plan[H_NROWS] = (int32_t)n_rows;
    plan[H_NNZ] = (int32_t)nnz;
    plan[H_NSWEEP] = (int32_t)n_sweep;
    plan[H_NLONG] = (int32_t)(longs.size() / 4);
    plan[H_NREDUCE] = (int32_t)(reduces.size() / 3);
    plan[H_NSLOTS] = n_slots;
    plan[H_LONG_THR] = kLongThr;
    plan[H_LONG_CHUNK] = kLongChunk;
    // Flat-mode threshold (mean degree of an item's rows <= factor * G, G = lane groups per wave at the launch's H).
    plan[H_RP_FACTOR] = kFlatFactor;
    plan[H_MIN_ITEM_DEG] = (int32_t)min_item_deg;
    int32_t shares = 0;  // 7 nibbles: floor(15 * share) of the sweep cost that is flat-eligible at G = 2^k
    for (int k = 0; k < 7; ++k) shares |= (int32_t)(sweep_cost > 0 ? 15 * flat_cost[k] / sweep_cost : 0) << (4 * k);
    plan[H_FLAT_SHARE] = shares;
    plan[H_OFF_SWEEP] = (int32_t)off_sweep;
    plan[H_OFF_LONG] = (int32_t)off_long;
    plan[H_OFF_REDUCE] = (int32_t)off_reduce;
    for (size_t i = 0; i < sweep.size(); ++i) plan[off_sweep + i] = sweep[i];
    for (size_t i = 0; i < longs.size(); ++i) plan[off_long + i] = longs[i];
    for (size_t i = 0; i < reduces.size(); ++i) plan[off_reduce + i] = reduces[i];
    return 0;
}

// the plan's reduce step on its own (for callers that ran the items in another launch: linear.hip)
extern "C" int glass_spmm_reduce_rows_f32(const float* partials, float* Y, int64_t ldy, int64_t H, const int32_t* reduce_rows_dev,
                                          int64_t n_reduce, void* stream) {
    GLASS_REQUIRE(partials && Y && reduce_rows_dev && n_reduce > 0 && H > 0 && ldy >= H, "spmm_reduce_rows: bad arguments");
    hipLaunchKernelGGL(spmm_reduce_kernel, dim3((unsigned)n_reduce), dim3(kBlock), 0, (hipStream_t)stream, partials, Y, ldy,
                       (int)H, reduce_rows_dev);
    return launch_status("glass_spmm_reduce_rows_f32");
}

extern "C" int64_t glass_spmm_ws_bytes(const int32_t* hdr, int64_t H) {
    if (!hdr || hdr[H_MAGIC] != kPlanMagic) return GLASS_E_PLAN;
    return (int64_t)hdr[H_NSLOTS] * H * (int64_t)sizeof(float);
}

extern "C" int glass_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                                  int64_t ldx, float* Y, int64_t ldy, int64_t n_rows, int64_t H, const int32_t* hdr,
                                  const int32_t* plan_dev, void* ws, void* stream) {
    GLASS_REQUIRE(rowptr && X && Y && hdr && plan_dev, "spmm: null pointer");
    GLASS_REQUIRE(H > 0 && ldx >= H && ldy >= H, "spmm: need H>0, ldx>=H, ldy>=H (H=%lld ldx=%lld ldy=%lld)",
                  (long long)H, (long long)ldx, (long long)ldy);
    if (hdr[H_MAGIC] != kPlanMagic || hdr[H_VER] != kPlanVersion || hdr[H_NROWS] != n_rows) {
        set_error("spmm: plan does not match (magic %x, rows %d vs %lld)", hdr[H_MAGIC], hdr[H_NROWS],
                  (long long)n_rows);
        return GLASS_E_PLAN;
    }
    GLASS_REQUIRE(hdr[H_NNZ] == 0 || (col && val), "spmm: null col/val");
    GLASS_REQUIRE(hdr[H_NSLOTS] == 0 || ws, "spmm: plan needs %d partial rows but ws is null", hdr[H_NSLOTS]);
    if (n_rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const bool vec = (H % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(X) && aligned16(Y) &&
                     (hdr[H_NSLOTS] == 0 || aligned16(ws));
#define GLASS_SPMM_CASE(VW, LPR) \
    return launch_spmm<VW, LPR>(rowptr, col, val, X, ldx, Y, ldy, H, hdr, plan_dev, wsf, st)
    if (vec) {
        switch (pow2_ceil_cap(H / 4, 64)) {
            case 1: GLASS_SPMM_CASE(4, 1);
            case 2: GLASS_SPMM_CASE(4, 2);
            case 4: GLASS_SPMM_CASE(4, 4);
            case 8: GLASS_SPMM_CASE(4, 8);
            case 16: GLASS_SPMM_CASE(4, 16);
            case 32: GLASS_SPMM_CASE(4, 32);
            default: GLASS_SPMM_CASE(4, 64);
        }
    }
    switch (pow2_ceil_cap(H, 64)) {
        case 1: GLASS_SPMM_CASE(1, 1);
        case 2: GLASS_SPMM_CASE(1, 2);
        case 4: GLASS_SPMM_CASE(1, 4);
        case 8: GLASS_SPMM_CASE(1, 8);
        case 16: GLASS_SPMM_CASE(1, 16);
        case 32: GLASS_SPMM_CASE(1, 32);
        default: GLASS_SPMM_CASE(1, 64);
    }
#undef GLASS_SPMM_CASE
}
