// K5t: the dense half of GLASSConv for WIDE layers on LARGE graphs (hidden 256 / 512: BASELINE config 5, N = 1 M):
// the Linear pair + bias + ELU + label mix (forward, reference impl/models.py:158-162, 167-173) and its data
// gradient as LDS-tiled GEMMs on the fp32 matrix cores, with the same fusions as dense.hip (GraphNorm apply + ELU +
// dropout in the operand staging, GraphNorm statistics in the epilogue, dZ synthesised on the fly, no cat, no [N,2H]
// gradient) — so config 5 runs the step program instead of library GEMMs + stand-alone mix / cat / GraphNorm passes.
//
// Why a second kernel family: dense.hip gives a wave 16 rows and ALL output columns (right where the graph is
// small and every CU gets one row tile); at hidden 256 that is 32 accumulator tiles per wave and a 128 KiB weight
// image per 64-deep K pass, restreamed for every 32 rows — it spilled and lost to hipBLASLt (121 vs 74 ms/step).
// Here: workgroup tile 128 rows x 256 output columns, K step 16, v_mfma_f32_32x32x2_f32; 4 waves as 2 (rows) x 2
// (columns), each 64 x 128 = 2 x 4 MFMA tiles = 128 accumulator registers; both operands go through LDS (double
// buffered, 24 KiB per stage), so a weight byte is fetched once per 128 rows and an activation byte once per 256
// output columns; 64 MFMAs (4 096 cycles per wave) per K step against 6 staging loads and 12 ds_read_b128.
//
// K assignment: the matrix core sums over k in any order, so lane (j = l & 31, h = l >> 5) feeds k = 16 ks + 8 h + s
// at MFMA step s: its 8 values per K step are two contiguous 16-B groups of both operands (plane q = 2h + s/4 of the
// LDS images) -> ds_read_b128 only, conflict-free.  Column assignment: which output column sits in which MFMA column
// slot is a permutation chosen by the weight packing (tiled_col, dense_common.h): the forward pairs the f1 / f0
// halves inside a wave, both kernels give a lane consecutive columns -> wide stores.
#include "dense_common.h"
#include "split_mma.h"

#include <atomic>
#include <type_traits>

// Laboratory build only (-DGLASS_DENSE_TRACE; tools/tiled_trace.py): per-wave wall-clock stamps of the phases of the tiled
// forward (sel 7) / data gradient (sel 8), [workgroup][wave][8 slots] — this translation unit's own trace words (device
// symbols are not linked across translation units).
#ifdef GLASS_DENSE_TRACE
__device__ unsigned long long* g_tiled_trace;
__device__ int g_tiled_trace_sel;
#define T_STAMP(sel, slot)                                                                                         \
    do {                                                                                                           \
        if (g_tiled_trace && g_tiled_trace_sel == (sel) && (threadIdx.x & 63) == 0 && blockIdx.x < 4096)            \
            g_tiled_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64();            \
    } while (0)
extern "C" int glass_tiled_trace_set(unsigned long long* p, int sel) {
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tiled_trace), &p, sizeof(p));
    return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tiled_trace_sel), &sel, sizeof(sel));
}
#else
#define T_STAMP(sel, slot) do { } while (0)
#endif

namespace glass {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTK = 16, kTThreads = 256;
// Tile geometry.  BM rows x BN column slots per workgroup; 4 waves as 2 (rows) x 2 (columns), each BM/2 x BN/2 =
// RB x CB MFMA tiles.  BM = 128 on large graphs (hidden 256 / 512: a weight byte fetched once per 128 rows); BM = 64 at
// hidden 128 (config 4, N = 50 000: 782 instead of 391 workgroups for 256 CUs; at that height the tile is held to 128
// VGPRs and an unpadded 40 KiB of LDS so that four workgroups fit a CU and all 782 are resident at once).
// BN = 256 everywhere except the data gradient with a 128-wide output (hidden 128, trans pair).
template <int BM, int BN>
struct Tile {
    static constexpr int RB = BM / 64, CB = BN / 64;
    static constexpr int kAPlane = BM == 64 ? BM : BM + 2;  // float4 per k-quad plane of the A image (+2, BM = 128: the 4 lanes that stage one row's
                                            // 64 B land in 4 different bank groups; reads are per-row consecutive either way)
    static constexpr int kBPlane = BN;
    static constexpr int kAImg = 4 * kAPlane, kBImg = 4 * kBPlane;  // float4 per stage
    static constexpr int kStageVecs = kAImg + kBImg;                // 128 x 256: 1 544 float4 = 24 704 B
    static constexpr size_t kLds = 2 * (size_t)kStageVecs * sizeof(float4);
    static constexpr int kAPer = BM * 4 / kTThreads, kBPer = BN * 4 / kTThreads;  // staging float4 per thread
};

bool tiled_shape_ok(int64_t H) { return H == 128 || H == 256 || H == 512; }
// shapes whose operand images carry the effective-weight appendix: the comb pair's data gradient at every tiled size, its
// forward where halving the column tiles is possible (hidden 128 has a single one)
bool tiled_eff_dgrad_shape(int64_t H, int64_t n_out) { return tiled_shape_ok(H) && n_out == 2 * H; }
bool tiled_eff_fwd_shape(int64_t H, int64_t K) { return (H == 256 || H == 512) && K == 2 * H; }
// hidden 128: 64-row tiles in the f32-input form (782 workgroups at N = 50 000, four per CU); 128-row tiles in the split form —
// its stages are 60 KiB at 64 rows too, so only two workgroups fit a CU and 782 of them ran as two rounds (the second one a
// third full), each re-streaming the whole 196 KiB weight image for 64 rows; 391 tiles of 128 rows are ONE round with half
// the weight traffic (phase stamps of tools/tiled_trace.py)
// ... when the 64-row tiles would not all be resident at once (more than 512 of them); a smaller graph keeps the finer
// tiles (hidden 96 padded to 128 at N = 17 080: 267 workgroups of 64 rows fill the chip, 134 of 128 rows half of it)
static bool tall128(int64_t N) {
    return lab_knob("GLASS_TILED_H128_ROWS128", 1) && tiled_split_products() && ceil_div(N, 64) > 512;
}
// rows per statistics partial: 64 at hidden 128 in either case (the staged hidden-128 kernels of dense.hip share that geometry;
// a 128-row tile writes the partials of its two row waves separately)
int tiled_rows(int64_t H) { return H == 128 ? 64 : 128; }


// acc[rb][cb] += A(rows wm*BM/2 + rb*32 ..) . B(slots wn*BN/2 + cb*32 ..) over the 16 k of one stage
template <int BM, int BN>
__device__ __forceinline__ void tile_mma(f32x16 (&acc)[BM / 64][BN / 64], const float4* __restrict__ A,
                                         const float4* __restrict__ B, int j, int h, int wm, int wn) {
    using TL = Tile<BM, BN>;
#pragma unroll
    for (int sq = 0; sq < 2; ++sq) {
        const int q = h * 2 + sq;
        float4 a[TL::RB], b[TL::CB];
#pragma unroll
        for (int rb = 0; rb < TL::RB; ++rb) a[rb] = A[q * TL::kAPlane + wm * (BM / 2) + rb * 32 + j];
#pragma unroll
        for (int cb = 0; cb < TL::CB; ++cb) b[cb] = B[q * TL::kBPlane + wn * (BN / 2) + cb * 32 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rb = 0; rb < TL::RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < TL::CB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[rb], e), f4e(b[cb], e), acc[rb][cb], 0, 0, 0);
    }
}

// Stage geometry of either product form (S3: the operands lie in LDS as bf16 pieces, split_mma.h), float4 units; NA = A images
// per stage (2 in the SPLIT data gradient)
template <int BM, int BN, bool S3, int NA = 1>
struct StageGeom {
    static constexpr int kA = S3 ? SplitImg<BM>::kUnits : Tile<BM, BN>::kAImg;
    static constexpr int kB = S3 ? SplitImg<BN>::kUnits : Tile<BM, BN>::kBImg;
    static constexpr int kStage = NA * kA + kB;
    static constexpr size_t kLds = 2 * (size_t)kStage * sizeof(float4);
};

// the same tile product from split operand images: 3 + 3 ds_read_b128 per (row block, column block), 6 MFMAs of 16 k each
template <int BM, int BN>
__device__ __forceinline__ void tile_mma_s(f32x16 (&acc)[BM / 64][BN / 64], const float4* __restrict__ A,
                                           const float4* __restrict__ B, int j, int h, int wm, int wn) {
    constexpr int RB = BM / 64, CB = BN / 64;
    uint4 a[RB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) SplitImg<BM>::frag(A, wm * (BM / 2) + rb * 32 + j, h, a[rb]);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        uint4 b[3];
        SplitImg<BN>::frag(B, wn * (BN / 2) + cb * 32 + j, h, b);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) split_mma(acc[rb][cb], a[rb], b);
    }
}

// Workgroup -> (row tile, column tile).  Blocks are dealt round-robin over the 8 XCDs, so blocks b and b + 8 share an
// L2: the NCT column tiles of one row tile are given ids 8 apart — the second one finds the activation rows in its
// XCD's L2.  (Speed only; any placement is correct.)
__device__ __forceinline__ bool tile_of_block(int nct, int n_rowtiles, int& rt, int& ct) {
    const int b = blockIdx.x, grp = 8 * nct, r = b % grp;
    ct = r / 8;
    rt = (b / grp) * 8 + (r % 8);
    return rt < n_rowtiles;
}

static inline unsigned tiled_grid(int64_t n_rowtiles, int nct) { return (unsigned)(ceil_div(n_rowtiles, 8) * 8 * nct); }

// The weight stage: 4 * BN float4 (PER = 2 or 4 per thread), already in LDS order in the packed image
template <int PER>
struct BStage {
    float4 v[PER];
    __device__ __forceinline__ void issue(const float4* __restrict__ img) {
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = img[threadIdx.x + kTThreads * i];
    }
    __device__ __forceinline__ void commit(float4* __restrict__ B) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) B[threadIdx.x + kTThreads * i] = v[i];
    }
};

// Split form: a K step's B operand is 24 KiB of the pre-cut image (glass_dense_pack_batch_f32 wrote it in LDS order, dense.hip
// pack_cut_put) copied global -> LDS by the DMA path: no registers, no vector instructions, no ds_write.  6 instructions per
// thread; instruction i of wave w lands its 64 x 16 bytes at unit i*256 + w*64 (wave-uniform base + lane x 16).  The copies
// count on vmcnt like loads: they are complete after the s_waitcnt vmcnt(0) in front of the barrier that publishes the stage.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr int kCutTileUnits = 6 * 256;  // 16-byte units per (column tile, K step) of a cut image
__device__ __forceinline__ void dma_b_stage(const uint4* __restrict__ tile, float4* __restrict__ Bimg) {
    const int tid = threadIdx.x;
    const int wbase = __builtin_amdgcn_readfirstlane((tid >> 6) * 64);
#pragma unroll
    for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds(tile + i * 256 + tid, (lds_void_ptr)(uintptr_t)(Bimg + i * 256 + wbase), 16, 0, 0);
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- forward ----------------------------------------------------------------------------------------------------
// EFF (comb pair, hidden 256 / 512): the operand image carries an appendix, W_unl = (1-z) W1 + z W0 ([H][2H], plain
// layout).  The comb pair has no activation before the mix, so a row tile WITHOUT a labeled row is one product
// [g || x_] @ W_unl^T + b_unl over 256 output columns per column tile — half the column tiles of the two-weight form: the
// blocks of the upper half of the column tiles leave at once.  A tile with up to 3 labeled rows runs the same product for
// all its rows and corrects those rows afterwards; beyond that it takes the two-weight path.
template <int H, bool COMB, int BM, bool EFF, bool S3>
__global__ __launch_bounds__(kTThreads, (BM == 64 && !S3) ? 3 : 2) void tiled_fwd_kernel(const float* __restrict__ xa, int64_t lda,
                                                                const float* __restrict__ xb, int64_t ldb,
                                                                const float* __restrict__ Wimg,
                                                                const float* __restrict__ bias,
                                                                const uint8_t* __restrict__ mask, float zr, float omz,
                                                                int act, float* __restrict__ T, int64_t ldt,
                                                                float* __restrict__ out, int64_t ldo, int64_t N,
                                                                double* __restrict__ stats, GnPrologue pro,
                                                                int n_rowtiles, const float* __restrict__ Wcut) {
    using TL = Tile<BM, 256>;  // 256 column slots = 128 columns of the f1 half + the same 128 of the f0 half
    using SG = StageGeom<BM, 256, S3>;
    constexpr int KT = COMB ? 2 * H : H, NKS = KT / kTK, NCT = H / 128, RB = TL::RB, AP = TL::kAPer;
    extern __shared__ float4 smem[];
    int rt, ct;
    if (EFF) {
        // column tile major: the blocks of the upper column tiles — which leave at once for every row tile without a labeled
        // row — sit at the END of the grid instead of alternating with working blocks in groups of 8
        const int per = (int)(gridDim.x / NCT);
        ct = (int)blockIdx.x / per;
        rt = (int)blockIdx.x % per;
        if (rt >= n_rowtiles) return;
    } else if (!tile_of_block(NCT, n_rowtiles, rt, ct)) {
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1, j = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)rt * BM;
    // staging assignment: float4 f = tid + 256 i of the A tile: row f >> 2, k-quad f & 3 (4 lanes cover one row's 64 B)
    int srow[AP], skq[AP];
    bool sok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int f = tid + kTThreads * i;
        srow[i] = f >> 2;
        skq[i] = f & 3;
        sok[i] = row0 + srow[i] < N;
    }
    // operand rows of xa: the tile's own rows, or (trans pair of layer 0) the rows of the embedding table its nodes index —
    // the lookup of input_emb fused into this load (reference impl/models.py:243-244); side output, dropout and statistics
    // stay keyed by the node row
    int64_t arow[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        arow[i] = row0 + srow[i];
        if (!COMB && pro.gather && sok[i]) {
            const int64_t g = pro.gather[arow[i]];
            arow[i] = g < 0 ? 0 : (g >= pro.gather_rows ? pro.gather_rows - 1 : g);
        }
    }
    Drop drop = pro.drop;
    if (pro.saved && drop.p > 0.f) {
        drop.seed = pro.rng_state[0];
        drop.step = pro.rng_state[1];
    }
    // EFF: the single-weight product for the whole tile when it holds at most kMaxFix labeled rows; those rows get their
    // label's weight back afterwards (correction below).  fix[] (LDS behind the stage buffers): [1..4] labeled rows per
    // wave, [8 .. 8+BM) correction slot of a tile row or -1, then the slots' rows.
    constexpr int kMaxFix = 3;  // a correction costs about a third of the second product of the two-weight path
    int* fix = reinterpret_cast<int*>(smem + 2 * SG::kStage);
    bool pure = false;  // workgroup-uniform
    int n_fix = 0;
    if (EFF && act == GLASS_ACT_NONE && T == nullptr) {
        const bool flag = tid < BM && row0 + tid < N && mask[row0 + tid] != 0;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) fix[1 + w] = __popcll(bal);
        if (tid < BM) fix[8 + tid] = -1;
        __syncthreads();
        int off = 0;
        for (int ww = 0; ww < w; ++ww) off += fix[1 + ww];
        n_fix = fix[1] + fix[2] + fix[3] + fix[4];
        pure = n_fix <= kMaxFix;
        if (pure && ct >= NCT / 2) return;  // 256 output columns per block now: the lower half of the column tiles covers H
        if (flag && pure) {
            const int ci = off + __popcll(bal & ((1ull << lane) - 1ull));  // ordered by row
            fix[8 + tid] = ci;
            fix[8 + BM + ci] = tid;
        }
    }
    const bool side_writer = pro.side != nullptr && ct == 0;  // every column tile computes the operand; one writes it
    const float4* wimg = reinterpret_cast<const float4*>(Wimg) +
                         (pure ? (int64_t)(NCT + ct) * NKS * TL::kBImg : (int64_t)ct * NKS * TL::kBImg);
    const uint4* wcut = reinterpret_cast<const uint4*>(Wcut) + (int64_t)(pure ? NCT + ct : ct) * NKS * kCutTileUnits;  // S3

    float4 av[AP], asc, ash;  // (every float4 of a thread sits in the same k-quad — tid & 3 —: one pair of coefficient vectors)
    BStage<TL::kBPer> bs;
    auto issue = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int k = ks * kTK + 4 * skq[i];
            const int64_t r = row0 + srow[i];
            const float* src = (!COMB || k < H) ? xa + (COMB ? r : arow[i]) * lda + k : xb + r * ldb + (k - H);
            av[i] = sok[i] ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (i == 0 && pro.saved && (!COMB || k < H)) {
                asc = *reinterpret_cast<const float4*>(pro.saved + 2 * pro.C + k);
                ash = *reinterpret_cast<const float4*>(pro.saved + 3 * pro.C + k);
            }
        }
        if constexpr (S3) dma_b_stage(wcut + (int64_t)ks * kCutTileUnits, smem + (ks & 1) * SG::kStage + SG::kA);
        else bs.issue(wimg + (int64_t)ks * TL::kBImg);
    };
    auto commit = [&](int ks, float4* stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int k = ks * kTK + 4 * skq[i];
            float4 v = av[i];
            if (pro.saved && (!COMB || k < H) && sok[i]) {
                const int64_t r = row0 + srow[i];
                float ds[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop.p > 0.f) drop_scales<4>(drop, r, k, ds);
                float o[4] = {fmaf(v.x, asc.x, ash.x), fmaf(v.y, asc.y, ash.y), fmaf(v.z, asc.z, ash.z), fmaf(v.w, asc.w, ash.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = act_fast(pro.act, o[e]);
                    o[e] *= ds[e];
                }
                v = make_float4(o[0], o[1], o[2], o[3]);
                if (side_writer) *reinterpret_cast<float4*>(pro.side + r * pro.lds + k) = v;
            }
            if constexpr (S3) SplitImg<BM>::put4(stage, srow[i], skq[i], v);
            else stage[skq[i] * TL::kAPlane + srow[i]] = v;
        }
        if constexpr (S3) dma_drain();  // this stage's B copies (issued a whole K step of MFMAs ago) have landed
        else bs.commit(stage + SG::kA);
    };

    f32x16 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

    T_STAMP(7, 0);
    issue(0);
    commit(0, smem);
    __syncthreads();
    T_STAMP(7, 1);
    for (int ks = 0; ks < NKS; ++ks) {
        float4* cur = smem + (ks & 1) * SG::kStage;
        float4* nxt = smem + ((ks + 1) & 1) * SG::kStage;
        if (ks == 2) T_STAMP(7, 5);
        if (ks + 1 < NKS) issue(ks + 1);  // in flight across this stage's MFMAs
        if constexpr (S3) tile_mma_s<BM, 256>(acc, cur, cur + SG::kA, j, h, wm, wn);
        else tile_mma<BM, 256>(acc, cur, cur + SG::kA, j, h, wm, wn);
        if (ks == 2) T_STAMP(7, 6);
        if (ks + 1 < NKS) commit(ks + 1, nxt);  // the other buffer: last read in stage ks - 1, before the previous barrier
        if (ks == 2) T_STAMP(7, 7);
        __syncthreads();
    }
    T_STAMP(7, 2);

    if (EFF && pure) {
        // A labeled row r is owed  (z - (1-z)) * ( sum_k A[r,k] (W1[c][k] - W0[c][k]) + b1[c] - b0[c] )  in every column c:
        // its label's weight minus the unlabeled one.  One thread per column of this block, A[r,:] rebuilt as the staging
        // code builds it (GraphNorm prologue), W1 / W0 read from the paired main image; added in the epilogue below.
        const float* corr = reinterpret_cast<const float*>(smem);  // [n_fix][256] (the loop's last barrier freed the stages)
        const bool fixing = n_fix > 0;
        if (fixing) {
            float* cw = reinterpret_cast<float*>(smem);
            const int c = ct * 256 + (tid >> 7) * 128 + 4 * (tid & 31) + ((tid >> 5) & 3);  // column of plain slot tid
            // its f1 / f0 slots in the paired image: column tile c / 128, slot wn*128 + cb*32 + j with c % 128 = wn*64 + 2j + (cb & 1)
            const int cp = c & 127;
            const float4* img1 = reinterpret_cast<const float4*>(Wimg) + (int64_t)(c >> 7) * NKS * TL::kBImg +
                                 ((cp >> 6) * 128 + (cp & 1) * 32 + ((cp & 63) >> 1));
            const float4* img0 = img1 + 64;  // cb + 2: the same column of the f0 half
            const float db = bias[c] - bias[H + c];
            float4* arow = reinterpret_cast<float4*>(cw + kMaxFix * 256);  // the operand row, built once by KT / 4 threads
            for (int f = 0; f < n_fix; ++f) {
                const int64_t r = row0 + fix[8 + BM + f];
                for (int kq = tid; kq < KT / 4; kq += kTThreads) {
                    const int k = 4 * kq;
                    float4 v = (k < H) ? *reinterpret_cast<const float4*>(xa + r * lda + k)
                                       : *reinterpret_cast<const float4*>(xb + r * ldb + (k - H));
                    if (pro.saved && k < H) {
                        const float4 sc = *reinterpret_cast<const float4*>(pro.saved + 2 * pro.C + k);
                        const float4 sh = *reinterpret_cast<const float4*>(pro.saved + 3 * pro.C + k);
                        float ds[4] = {1.f, 1.f, 1.f, 1.f};
                        if (drop.p > 0.f) drop_scales<4>(drop, r, k, ds);
                        float o[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = act_fast(pro.act, o[e]);
                            o[e] *= ds[e];
                        }
                        v = make_float4(o[0], o[1], o[2], o[3]);
                    }
                    arow[kq] = v;
                }
                __syncthreads();
                float sacc = 0.f;
#pragma unroll 4
                for (int kq = 0; kq < KT / 4; ++kq) {
                    const float4 v = arow[kq];
                    const float4 w1v = img1[(int64_t)kq * 256], w0v = img0[(int64_t)kq * 256];
                    sacc += v.x * (w1v.x - w0v.x) + v.y * (w1v.y - w0v.y) + v.z * (w1v.z - w0v.z) + v.w * (w1v.w - w0v.w);
                }
                cw[f * 256 + tid] = (zr - omz) * (sacc + db);
                __syncthreads();  // arow is rebuilt for the next row; after the last one: corrections visible
            }
        }
        // single-weight epilogue: plain layout — this lane's four consecutive columns col0 .. col0 + 3 (cb = 0..3)
        const int col0 = ct * 256 + wn * 128 + 4 * j;
        const float4 c1 = *reinterpret_cast<const float4*>(bias + col0), c0 = *reinterpret_cast<const float4*>(bias + H + col0);
        const float be[4] = {omz * c1.x + zr * c0.x, omz * c1.y + zr * c0.y, omz * c1.z + zr * c0.z, omz * c1.w + zr * c0.w};
        double ss[4] = {0.0, 0.0, 0.0, 0.0}, qq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t r = row0 + wm * (BM / 2) + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                if (r < N) {
                    float o[4];
                    const int ci = fixing ? fix[8 + wm * (BM / 2) + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3)] : -1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = acc[rb][e][i] + be[e];
                        if (ci >= 0) o[e] += corr[ci * 256 + wn * 128 + e * 32 + j];
                        ss[e] += (double)o[e];
                        qq[e] += (double)o[e] * (double)o[e];
                    }
                    *reinterpret_cast<float4*>(out + r * ldo + col0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        if (stats == nullptr) return;
        double* red = reinterpret_cast<double*>(smem);  // [wm][256 columns][2]
        if (fixing) __syncthreads();  // `red` lies over the corrections other waves may still be reading
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ss[e] += __shfl_xor(ss[e], 32);
            qq[e] += __shfl_xor(qq[e], 32);
        }
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = wn * 128 + 4 * j + e;
                red[(wm * 256 + c) * 2] = ss[e];
                red[(wm * 256 + c) * 2 + 1] = qq[e];
            }
        }
        __syncthreads();
        stats[((size_t)rt * 2) * H + ct * 256 + tid] = red[tid * 2] + red[(256 + tid) * 2];
        stats[((size_t)rt * 2 + 1) * H + ct * 256 + tid] = red[tid * 2 + 1] + red[(256 + tid) * 2 + 1];
        return;
    }
    // epilogue: acc[rb][cb][i] = row rt*BM + wm*BM/2 + rb*32 + 8(i>>2) + 4h + (i&3); cb 0,1: f1 columns colp, colp+1;
    // cb 2,3: the same two columns of the f0 half
    const int colp = ct * 128 + wn * 64 + 2 * j;
    const float2 b1 = *reinterpret_cast<const float2*>(bias + colp);
    const float2 b0 = *reinterpret_cast<const float2*>(bias + H + colp);
    double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;  // column sums of `out` over this lane's rows (GraphNorm that follows)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row0 + wm * (BM / 2) + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            if (r < N) {
                const bool lab = mask[r] != 0;
                const float w1 = lab ? zr : omz, w0 = lab ? omz : zr;
                const float v1x = acc[rb][0][i] + b1.x, v1y = acc[rb][1][i] + b1.y;
                const float v0x = acc[rb][2][i] + b0.x, v0y = acc[rb][3][i] + b0.y;
                if (T) {
                    *reinterpret_cast<float2*>(T + r * ldt + colp) = make_float2(v1x, v1y);
                    *reinterpret_cast<float2*>(T + r * ldt + H + colp) = make_float2(v0x, v0y);
                }
                float a1x = v1x, a1y = v1y, a0x = v0x, a0y = v0y;
                a1x = act_fast(act, a1x), a1y = act_fast(act, a1y), a0x = act_fast(act, a0x), a0y = act_fast(act, a0y);
                const float ox = w1 * a1x + w0 * a0x, oy = w1 * a1y + w0 * a0y;
                *reinterpret_cast<float2*>(out + r * ldo + colp) = make_float2(ox, oy);
                s0 += (double)ox; q0 += (double)ox * (double)ox;
                s1 += (double)oy; q1 += (double)oy * (double)oy;
            }
        }
    T_STAMP(7, 3);
    if (stats == nullptr) return;
    // stats[rt][2][H]: sum / sum of squares of this row tile, columns of this column tile (fp64 from the first add)
    s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
    double* red = reinterpret_cast<double*>(smem);  // [wm][128 columns][2]; the last barrier of the loop freed the stages
    if (h == 0) {
        const int c = wn * 64 + 2 * j;
        red[(wm * 128 + c) * 2] = s0; red[(wm * 128 + c) * 2 + 1] = q0;
        red[(wm * 128 + c + 1) * 2] = s1; red[(wm * 128 + c + 1) * 2 + 1] = q1;
    }
    __syncthreads();
    if (H == 128 && BM == 128) {  // partials per 64 rows (tiled_rows): one per row wave
        if (tid < 128) {
#pragma unroll
            for (int hw = 0; hw < 2; ++hw) {
                const int64_t part = 2 * (int64_t)rt + hw;
                if (part * 64 < N) {
                    stats[((size_t)part * 2) * H + ct * 128 + tid] = red[(hw * 128 + tid) * 2];
                    stats[((size_t)part * 2 + 1) * H + ct * 128 + tid] = red[(hw * 128 + tid) * 2 + 1];
                }
            }
        }
    } else if (tid < 128) {
        const double s = red[tid * 2] + red[(128 + tid) * 2], q = red[tid * 2 + 1] + red[(128 + tid) * 2 + 1];
        stats[((size_t)rt * 2) * H + ct * 128 + tid] = s;
        stats[((size_t)rt * 2 + 1) * H + ct * 128 + tid] = q;
    }
    T_STAMP(7, 4);
}

// ---- backward data gradient ---------------------------------------------------------------------------------------
// out[N, NOUT] = dZ[N, 2H] @ Wstack[2H, NOUT] (+ addend)(* dropout mask), dZ[n, o] = coef(n, o < H) * dsrc[n, o mod H]
// * act'(T[n, o]) synthesised while staging; WTimg = Wstack^T packed plain-tiled: a lane holds four consecutive output
// columns (cb = 0..3).
// SPLIT — kept for reference, NOT instantiated since round 6 (hidden 128's trans pair runs trans_dgrad2 / trans_dgrad3 of dense.hip) —
// (hidden 128, trans pair: a 128-wide output would fill only half of the 256-slot column tile): the product is
// dZ1 @ W1 + dZ0 @ W0 with both terms [N, 128] — the two wave columns of the workgroup take one term each: the left
// pair of waves multiplies the f1 half (its own A image) into column slots 0..127, the right pair the f0 half into slots
// 128..255, K is H instead of 2H, and the right pair hands its accumulators to the left through LDS before the epilogue.
// Operand image: layout kLayoutTiledSplit (slot s < 128: Wstack[k][s], else Wstack[H + k][s - 128]).
// EFF (comb pair): the operand image carries an appendix, the effective weight of UNLABELED rows
// (1-z) W1 + z W0 over K = H (kLayoutTiledPlainEff).  dZ has no activation factor for the comb pair, so an unlabeled row of
// the product is dc @ W_unl — half the K loop.  A tile runs that for ALL its rows and then corrects its (few) labeled rows
// with a thread-per-column dot product; only tiles with more than 7 labeled rows fall back to the two-term product.
template <int H, int NOUT, int BM, int BN, bool SPLIT, bool EFF, bool S3>
__global__ __launch_bounds__(kTThreads, S3 ? 2 : (SPLIT ? 3 : (BM == 64 ? 4 : 2))) void tiled_dgrad_kernel(const float* __restrict__ dsrc, int64_t ldd,
                                                                  const float* __restrict__ T, int64_t ldt,
                                                                  const uint8_t* __restrict__ mask, float zr, float omz,
                                                                  int act, const float* __restrict__ WTimg,
                                                                  const float* __restrict__ addend, int64_t ldadd,
                                                                  Drop drop, const uint64_t* __restrict__ rng_state,
                                                                  float* __restrict__ out, int64_t ldo, int64_t N,
                                                                  GnBwdStats gs, int n_rowtiles, const float* __restrict__ Wcut) {
    using TL = Tile<BM, BN>;
    constexpr int KT = SPLIT ? H : 2 * H, NKS = KT / kTK, NCT = SPLIT ? 1 : NOUT / BN, RB = TL::RB, CB = TL::CB, AP = TL::kAPer;
    constexpr int NA = SPLIT ? 2 : 1;                          // A images per stage
    using SG = StageGeom<BM, BN, S3, NA>;
    constexpr int kStage = SG::kStage;                         // float4 per stage
    static_assert(!S3 || BN == 256, "split operand images: 256-slot column tiles");
    static_assert(CB == 4 && (SPLIT ? (NOUT == H && 2 * NOUT == BN) : NOUT % BN == 0),
                  "output width must be a multiple of the 256-slot column tile (SPLIT: exactly half of one)");
    extern __shared__ float4 smem[];
    int rt, ct;
    if (!tile_of_block(NCT, n_rowtiles, rt, ct)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1, j = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)rt * BM;
    int srow[AP], skq[AP];
    bool sok[AP], slab[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int f = tid + kTThreads * i;
        srow[i] = f >> 2;
        skq[i] = f & 3;
        sok[i] = row0 + srow[i] < N;
        slab[i] = sok[i] && mask[row0 + srow[i]] != 0;
    }
    // EFF: the single-term product (appendix image, K = H) for the whole tile when it holds at most kMaxFix labeled rows;
    // those rows get their label's weight back afterwards as a correction (below).  fix[] (LDS behind the stage
    // buffers): [1..4] labeled rows per wave, [8 .. 8+BM) correction slot of a tile row or -1, then the slots' rows.
    constexpr int kMaxFix = 7;  // a correction costs about an eighth of the second term of the two-term product
    int* fix = reinterpret_cast<int*>(smem + 2 * kStage);
    bool pure = false;  // workgroup-uniform
    int n_fix = 0;
    if (EFF && act == GLASS_ACT_NONE) {
        const bool flag = tid < BM && row0 + tid < N && mask[row0 + tid] != 0;
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) fix[1 + w] = __popcll(bal);
        if (tid < BM) fix[8 + tid] = -1;
        __syncthreads();
        int off = 0;
        for (int ww = 0; ww < w; ++ww) off += fix[1 + ww];
        n_fix = fix[1] + fix[2] + fix[3] + fix[4];
        pure = n_fix <= kMaxFix;
        if (flag && pure) {
            const int ci = off + __popcll(bal & ((1ull << lane) - 1ull));  // ordered by row
            fix[8 + tid] = ci;
            fix[8 + BM + ci] = tid;
        }
    }
    const float4* wimg = reinterpret_cast<const float4*>(WTimg) +
                         (pure ? (int64_t)NCT * NKS * TL::kBImg + (int64_t)ct * (NKS / 2) * TL::kBImg : (int64_t)ct * NKS * TL::kBImg);
    const uint4* wcut = reinterpret_cast<const uint4*>(Wcut) +
                        (pure ? (int64_t)NCT * NKS + (int64_t)ct * (NKS / 2) : (int64_t)ct * NKS) * kCutTileUnits;  // S3
    float4 dv[AP], tv[AP], tw[SPLIT ? AP : 1];
    BStage<TL::kBPer> bs;
    auto issue = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int o = ks * kTK + 4 * skq[i];  // column of dZ (SPLIT: of both halves)
            const int64_t r = row0 + srow[i];
            dv[i] = sok[i] ? *reinterpret_cast<const float4*>(dsrc + r * ldd + (o < H ? o : o - H)) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (act != GLASS_ACT_NONE) {
                tv[i] = sok[i] ? *reinterpret_cast<const float4*>(T + r * ldt + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (SPLIT) tw[i] = sok[i] ? *reinterpret_cast<const float4*>(T + r * ldt + H + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (S3) dma_b_stage(wcut + (int64_t)ks * kCutTileUnits, smem + (ks & 1) * kStage + NA * SG::kA);
        else bs.issue(wimg + (int64_t)ks * TL::kBImg);
    };
    auto commit = [&](int ks, float4* stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int o = ks * kTK + 4 * skq[i];
            const float coef = sok[i] ? (pure ? 1.f : ((slab[i] == (o < H)) ? zr : omz)) : 0.f;
            float4 v = make_float4(dv[i].x * coef, dv[i].y * coef, dv[i].z * coef, dv[i].w * coef);
            if (act != GLASS_ACT_NONE) {
                v.x *= act_grad(act, tv[i].x); v.y *= act_grad(act, tv[i].y); v.z *= act_grad(act, tv[i].z); v.w *= act_grad(act, tv[i].w);
            }
            if constexpr (S3) SplitImg<BM>::put4(stage, srow[i], skq[i], v);
            else stage[skq[i] * TL::kAPlane + srow[i]] = v;
            if (SPLIT) {  // the f0 half of the same dsrc columns: the other label coefficient, the other half of T
                const float c0 = sok[i] ? (slab[i] ? omz : zr) : 0.f;
                float4 u = make_float4(dv[i].x * c0, dv[i].y * c0, dv[i].z * c0, dv[i].w * c0);
                if (act != GLASS_ACT_NONE) {
                    u.x *= act_grad(act, tw[i].x); u.y *= act_grad(act, tw[i].y); u.z *= act_grad(act, tw[i].z); u.w *= act_grad(act, tw[i].w);
                }
                if constexpr (S3) SplitImg<BM>::put4(stage + SG::kA, srow[i], skq[i], u);
                else stage[TL::kAImg + skq[i] * TL::kAPlane + srow[i]] = u;
            }
        }
        if constexpr (S3) dma_drain();
        else bs.commit(stage + NA * SG::kA);
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

    // the K loop with a compile-time trip count (a run-time bound sent the staging registers to scratch memory)
    auto k_loop = [&](auto n_c) __attribute__((always_inline)) {
        constexpr int n = decltype(n_c)::value;
        T_STAMP(8, 0);
        issue(0);
        commit(0, smem);
        __syncthreads();
        T_STAMP(8, 1);
        for (int ks = 0; ks < n; ++ks) {
            float4* cur = smem + (ks & 1) * kStage;
            float4* nxt = smem + ((ks + 1) & 1) * kStage;
            if (ks + 1 < n) issue(ks + 1);
            if constexpr (S3) tile_mma_s<BM, BN>(acc, cur + (SPLIT ? wn * SG::kA : 0), cur + NA * SG::kA, j, h, wm, wn);
            else tile_mma<BM, BN>(acc, cur + (SPLIT ? wn * SG::kA : 0), cur + NA * SG::kA, j, h, wm, wn);
            if (ks + 1 < n) commit(ks + 1, nxt);
            __syncthreads();
        }
    };
    if (EFF && pure) k_loop(std::integral_constant<int, NKS / 2>{});
    else k_loop(std::integral_constant<int, NKS>{});
    T_STAMP(8, 2);
    if (SPLIT) {  // right wave column -> left wave column (same row block, same lane <-> same rows and column slots)
        float* xf = reinterpret_cast<float*>(smem);
        if (wn == 1) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) xf[(((wm * RB + rb) * CB + cb) * 16 + i) * 64 + lane] = acc[rb][cb][i];
        }
        __syncthreads();
        if (wn == 0) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[rb][cb][i] += xf[(((wm * RB + rb) * CB + cb) * 16 + i) * 64 + lane];
        }
        __syncthreads();  // the statistics below reuse this memory
    }
    const bool active = !SPLIT || wn == 0;  // waves that own output columns
    // EFF: a labeled row r of a single-term tile is owed  (z - (1-z)) * sum_k dc[r,k] (B[c][k] - B[c][H+k])  in every
    // column c — its label's weight minus the unlabeled one.  One thread per column slot, the main image read as it lies
    // (consecutive threads, consecutive float4); the row's owner lanes add it in the epilogue.
    const float* corr = reinterpret_cast<const float*>(smem);  // [n_fix][256 slots] (the loop's last barrier freed the stages)
    const bool fixing = EFF && pure && n_fix > 0;
    if (fixing) {
        const float4* main_img = reinterpret_cast<const float4*>(WTimg) + (int64_t)ct * NKS * TL::kBImg;
        float* cw = reinterpret_cast<float*>(smem);
        for (int f = 0; f < n_fix; ++f) {
            const float* drow = dsrc + (row0 + fix[8 + BM + f]) * ldd;
            float sacc = 0.f;
#pragma unroll 4
            for (int kq = 0; kq < H / 4; ++kq) {
                const float4 d = *reinterpret_cast<const float4*>(drow + 4 * kq);
                const float4 b1 = main_img[(int64_t)kq * 256 + tid], b0 = main_img[(int64_t)(kq + H / 4) * 256 + tid];
                sacc += d.x * (b1.x - b0.x) + d.y * (b1.y - b0.y) + d.z * (b1.z - b0.z) + d.w * (b1.w - b0.w);
            }
            cw[f * 256 + tid] = (zr - omz) * sacc;
        }
        __syncthreads();
    }

    // epilogue: this lane's four consecutive output columns col0 .. col0 + 3 (cb = 0..3) of its 16 * RB rows
    const int col0 = ct * BN + wn * (BN / 2) + CB * j;
    if (drop.p > 0.f) {
        drop.seed = rng_state[0];
        drop.step = rng_state[1];
    }
    const bool gn_half = active && gs.partial != nullptr && col0 < H;  // wave-uniform (H is a multiple of 128 >= BN / 2)
    float g_mu[CB], g_rstd[CB], g_scale[CB], g_shift[CB], g_al[CB];
    double s1[CB], s2[CB];
#pragma unroll
    for (int e = 0; e < CB; ++e) s1[e] = s2[e] = 0.0;
    if (gn_half) {
        if (gs.drop.p > 0.f) {
            gs.drop.seed = rng_state[0];
            gs.drop.step = rng_state[1];
        }
#pragma unroll
        for (int e = 0; e < CB; ++e) {
            g_mu[e] = gs.saved[col0 + e];
            g_rstd[e] = gs.saved[H + col0 + e];
            g_scale[e] = gs.saved[2 * H + col0 + e];
            g_shift[e] = gs.saved[3 * H + col0 + e];
            g_al[e] = gs.alpha[col0 + e];
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row0 + wm * (BM / 2) + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            if (active && r < N) {
                float v[CB];
#pragma unroll
                for (int e = 0; e < CB; ++e) v[e] = acc[rb][e][i];
                if (fixing) {
                    const int ci = fix[8 + wm * (BM / 2) + rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3)];
                    if (ci >= 0) {
#pragma unroll
                        for (int e = 0; e < CB; ++e) v[e] += corr[ci * 256 + wn * (BN / 2) + e * 32 + j];
                    }
                }
                if (addend) {
                    const float4 ad = *reinterpret_cast<const float4*>(addend + r * ldadd + col0);
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
                if (drop.p > 0.f) {  // gradient w.r.t. the pre-dropout tensor: same mask as the forward drew
                    float ds[4];
                    drop_scales<4>(drop, r, col0, ds);
#pragma unroll
                    for (int e = 0; e < CB; ++e) v[e] *= ds[e];
                }
                *reinterpret_cast<float4*>(out + r * ldo + col0) = make_float4(v[0], v[1], v[2], v[3]);
                if (gn_half) {
                    const float4 x4 = *reinterpret_cast<const float4*>(gs.x + r * gs.ldx + col0);
                    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
                    float ds[4] = {1.f, 1.f, 1.f, 1.f};
                    if (gs.drop.p > 0.f) drop_scales<4>(gs.drop, r, col0, ds);
#pragma unroll
                    for (int e = 0; e < CB; ++e) {
                        float gp = v[e] * ds[e];
                        if (gs.act != GLASS_ACT_NONE) gp *= act_grad(gs.act, fmaf(xv[e], g_scale[e], g_shift[e]));
                        const float xhat = (xv[e] - g_al[e] * g_mu[e]) * g_rstd[e];
                        s1[e] += (double)gp;
                        s2[e] += (double)gp * (double)xhat;
                    }
                }
            }
        }
    T_STAMP(8, 3);
    if (gs.partial == nullptr) return;
    // partial[rt][2][H] over this row tile: lanes h = 0/1, then the two row waves through LDS
    double* red = reinterpret_cast<double*>(smem);  // [wm][BN columns][2]
    if (fixing) __syncthreads();  // `red` lies over the corrections other waves may still be reading
    if (gn_half) {
#pragma unroll
        for (int e = 0; e < CB; ++e) {
            s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 32);
        }
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < CB; ++e) {
                const int c = wn * (BN / 2) + CB * j + e;
                red[(wm * BN + c) * 2] = s1[e];
                red[(wm * BN + c) * 2 + 1] = s2[e];
            }
        }
    }
    __syncthreads();
    const int c = ct * BN + tid;  // threads 0 .. BN-1 <-> the BN columns of this tile
    if (H == 128 && BM == 128) {  // partials per 64 rows (tiled_rows): one per row wave
        if (tid < BN && c < H) {
#pragma unroll
            for (int hw = 0; hw < 2; ++hw) {
                const int64_t part = 2 * (int64_t)rt + hw;
                if (part * 64 < N) {
                    gs.partial[((size_t)part * 2) * H + c] = red[(hw * BN + tid) * 2];
                    gs.partial[((size_t)part * 2 + 1) * H + c] = red[(hw * BN + tid) * 2 + 1];
                }
            }
        }
    } else if (tid < BN && c < H) {
        gs.partial[((size_t)rt * 2) * H + c] = red[tid * 2] + red[(BN + tid) * 2];
        gs.partial[((size_t)rt * 2 + 1) * H + c] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
    }
}

template <typename K>
static void allow_tiled_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Product form of the tiled kernels: six bf16 partial products per fp32 product (split_mma.h; default) or the f32-input
// matrix-core instruction (1/16 of the bf16 rate — the reference point of the accuracy and A/B runs): an option of the CALL
// (GLASS_DENSE_F32_PRODUCTS in the entry's `act` word; dense_common.h CallOptions), never process state.
thread_local int t_call_options = 0;

template <int HH, int BM, bool S3>
static void tiled_fwd_launch(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg, const float* bias,
                             const uint8_t* mask, float zr, float omz, int act, float* T, int64_t ldt, float* out,
                             int64_t ldo, int64_t N, double* stats, const GnPrologue& pro, hipStream_t st) {
    const bool comb = xb != nullptr;
    const int64_t n_rt = ceil_div(N, BM);
    const dim3 grid(tiled_grid(n_rt, HH / 128));
    // + the labeled-row bookkeeping of the effective-weight path (hidden 256 / 512, comb pair)
    const size_t lab_pad = (size_t)lab_knob("GLASS_TILED_LDS_PAD", 0);  // laboratory: bytes of LDS padding lower the workgroups per CU
    const size_t lds = StageGeom<BM, 256, S3>::kLds + ((comb && HH >= 256) ? 1024 : 0) + lab_pad;
    // the cut image lies behind the fp32 image (and its effective-weight appendix): glass_dense_image_floats
    const int64_t K = comb ? 2 * HH : HH, base = 2 * (int64_t)HH * K;
    const float* Wcut = Wimg + ((comb && HH >= 256) ? base + base / 2 : base);
    if (comb) {
        constexpr bool kEff = HH >= 256;  // tiled_eff_fwd_shape: the image has the effective-weight appendix
        allow_tiled_lds(tiled_fwd_kernel<HH, true, BM, kEff, S3>, lds);
        hipLaunchKernelGGL((tiled_fwd_kernel<HH, true, BM, kEff, S3>), grid, dim3(kTThreads), lds, st, xa, lda, xb, ldb, Wimg,
                           bias, mask, zr, omz, act, T, ldt, out, ldo, N, stats, pro, (int)n_rt, Wcut);
    } else {
        allow_tiled_lds(tiled_fwd_kernel<HH, false, BM, false, S3>, lds);
        hipLaunchKernelGGL((tiled_fwd_kernel<HH, false, BM, false, S3>), grid, dim3(kTThreads), lds, st, xa, lda, xb, ldb, Wimg,
                           bias, mask, zr, omz, act, T, ldt, out, ldo, N, stats, pro, (int)n_rt, Wcut);
    }
}

int launch_tiled_fwd(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg, const float* bias,
                     const uint8_t* mask, float zr, float omz, int act, float* T, int64_t ldt, float* out, int64_t ldo,
                     int64_t N, int64_t H, double* stats, const GnPrologue& pro, hipStream_t st) {
#define GLASS_TFWD(HH, BM)                                                                                           \
    if (H == HH) {                                                                                                   \
        if (tiled_split_products())                                                                                  \
            tiled_fwd_launch<HH, BM, true>(xa, lda, xb, ldb, Wimg, bias, mask, zr, omz, act, T, ldt, out, ldo, N, stats, pro, st); \
        else                                                                                                         \
            tiled_fwd_launch<HH, BM, false>(xa, lda, xb, ldb, Wimg, bias, mask, zr, omz, act, T, ldt, out, ldo, N, stats, pro, st); \
    }
    if (H == 128 && tall128(N)) {
        tiled_fwd_launch<128, 128, true>(xa, lda, xb, ldb, Wimg, bias, mask, zr, omz, act, T, ldt, out, ldo, N, stats, pro, st);
        return launch_status("glass_dual_linear_fwd_f32 (tiled)");
    }
    GLASS_TFWD(128, 64) GLASS_TFWD(256, 128) GLASS_TFWD(512, 128)
#undef GLASS_TFWD
    return launch_status("glass_dual_linear_fwd_f32 (tiled)");
}

template <int HH, int NOUT, int BM, int BN, bool SPLIT, bool S3>
static void tiled_dgrad_launch(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, float zr,
                               float omz, int act, const float* WTimg, const float* addend, int64_t ldadd, const Drop& drop,
                               const uint64_t* rng_state, float* out, int64_t ldo, int64_t N, const GnBwdStats& gs,
                               hipStream_t st) {
    const int64_t n_rt = ceil_div(N, BM);
    constexpr bool kEff = !SPLIT && NOUT == 2 * HH;  // tiled_eff_dgrad_shape
    constexpr int NCT = SPLIT ? 1 : NOUT / BN;
    const size_t lds = StageGeom<BM, BN, S3, SPLIT ? 2 : 1>::kLds + (kEff ? 1024 : 0);  // + the labeled-row bookkeeping
    const int64_t base = (int64_t)NOUT * 2 * HH;  // fp32 image [NT = NOUT][KT = 2H]; then its appendix; then the cut image
    const float* Wcut = WTimg + (kEff ? base + base / 2 : base);
    allow_tiled_lds(tiled_dgrad_kernel<HH, NOUT, BM, BN, SPLIT, kEff, S3>, lds);
    hipLaunchKernelGGL((tiled_dgrad_kernel<HH, NOUT, BM, BN, SPLIT, kEff, S3>), dim3(tiled_grid(n_rt, NCT)), dim3(kTThreads), lds,
                       st, dsrc, ldd, T, ldt, mask, zr, omz, act, WTimg, addend, ldadd, drop, rng_state, out, ldo, N, gs,
                       (int)n_rt, Wcut);
}

int launch_tiled_dgrad(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, float zr,
                       float omz, int act, const float* WTimg, int64_t n_out, const float* addend, int64_t ldadd,
                       const Drop& drop, const uint64_t* rng_state, float* out, int64_t ldo, int64_t N, int64_t H,
                       const GnBwdStats& gs, hipStream_t st) {
#define GLASS_TDG1(HH, NOUT, BM, BN, SPLIT)                                                                           \
    {                                                                                                                \
        if (tiled_split_products())                                                                                  \
            tiled_dgrad_launch<HH, NOUT, BM, BN, SPLIT, true>(dsrc, ldd, T, ldt, mask, zr, omz, act, WTimg, addend, ldadd, drop, \
                                                              rng_state, out, ldo, N, gs, st);                       \
        else                                                                                                         \
            tiled_dgrad_launch<HH, NOUT, BM, BN, SPLIT, false>(dsrc, ldd, T, ldt, mask, zr, omz, act, WTimg, addend, ldadd, drop, \
                                                               rng_state, out, ldo, N, gs, st);                      \
    }
    // (hidden 128's data gradients left this family in round 6: trans_dgrad2 / trans_dgrad3 / comb_dgrad3 kernels of dense.hip,
    //  whose operand images are in other layouts — dgrad_launch never comes here with H = 128)
    if (H == 256) {
        if (n_out == H) GLASS_TDG1(256, 256, 128, 256, false) else GLASS_TDG1(256, 512, 128, 256, false)
    } else if (H == 512) {
        if (n_out == H) GLASS_TDG1(512, 512, 128, 256, false) else GLASS_TDG1(512, 1024, 128, 256, false)
    }
#undef GLASS_TDG1
    return launch_status("glass_dual_linear_dgrad_f32 (tiled)");
}

}  // namespace glass
