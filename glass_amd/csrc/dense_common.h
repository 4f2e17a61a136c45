// Pieces shared by the two families of fused dense kernels: dense.hip (16x16x4 MFMA, a wave owns 16 rows and all
// output columns: small graphs, hidden 64 / 128) and dense_tiled.hip (LDS-tiled 32x32x2 MFMA GEMM: hidden 256 / 512).
#pragma once
#include "common.h"
#include "gn_acc.h"

namespace glass {

// Optional GraphNorm prologue on the xa operand of the forward kernels: xa is the INPUT of a GraphNorm whose statistics
// are already final (saved[4C] = mean, rstd, scale, shift); the kernel normalises (+ ELU + dropout) its operand while
// staging it, uses it as the MFMA operand and writes it to `side` (the layer's backward and, for the trans pair, the
// comb pair of the same layer read it) — the GraphNorm apply launch and its read of xa disappear.
// Laboratory build only (-DGLASS_DENSE_TRACE; tools/dense_trace.py): per-wave wall-clock stamps (100 MHz) of the phases of
// one selected dense kernel, [workgroup][wave][8 slots].
#ifdef GLASS_DENSE_TRACE
extern __device__ unsigned long long* g_dense_trace;
extern __device__ int g_dense_trace_sel;
#define D_STAMP(sel, slot)                                                                                         \
    do {                                                                                                           \
        if (g_dense_trace && g_dense_trace_sel == (sel) && (threadIdx.x & 63) == 0 && threadIdx.x < 256)                              \
            g_dense_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64();            \
    } while (0)
#else
#define D_STAMP(sel, slot) do { } while (0)
#endif

struct GnPrologue {
    const float* saved;  // nullptr: no prologue
    int C, act;
    Drop drop;
    const uint64_t* rng_state;
    float* side;
    int64_t lds;
    GnExactSrc src;      // src.acc != nullptr: the statistics are still in exact accumulators (gn_acc.h), `saved` gets written
    const int64_t* gather = nullptr;  // tiled forward (trans pair, layer 0): operand row n is row gather[n] of xa (the embedding table)
    int64_t gather_rows = 0;          // rows of that table (indices are clamped into it)
};

// Optional epilogue of the data-gradient kernels: their first H output columns are the gradient dy of a GraphNorm
// OUTPUT (conv.gn's for the comb pair, gns[l]'s for the next layer's trans pair), so the two backward column sums of
// that GraphNorm — S1 = sum g, S2 = sum g*xhat with g = dy * dropmask * act'(x*scale + shift) — are accumulated here
// from the tile in registers plus one read of the GraphNorm input x, instead of by a statistics launch re-reading
// dy and x.  partial[row tile][2][H] doubles, consumed by glass_graphnorm_bwd_from_stats_f32.
struct GnBwdStats {
    double* partial;  // nullptr: off.  exact != 0: not per-workgroup partials but the exact accumulators of gn_acc.h (int64)
    int exact;
    const float* x; int64_t ldx;
    const float *saved, *alpha;
    int act;
    Drop drop;
};

// The gradient operand of the comb pair's backward NOT materialised: it is the input gradient of the GraphNorm between
// two layers (gns[l], impl/models.py:257-259 backward), derived while the staged kernels load their rows —
//   dc = A * g + Bx * x + K (+ addend),  g = dy * dropmask * act'(x * scale + shift)
// with the coefficients from that GraphNorm's two backward sums (exact accumulators, gn_acc.h): the backward-apply launch
// between the next layer's trans backward and this launch disappears.  acc == nullptr: off (dsrc is final).
struct GnBwdSrc {
    const long long* acc;
    int n_rep;
    const float* dy; int64_t lddy;
    const float* x; int64_t ldx;
    const float* addend; int64_t ldadd;
    const float *saved, *gamma, *alpha;
    float *dgamma, *dbeta, *dalpha;
    int accumulate, act;
    Drop drop;
};

// The unique labeled rows of the batch (glass_batch_labels): the comb pair in effective-weight form runs every row tile
// with the unlabeled-row weight and `n_main` row tiles first, then ceil(cap / 64) extra workgroups that recompute the listed
// rows with the labeled-row weight (the main tiles do not store those rows).
struct LabRows {
    const int32_t* rows;   // [cap] unique labeled node ids, first-occurrence order
    const int32_t* count;  // device word: how many of them
    int n_main;            // row-tile workgroups in front of the extra ones
    int cap;               // capacity of `rows` (entries): bounds list reads issued before `count` is known
};

// Operand-image layouts written by glass_dense_pack_batch_f32 (bits 1.. of its per-job flags; bit 0 = transposed source)
// (glass_dual_linear_layout(H) == 2: no images at all — the narrow kernels read the row-major weight)
enum { kLayoutWave16 = 0, kLayoutTiledPaired = 1, kLayoutTiledPlain = 2, kLayoutTiledSplit = 3, kLayoutTiledPlainEff = 4,
       kLayoutTiledPairedEff = 5, kLayoutWave16EffFwd = 6, kLayoutWave16EffDgrad = 7 };
constexpr int kLayoutWave16EffDgradCols = 10;  // as 7 with that column order (comb_bwd_eff2_kernel)
constexpr int kLayoutWave16Cols = 9;  // as 0 with tile t = columns 64 (t >> 2) + 16 (t & 3) .. + 15 (trans_fwd2_kernel)
constexpr int kLayoutWave16EffFwdCols = 8;  // as 6 with tile t = output columns 16t .. 16t+15 (comb_fwd_eff2_kernel)
// kLayoutWave16EffFwd / EffDgrad (comb pair at hidden 64, dense.hip): TWO wave16 images back to back, of the effective weight
// of unlabeled rows (1-z)*W1 + z*W0 and of labeled rows z*W1 + (1-z)*W0 — forward: [H outputs][2H inputs]; data gradient
// (transposed source): [2H outputs of the product = inputs of the pair][H].  NT*KT floats, as the plain image.
// kLayoutTiledPairedEff: the paired forward image of a comb pair followed by W_unl = (1-z)*W1 + z*W0 ([H][KT]) in the plain
// tiling — row tiles without a labeled row produce 256 output columns per column tile from it
// kLayoutTiledPlainEff: the plain data-gradient image of a comb pair followed by the image of its UNLABELED-row effective
// weight (1-z)*W1 + z*W0 over K = KT / 2 — row tiles without a labeled row multiply that one instead (half the K loop)
bool tiled_eff_dgrad_shape(int64_t H, int64_t n_out);  // dense_tiled.hip: shapes whose data gradient reads a PlainEff image
bool tiled_eff_fwd_shape(int64_t H, int64_t K);        // ... whose forward reads a PairedEff image

// dense_tiled.hip
bool tiled_shape_ok(int64_t H);
// Options of the CURRENT entry-point call (the bits of its `act` word above GLASS_ACT_MASK), visible to the launch helpers
// of the other translation units for the duration of that call, on the calling thread only: set by a CallOptions object at
// the top of the four dense entries, restored when it goes out of scope.  Not state: nothing outlives the call.
extern thread_local int t_call_options;
struct CallOptions {
    int prev;
    explicit CallOptions(int& act) : prev(t_call_options) {
        t_call_options = act & ~GLASS_ACT_MASK;
        act &= GLASS_ACT_MASK;
    }
    ~CallOptions() { t_call_options = prev; }
    CallOptions(const CallOptions&) = delete;
    CallOptions& operator=(const CallOptions&) = delete;
};
// fp32 products as six bf16 partial products (split_mma.h; the tiled family's default) or the f32-input MFMA: per call
inline bool tiled_split_products() { return !(t_call_options & GLASS_DENSE_F32_PRODUCTS); }

// Laboratory knobs (variants kept for A/B measurements): CONSTANTS in the product build — the library reads no environment
// variable.  A lab build (tools/build_trace.sh: -DGLASS_LAB=1, linked with tools/lab/lab_knobs.cpp) resolves them at run time.
#ifndef GLASS_LAB
#define GLASS_LAB 0
#endif
#if GLASS_LAB
int lab_knob(const char* name, int dflt);
#else
inline int lab_knob(const char*, int dflt) { return dflt; }
#endif
// Timing-only laboratory switch (results WRONG): keep 3 of every 8 MFMAs of the hidden-64 backward bodies — the matrix-core
// time their split product form would leave (6/16) — to bound what that form can return before building it.  Never in the product.
#if GLASS_LAB && defined(GLASS_LAB_MFMA38)
#define GLASS_MFMA_KEEP(i) (((i) & 7) < 3)
#else
#define GLASS_MFMA_KEEP(i) true
#endif
int tiled_rows(int64_t H);  // rows per workgroup = rows per statistics partial of the tiled kernels (64 at hidden 128, else 128)
int launch_tiled_fwd(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg, const float* bias,
                     const uint8_t* mask, float zr, float omz, int act, float* T, int64_t ldt, float* out, int64_t ldo,
                     int64_t N, int64_t H, double* stats, const GnPrologue& pro, hipStream_t st);
int launch_tiled_dgrad(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, float zr,
                       float omz, int act, const float* WTimg, int64_t n_out, const float* addend, int64_t ldadd,
                       const Drop& drop, const uint64_t* rng_state, float* out, int64_t ldo, int64_t N, int64_t H,
                       const GnBwdStats& gs, hipStream_t st);
// dense_narrow.hip: hidden <= 32 (the shipped YAMLs' widths: 8, 17, 20) — a thread owns a row, weights read as they are
struct WgradSynth;
bool narrow_shape_ok(int64_t H);
int narrow_rows();
void narrow_wgrad_geom(int64_t N, int64_t O, int64_t I, int* n_slabs, int* stride);
int launch_narrow_fwd(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* W, const float* bias,
                      const uint8_t* mask, float zr, float omz, int act, float* T, int64_t ldt, float* out, int64_t ldo,
                      int64_t N, int64_t H, double* stats, const GnPrologue& pro, const int64_t* xa_index, int64_t xa_rows,
                      hipStream_t st);
int launch_narrow_dgrad(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, float zr, float omz,
                        int act, const float* W, int64_t n_out, const float* addend, int64_t ldadd, const Drop& drop,
                        const uint64_t* rng_state, float* out, int64_t ldo, int64_t N, int64_t H, const GnBwdStats& gs,
                        hipStream_t st);
int launch_narrow_wgrad(const WgradSynth& sy, const float* X, int64_t ldx, int64_t N, int64_t O, int64_t I, float* part,
                        hipStream_t st);
// image element source of the tiled layouts (used by the pack kernel): output column of image slot nl of column tile ct
__host__ __device__ inline int tiled_col(int layout, int ct, int nl, int H) {
    const int wn = nl >> 7, cb = (nl >> 5) & 3, j = nl & 31;
    if (layout == kLayoutTiledPaired) {
        // a wave's 128 slots = 64 columns of the f1 half (cb 0,1) + the SAME 64 columns of the f0 half (cb 2,3), so the
        // label mix of the forward epilogue is register-local; slot (cb, j) <-> column 2j + (cb & 1): a lane holds two
        // consecutive columns -> 8-byte stores, 256 contiguous bytes per row and half-wave
        const int c = ct * 128 + wn * 64 + 2 * j + (cb & 1);
        return (cb >> 1) ? H + c : c;
    }
    // plain: slot (cb, j) <-> column 4j + cb: a lane holds four consecutive columns -> 16-byte stores
    return ct * 256 + wn * 128 + 4 * j + cb;
}

}  // namespace glass
