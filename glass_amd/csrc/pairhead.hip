// K9: the link-prediction head of the pre-training path on node pairs, forward and backward, hidden 64.
// reference: EdgeGNN.Pool + the 2-layer MLP head + BCE (impl/models.py:497-509; GNNEmb.py:94-99, 108-163: 131 072 edge /
// non-edge pairs per step):
//     pooled[p] = (emb[i0] + emb[i1]) / 2                          emb[subG_node]; torch.mean(emb, dim=1)
//     hid[p]    = relu(dropout(pooled[p] W0^T + b0))               MLP: Linear, Dropout, activation  (impl/models.py:33-50)
//     logit[p]  = hid[p] . w1 + b1                                 Linear(hidden, 1)
//     loss      = mean_p  max(x, 0) - x y + log1p(exp(-|x|))       BCEWithLogitsLoss (GNNEmb.py:129-130)
// Round 3 ran this as pair gather + library GEMMs + ATen dropout / ReLU / loss kernels (~30 launches, 0.45 ms).  Here:
//   forward  ONE launch: the pair rows are gathered and averaged while the 16-row stages are loaded (pooled is never
//            written), W0 on the fp32 matrix cores (a wave owns 16 of the 64 hidden columns, its weight slice in 16
//            registers), dropout + ReLU in the epilogue, the 64-wide dot with w1 by lane shuffles + LDS, loss term and
//            d loss / d logit per pair; hid [P, 64] is the only large output (the backward needs it: its sign pattern IS the
//            ReLU / dropout mask, so no mask is stored or re-drawn).
//   backward weight gradients as per-slab partial tiles (dW0 = dhid^T pooled over P rows: pooled is gathered again, dhid
//            = dlogit w1 . [hid > 0] / (1 - p) is synthesised) summed in slab order by a small reduce launch — deterministic;
//            the embedding gradient demb[n] = (sum over the pairs naming n of dhid / 2) W0 through the node-bucketed entry
//            lists of bucket.h: exact fixed-point sums (no float atomic, bitwise repeatable), then the 64 x 64 product per
//            node inside the same kernel.
#include "common.h"
#include "bucket.h"

namespace glass {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPH = 64;              // hidden width (and input width) of the head
constexpr int kPRS = kPH + 4;        // LDS row stride of a 16-row stage (floats)
constexpr int kPTile = 64;           // pairs per workgroup (forward)

struct PairHeadFwd {
    const float* emb; int64_t lde; int64_t n_nodes;
    const int64_t* pairs; int64_t P;
    const float *W0, *b0, *w1, *b1;
    const float* target;      // [P] 0 / 1 (float); nullptr: no loss (evaluation)
    Drop drop;                // p = 0: off
    const uint64_t* rng_state;
    float* hid;               // [P, 64] (nullptr: not kept — evaluation)
    float* logits;            // [P]
    float* dlogit;            // [P]  (sigmoid(x) - y) * gscale / P   (nullptr without target)
    double* loss_part;        // [gridDim.x] sum of the workgroup's loss terms
    const float* gscale;      // device scalar seed of the backward (nullptr: 1)
};

__global__ __launch_bounds__(kBlock) void pair_head_fwd_kernel(PairHeadFwd a) {
    __shared__ __attribute__((aligned(16))) float tile[2][16 * kPRS];   // pooled rows of a stage
    __shared__ __attribute__((aligned(16))) float mtile[2][16 * kPRS];  // their dropout keep-scales (for the epilogue)
    __shared__ float part[4][kPTile];                                  // per wave: partial logit of every row
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid >> 4, ga = tid & 15;
    const int64_t p0 = (int64_t)blockIdx.x * kPTile;
    const buf_rsrc r_emb = make_rsrc(a.emb, a.n_nodes * a.lde * 4);
    const buf_rsrc r_hid = make_rsrc(a.hid ? a.hid : a.emb, a.hid ? a.P * kPH * 4 : 0);
    // this wave's slice of W0 ([64 out][64 in], row-major): output column 16w + j, K-chunk q (16 consecutive inputs)
    float4 bw[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bw[v] = *reinterpret_cast<const float4*>(a.W0 + (16 * w + j) * kPH + 16 * q + 4 * v);
    const float b0c = a.b0[16 * w + j], w1c = a.w1[16 * w + j];
    Drop drop = a.drop;
    if (drop.p > 0.f) {
        drop.seed = a.rng_state[0];
        drop.step = a.rng_state[1];
    }
    // the four stages' pair indices of this thread's row (16 threads share a row: L1 hits)
    int i0[4], i1[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int64_t p = p0 + 16 * st + rs;
        int64_t x = -1, y = -1;
        if (p < a.P) {
            x = a.pairs[2 * p];
            y = a.pairs[2 * p + 1];
        }
        i0[st] = (x >= 0 && x < a.n_nodes) ? (int)x : -1;
        i1[st] = (y >= 0 && y < a.n_nodes) ? (int)y : -1;
    }
    auto issue = [&](int st, float4 (&raw)[2]) __attribute__((always_inline)) {
        raw[0] = buf_load4(r_emb, i0[st] >= 0 ? (int)((i0[st] * a.lde + 4 * ga) * 4) : kBufOOB);
        raw[1] = buf_load4(r_emb, i1[st] >= 0 ? (int)((i1[st] * a.lde + 4 * ga) * 4) : kBufOOB);
    };
    auto commit = [&](int st, const float4 (&raw)[2]) __attribute__((always_inline)) {
        // mean over the pair as torch.mean(emb[pair], dim=1): (x + y) * 0.5 (a missing endpoint counts as a zero row)
        const float4 v = make_float4((raw[0].x + raw[1].x) * 0.5f, (raw[0].y + raw[1].y) * 0.5f, (raw[0].z + raw[1].z) * 0.5f,
                                     (raw[0].w + raw[1].w) * 0.5f);
        *reinterpret_cast<float4*>(tile[st & 1] + rs * kPRS + 4 * ga) = v;
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop.p > 0.f) drop_scales<4>(drop, p0 + 16 * st + rs, 4 * ga, ds);
        *reinterpret_cast<float4*>(mtile[st & 1] + rs * kPRS + 4 * ga) = make_float4(ds[0], ds[1], ds[2], ds[3]);
    };
    float4 rawA[2], rawB[2];
    issue(0, rawA);
    issue(1, rawB);
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const float* A = tile[st & 1] + j * kPRS + 16 * q;
        float4 a4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) a4[v] = *reinterpret_cast<const float4*>(A + 4 * v);
        float dsr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dsr[r] = mtile[st & 1][(4 * q + r) * kPRS + 16 * w + j];
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
            const float x0[4] = {a4[v].x, a4[v].y, a4[v].z, a4[v].w}, y0[4] = {bw[v].x, bw[v].y, bw[v].z, bw[v].w};
            const float x1[4] = {a4[v + 1].x, a4[v + 1].y, a4[v + 1].z, a4[v + 1].w};
            const float y1[4] = {bw[v + 1].x, bw[v + 1].y, bw[v + 1].z, bw[v + 1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], y0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], y1[e], acc1, 0, 0, 0);
            }
        }
        if (st + 1 < 4) {
            if (st & 1) {
                commit(st + 1, rawA);
                if (st + 3 < 4) issue(st + 3, rawA);
            } else {
                commit(st + 1, rawB);
                if (st + 3 < 4) issue(st + 3, rawB);
            }
        }
        // acc[r] = row 16 st + 4q + r, hidden column 16w + j:  Linear -> Dropout -> ReLU, then this column's term of the logit
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t p = p0 + 16 * st + 4 * q + r;
            const float h = fmaxf((acc0[r] + acc1[r] + b0c) * dsr[r], 0.f);
            buf_store1(r_hid, p < a.P ? (int)((p * kPH + 16 * w + j) * 4) : kBufOOB, h);
            t[r] = h * w1c;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // sum over the wave's 16 columns (lanes j)
            t[r] += __shfl_xor(t[r], 1);
            t[r] += __shfl_xor(t[r], 2);
            t[r] += __shfl_xor(t[r], 4);
            t[r] += __shfl_xor(t[r], 8);
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[w][16 * st + 4 * q + r] = t[r];
        }
        if (st + 1 < 4) lds_barrier();
    }
    __syncthreads();
    double term = 0.0;
    if (tid < kPTile) {
        const int64_t p = p0 + tid;
        if (p < a.P) {
            // the four waves' column blocks in fixed order: deterministic
            const float x = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + a.b1[0];
            a.logits[p] = x;
            if (a.target) {
                const float y = a.target[p];
                const float gs = a.gscale ? a.gscale[0] : 1.f;
                a.dlogit[p] = gs * (1.f / (1.f + expf(-x)) - y) / (float)a.P;
                term = (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));  // torch's stable BCE-with-logits
            }
        }
    }
    if (w == 0 && a.loss_part) {  // (kPTile == one wave)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) term += __shfl_xor(term, d);
        if (lane == 0) a.loss_part[blockIdx.x] = term;
    }
}

// ---- backward, weight side -------------------------------------------------------------------------------------------
// Per slab of rows: dW0 partial [64 out][64 in] = sum_p dhid[p]^T pooled[p], db0 / dw1 partials [64], db1 partial.
// The node index is the MFMA k index: the stage tiles are plain row-major [16 rows][64] and an operand is ONE ds_read_b32
// per MFMA (consecutive lanes, consecutive words: conflict-free) — no transposed tiles.  A wave owns 16 outputs x all 64
// inputs (4 accumulator tiles).
constexpr int kPSlabRows = 256;   // 16 stages

struct PairHeadBwd {
    const float* emb; int64_t lde; int64_t n_nodes;
    const int64_t* pairs; int64_t P;
    const float* hid;      // [P, 64]
    const float* dlogit;   // [P]
    const float* w1;       // [64]
    float inv_keep;        // 1 / (1 - p): hid > 0 implies the element was kept
    float* part;           // [n_slabs][64*64 + 64 + 64 + 4]
};
constexpr int kPPart = kPH * kPH + 2 * kPH + 4;

__global__ __launch_bounds__(kBlock) void pair_head_wgrad_kernel(PairHeadBwd a) {
    __shared__ __attribute__((aligned(16))) float dt[2][16 * kPRS];  // dhid rows of a stage
    __shared__ __attribute__((aligned(16))) float pt[2][16 * kPRS];  // pooled rows
    __shared__ float red[3][16][kPH + 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rs = tid >> 4, ga = tid & 15;
    const int64_t r0 = (int64_t)blockIdx.x * kPSlabRows;
    const int64_t r1 = r0 + kPSlabRows < a.P ? r0 + kPSlabRows : a.P;
    const int n_st = (int)((r1 - r0 + 15) / 16);
    const buf_rsrc r_emb = make_rsrc(a.emb, a.n_nodes * a.lde * 4), r_hid = make_rsrc(a.hid, a.P * kPH * 4);
    const float4 w14 = *reinterpret_cast<const float4*>(a.w1 + 4 * ga);
    struct Raw {
        float4 h, e0, e1;
        float dl;
    };
    auto issue = [&](int st, Raw& R) __attribute__((always_inline)) {
        const int64_t p = r0 + 16 * st + rs;
        const bool ok = st < n_st && p < r1;
        int64_t x = -1, y = -1;
        R.dl = 0.f;
        if (ok) {
            x = a.pairs[2 * p];
            y = a.pairs[2 * p + 1];
            R.dl = a.dlogit[p];
        }
        const bool vx = x >= 0 && x < a.n_nodes, vy = y >= 0 && y < a.n_nodes;
        R.h = buf_load4(r_hid, ok ? (int)((p * kPH + 4 * ga) * 4) : kBufOOB);
        R.e0 = buf_load4(r_emb, vx ? (int)((x * a.lde + 4 * ga) * 4) : kBufOOB);
        R.e1 = buf_load4(r_emb, vy ? (int)((y * a.lde + 4 * ga) * 4) : kBufOOB);
    };
    float dsum[4] = {0.f, 0.f, 0.f, 0.f}, wsum[4] = {0.f, 0.f, 0.f, 0.f}, lsum = 0.f;
    auto commit = [&](int st, const Raw& R) __attribute__((always_inline)) {
        const float hv[4] = {R.h.x, R.h.y, R.h.z, R.h.w}, wv[4] = {w14.x, w14.y, w14.z, w14.w};
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = hv[k] > 0.f ? R.dl * wv[k] * a.inv_keep : 0.f;   // d loss / d (Linear output): through ReLU and dropout
            dsum[k] += d[k];
            wsum[k] = fmaf(R.dl, hv[k], wsum[k]);
        }
        if (ga == 0) lsum += R.dl;
        *reinterpret_cast<float4*>(dt[st & 1] + rs * kPRS + 4 * ga) = make_float4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<float4*>(pt[st & 1] + rs * kPRS + 4 * ga) =
            make_float4((R.e0.x + R.e1.x) * 0.5f, (R.e0.y + R.e1.y) * 0.5f, (R.e0.z + R.e1.z) * 0.5f, (R.e0.w + R.e1.w) * 0.5f);
    };
    Raw rawA, rawB;
    issue(0, rawA);
    issue(1, rawB);
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    commit(0, rawA);
    issue(2, rawA);
    lds_barrier();
    for (int st = 0; st < n_st; ++st) {
        const float* D = dt[st & 1];
        const float* Pm = pt[st & 1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // MFMA k index = row 4s + q of the stage
            const float av = D[(4 * s + q) * kPRS + 16 * w + j];  // A[o = 16w + j][k]
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Pm[(4 * s + q) * kPRS + 16 * b + j], acc[b], 0, 0, 0);
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
    // acc[b][r] = dW0[o = 16w + 4q + r][i = 16b + j]
    float* pw = a.part + (int64_t)blockIdx.x * kPPart;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(16 * w + 4 * q + r) * kPH + 16 * b + j] = acc[b][r];
    // column sums over the 16 row slots of the loader threads, in slot order
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][rs][4 * ga + k] = dsum[k];
        red[1][rs][4 * ga + k] = wsum[k];
    }
    if (ga == 0) red[2][rs][0] = lsum;
    __syncthreads();
    if (tid < kPH) {
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < 16; ++r) {
            s0 += red[0][r][tid];
            s1 += red[1][r][tid];
        }
        pw[kPH * kPH + tid] = s0;
        pw[kPH * kPH + kPH + tid] = s1;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += red[2][r][0];
        pw[kPH * kPH + 2 * kPH] = s;
    }
}

// partials -> gradients + the mean loss.  16 output elements x 16 slab groups per workgroup: a thread adds the slabs
// b = g, g + 16, ... (eight loads in flight; a serial walk over 512 slabs cost 123 us: one dependent L2 round trip per slab),
// the 16 groups are combined through LDS in group order — a fixed summation tree: deterministic.
constexpr int kPRedElems = 16;

__global__ __launch_bounds__(kBlock) void pair_head_reduce_kernel(const float* __restrict__ part, int n_slabs,
                                                                  const double* __restrict__ loss_part, int n_loss, int64_t P,
                                                                  float* __restrict__ dW0, float* __restrict__ db0,
                                                                  float* __restrict__ dw1, float* __restrict__ db1,
                                                                  float* __restrict__ loss, int accumulate) {
    __shared__ float sm[16][kPRedElems];
    constexpr int kOut = kPH * kPH + 2 * kPH + 1;
    const int el = threadIdx.x & (kPRedElems - 1), g = threadIdx.x >> 4;
    const int e = blockIdx.x * kPRedElems + el;
    float acc = 0.f;
    if (e < kOut) {
        for (int b0 = g; b0 < n_slabs; b0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 16 * u;
                v[u] = b < n_slabs ? part[(int64_t)b * kPPart + e] : 0.f;
            }
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    sm[g][el] = acc;
    __syncthreads();
    if (g == 0 && e < kOut) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sm[k][el];
        float* dst = e < kPH * kPH ? dW0 + e : e < kPH * kPH + kPH ? db0 + (e - kPH * kPH)
                     : e < kPH * kPH + 2 * kPH ? dw1 + (e - kPH * kPH - kPH) : db1;
        *dst = (accumulate ? *dst : 0.f) + s;
    }
    if (blockIdx.x == gridDim.x - 1 && loss && loss_part) {
        __shared__ double sl[kBlock];
        double s = 0.0;
        for (int k = threadIdx.x; k < n_loss; k += kBlock) s += loss_part[k];
        sl[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < kBlock; ++k) t += sl[k];
            loss[0] = (float)(t / (double)P);
        }
    }
}

// ---- backward, embedding side ------------------------------------------------------------------------------------------
// demb[n] = ( sum over the entries (pair p, side) naming n of dhid[p] / 2 ) W0
//         = 0.5 / (1 - p) * ( w1 . sum_e dlogit[p_e] [hid[p_e] > 0] ) W0          (dhid synthesised: nothing but hid is read)
// The sum runs over node n's entry list (bucket.h: filled through integer cursors, arbitrary order) in exact fixed point,
// so the result is the same bits every run; then the 64 x 64 product per node from an LDS copy of W0.  16 lanes x float4
// cover a row, 16 nodes per workgroup; lists of >= 64 entries are walked by all 16 lane groups (hub nodes).
constexpr int kPLong = 64;

__global__ __launch_bounds__(kBlock) void pair_head_demb_kernel(const float* __restrict__ hid, const float* __restrict__ dlogit,
                                                                const float* __restrict__ w1, const float* __restrict__ W0,
                                                                float half_inv_keep, const int32_t* __restrict__ off,
                                                                const int32_t* __restrict__ list, float* __restrict__ demb,
                                                                int64_t lde, int64_t n_nodes) {
    __shared__ __attribute__((aligned(16))) float w0s[kPH * kPH];
    __shared__ __attribute__((aligned(16))) float srow[16][kPH];
    __shared__ long long red[kBlock * 4 * 2];
    const int tid = threadIdx.x, li = tid & 15, slot = tid >> 4;
    for (int k = tid; k < kPH * kPH / 4; k += kBlock) reinterpret_cast<float4*>(w0s)[k] = reinterpret_cast<const float4*>(W0)[k];
    const int64_t node0 = (int64_t)blockIdx.x * 16;
    const int64_t node = node0 + slot;
    const float4 w14 = *reinterpret_cast<const float4*>(w1 + 4 * li);
    const float wv[4] = {w14.x, w14.y, w14.z, w14.w};
    auto term = [&](int e, float (&t)[4]) __attribute__((always_inline)) {
        const int64_t p = e >> 1;  // (bit 0 of a pair-form entry: both endpoints valid — the mean divides by 2 either way)
        const float4 h = *reinterpret_cast<const float4*>(hid + p * kPH + 4 * li);
        const float dl = dlogit[p];
        t[0] = h.x > 0.f ? dl : 0.f;
        t[1] = h.y > 0.f ? dl : 0.f;
        t[2] = h.z > 0.f ? dl : 0.f;
        t[3] = h.w > 0.f ? dl : 0.f;
    };
    ExactSum s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k].hi = s[k].lo = 0;
    const int beg = node < n_nodes ? off[node] : 0, end = node < n_nodes ? off[node + 1] : 0;
    const bool is_long = end - beg >= kPLong;
    if (!is_long) {
        int i = beg;
        for (; i + 1 < end; i += 2) {  // two independent chains of loads in flight
            float t0[4], t1[4];
            term(list[i], t0);
            term(list[i + 1], t1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k].add(t0[k]);
                s[k].add(t1[k]);
            }
        }
        if (i < end) {
            float t0[4];
            term(list[i], t0);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k].add(t0[k]);
        }
    }
    // long lists: the whole workgroup takes one node at a time
    for (int t = 0; t < 16; ++t) {
        const int64_t nd = node0 + t;
        if (nd >= n_nodes) break;
        const int b2 = off[nd], e2 = off[nd + 1];
        if (e2 - b2 < kPLong) continue;  // workgroup-uniform
        ExactSum u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k].hi = u[k].lo = 0;
        for (int i = b2 + slot; i < e2; i += 16) {
            float t0[4];
            term(list[i], t0);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k].add(t0[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(tid * 4 + k) * 2] = u[k].hi;
            red[(tid * 4 + k) * 2 + 1] = u[k].lo;
        }
        __syncthreads();
        if (slot == t) {  // the node's own lane group collects the 16 slots
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ExactSum tot{0, 0};
                for (int r = 0; r < 16; ++r) {
                    tot.hi += red[((r * 16 + li) * 4 + k) * 2];
                    tot.lo += red[((r * 16 + li) * 4 + k) * 2 + 1];
                }
                s[k] = tot;
            }
        }
    }
    // S[node][4 li + k] = 0.5 / (1 - p) * w1 * sum;  then demb[node] = S[node] W0
#pragma unroll
    for (int k = 0; k < 4; ++k) srow[slot][4 * li + k] = s[k].value() * wv[k] * half_inv_keep;
    __syncthreads();
    if (node < n_nodes) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < kPH; ++c) {
            const float sv = srow[slot][c];
            const float4 wr = *reinterpret_cast<const float4*>(w0s + c * kPH + 4 * li);
            o[0] = fmaf(sv, wr.x, o[0]);
            o[1] = fmaf(sv, wr.y, o[1]);
            o[2] = fmaf(sv, wr.z, o[2]);
            o[3] = fmaf(sv, wr.w, o[3]);
        }
        *reinterpret_cast<float4*>(demb + node * lde + 4 * li) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_pair_head_supported(int64_t hidden) { return hidden == kPH ? 1 : 0; }

// floats of `ws` (weight-gradient partial tiles + loss partials as doubles behind them) and bytes of the bucket scratch
extern "C" int64_t glass_pair_head_ws_bytes(int64_t n_nodes, int64_t P) {
    if (n_nodes <= 0 || P <= 0) return GLASS_E_ARG;
    const int64_t n_slabs = ceil_div(P, (int64_t)kPSlabRows), n_blk = ceil_div(P, (int64_t)kPTile);
    int64_t bytes = n_slabs * kPPart * (int64_t)sizeof(float);
    bytes = (bytes + 15) / 16 * 16 + n_blk * (int64_t)sizeof(double);
    bytes = (bytes + 15) / 16 * 16 + (int64_t)sizeof(int32_t) * bucket_ws_words(n_nodes, P, 2, true);
    return bytes + 64;
}

extern "C" int glass_pair_head_fwd_f32(const float* emb, int64_t lde, int64_t n_nodes, const int64_t* pairs, int64_t P,
                                       const float* W0, const float* b0, const float* w1, const float* b1, const float* target,
                                       float p_drop, const uint64_t* rng_state, uint64_t call_id, const float* grad_scale,
                                       float* hid, float* logits, float* dlogit, void* ws, void* stream) {
    GLASS_REQUIRE(emb && pairs && W0 && b0 && w1 && b1 && logits && n_nodes > 0 && P > 0 && lde >= kPH, "pair_head_fwd: null pointer / bad sizes");
    GLASS_REQUIRE(lde % 4 == 0 && aligned16(emb) && aligned16(W0) && aligned16(w1) && (!hid || aligned16(hid)),
                  "pair_head_fwd: operands must be 16-B aligned with ld %% 4 == 0");
    GLASS_REQUIRE(!target || (dlogit && hid && ws && aligned16(ws)), "pair_head_fwd: a training pass needs hid, dlogit and the workspace");
    GLASS_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || rng_state), "pair_head_fwd: bad dropout arguments");
    GLASS_REQUIRE(n_nodes * lde * 4 < (1ll << 31) && P * kPH * 4 < (1ll << 31), "pair_head_fwd: rows * ld * 4 must stay below 2^31 (32-bit buffer offsets)");
    const int64_t n_blk = ceil_div(P, (int64_t)kPTile), n_slabs = ceil_div(P, (int64_t)kPSlabRows);
    double* loss_part = nullptr;
    if (target) {
        int64_t off = n_slabs * kPPart * (int64_t)sizeof(float);
        off = (off + 15) / 16 * 16;
        loss_part = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + off);
    }
    PairHeadFwd a{emb, lde, n_nodes, pairs, P, W0, b0, w1, b1, target, make_drop(p_drop, call_id, kPH), rng_state, hid, logits, dlogit,
                  loss_part, grad_scale};
    hipLaunchKernelGGL(pair_head_fwd_kernel, dim3((unsigned)n_blk), dim3(kBlock), 0, (hipStream_t)stream, a);
    return launch_status("glass_pair_head_fwd_f32");
}

extern "C" int glass_pair_head_bwd_f32(const float* emb, int64_t lde, int64_t n_nodes, const int64_t* pairs, int64_t P,
                                       const float* W0, const float* w1, const float* hid, const float* dlogit, float p_drop,
                                       float* dW0, float* db0, float* dw1, float* db1, int accumulate, float* loss, float* demb,
                                       int64_t ldde, void* ws, void* stream) {
    GLASS_REQUIRE(emb && pairs && W0 && w1 && hid && dlogit && dW0 && db0 && dw1 && db1 && demb && ws && n_nodes > 0 && P > 0,
                  "pair_head_bwd: null pointer");
    GLASS_REQUIRE(lde >= kPH && lde % 4 == 0 && ldde >= kPH && ldde % 4 == 0 && aligned16(emb) && aligned16(hid) && aligned16(demb) &&
                      aligned16(W0) && aligned16(w1) && aligned16(ws) && p_drop >= 0.f && p_drop < 1.f,
                  "pair_head_bwd: operands must be 16-B aligned with ld %% 4 == 0");
    GLASS_REQUIRE(n_nodes * lde * 4 < (1ll << 31) && P * kPH * 4 < (1ll << 31) && P < (1ll << 29), "pair_head_bwd: sizes beyond the 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_slabs = ceil_div(P, (int64_t)kPSlabRows), n_blk = ceil_div(P, (int64_t)kPTile);
    const float inv_keep = 1.f / (1.f - p_drop);
    float* part = reinterpret_cast<float*>(ws);
    int64_t off = n_slabs * kPPart * (int64_t)sizeof(float);
    off = (off + 15) / 16 * 16;
    const double* loss_part = reinterpret_cast<const double*>(reinterpret_cast<char*>(ws) + off);
    off = (off + n_blk * (int64_t)sizeof(double) + 15) / 16 * 16;
    void* bws = reinterpret_cast<char*>(ws) + off;
    PairHeadBwd a{emb, lde, n_nodes, pairs, P, hid, dlogit, w1, inv_keep, part};
    hipLaunchKernelGGL(pair_head_wgrad_kernel, dim3((unsigned)n_slabs), dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(pair_head_reduce_kernel, dim3((unsigned)ceil_div(kPH * kPH + 2 * kPH + 1, (int64_t)kPRedElems)), dim3(kBlock), 0, st,
                       part, (int)n_slabs, loss ? loss_part : nullptr, (int)n_blk, P, dW0, db0, dw1, db1, loss, accumulate);
    BucketLists bl;
    int rc = bucket_build(pairs, P, 2, GLASS_POOL_MEAN, true, n_nodes, bws, st, &bl);
    if (rc) return rc;
    hipLaunchKernelGGL(pair_head_demb_kernel, dim3((unsigned)ceil_div(n_nodes, (int64_t)16)), dim3(kBlock), 0, st, hid, dlogit, w1, W0,
                       0.5f * inv_keep, bl.off, bl.list, demb, ldde, n_nodes);
    return launch_status("glass_pair_head_bwd_f32");
}
