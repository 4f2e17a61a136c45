// Training-step readout: final GraphNorm (apply) -> subgraph pooling -> Linear head -> loss, and the whole
// backward down to the gradient of the jumping-knowledge buffer, in four launches.
// reference: impl/models.py:266/271 (gns[-1]), 346-350 (Pool), GLASSTest.py:159-160 (head), :57-58/:69 (loss).
//
// Unfused this is 12 dependent launches (GraphNorm apply; pool; logits; loss mean; head backward; zero-fill of
// the [N,C] embedding gradient; pool scatter; GraphNorm backward statistics / finalize / apply), five of them
// full passes over [N,C] although only the P = sum of subgraph sizes pooled rows carry a gradient.  Here:
//   R1  one workgroup per subgraph: gathers its raw rows, applies the GraphNorm affine on the fly, pools, computes
//       its logits, its loss term, d logits, d pooled — and, because the gradient of the GraphNorm OUTPUT is nonzero
//       only on pooled rows, the subgraph's share of the two backward column sums
//       S1 = sum_n g[n], S2 = sum_n g[n]*xhat[n]   (g[n] = sum over the subgraphs holding n of scale_b * dpooled[b]).
//   R2  head weight/bias gradient rows, the mean loss, and the GraphNorm backward finalize (independent roles of one
//       launch).
//   R3  dense part of the GraphNorm backward over all nodes: d jk = Bx*x + K.
//   R4  sparse part: d jk[n] += A * g[n] on the pooled rows: an ordered, atomic-free scatter (bitwise repeatable) while
//       pos fits LDS (B*Smax <= 16 384); beyond that float atomics (order-dependent in the last bits when subgraphs
//       share nodes, because the row already holds the dense part).
#include "bucket.h"
#include "common.h"
#include "gn_math.h"
#include "gn_acc.h"

namespace glass {

constexpr int kReadoutMaxK = 256;
constexpr int64_t kReadoutOrderedMax = 16384;  // pos entries staged in LDS by the atomic-free scatter (64 KiB)
constexpr int kLossCE = 0, kLossBCE = 1;

struct ReadoutWs {
    double* partial;   // [B][2][C]
    float* coef;       // [3C]  A, Bx, K
    float* dys;        // [B][C]  gradient of every pooled row of subgraph b (already scaled by the pool scale)
    float* dlogits;    // [B][K]
    float* loss_rows;  // [B]
};

static ReadoutWs carve_ws(void* ws, int64_t B, int64_t C, int64_t K) {
    ReadoutWs w;
    char* p = (char*)ws;
    w.partial = (double*)p;
    p += sizeof(double) * 2 * B * C;
    w.coef = (float*)p;
    p += sizeof(float) * 4 * C;
    w.dys = (float*)p;
    p += sizeof(float) * B * C;
    w.dlogits = (float*)p;
    p += sizeof(float) * B * K;
    w.loss_rows = (float*)p;
    return w;
}

__device__ __forceinline__ int valid_count(const int64_t* __restrict__ prow, int Smax, int64_t n_nodes) {
    int cnt = 0;
    for (int j0 = 0; j0 < Smax; j0 += kBlock) {
        const int j = j0 + threadIdx.x;
        cnt += __syncthreads_count(j < Smax && prow[j] >= 0 && prow[j] < n_nodes);
    }
    return cnt;
}

__device__ __forceinline__ float readout_pool_scale(int mode, int cnt) {
    if (mode == GLASS_POOL_MEAN) return 1.0f / (float)(cnt > 0 ? cnt : 1);
    if (mode == GLASS_POOL_SIZE) return cnt > 0 ? 1.0f / sqrtf((float)cnt) : 0.f;
    return 1.0f;
}

struct R1Args {
    const float* jk; int64_t ldj;
    const float *saved, *alpha;
    const int64_t* pos; int Smax, mode;
    const float *Wh, *bh; const void* target; int loss_mode, B, C, K;
    const float* gl;
    float *pooled, *logits;
    ReadoutWs ws;
    int64_t n_nodes;
    int tc_log2;
    GnExactSrc src;  // src.acc: the final GraphNorm's forward sums are still in exact accumulators (gn_acc.h)
    int stash;           // != 0: the padded row's node ids are staged in LDS (Smax ints behind the other arrays) by the counting pass
    long long* bwd_acc;  // != nullptr: the subgraph's share of the two backward column sums goes to exact accumulators
    int bwd_rep;         // (the backfill launch folds them: no reduce launch in between)
};

// dynamic LDS (floats): sums[2C doubles] | coef_s[2C] | mu_rstd_s[2C] | pooled_s[C] | xh_s[C] | red[kBlock*8] |
// zs[kReadoutMaxK] | dl[kReadoutMaxK]
// VW = floats per lane and access: 4 (C % 4 == 0, 16-B aligned rows) or 1 (any C: the 17-wide layers of config/component.yml)
template <int VW>
__global__ __launch_bounds__(kBlock) void readout_subgraph_kernel(R1Args a) {
    extern __shared__ float sm[];
    const int C = a.C, K = a.K;
    double* sums = reinterpret_cast<double*>(sm);
    float* coef_s = sm + 4 * C;
    float* mu_rstd_s = sm + 6 * C;
    float* pooled_s = sm + 8 * C;
    float* xh_s = sm + 9 * C;
    float* red = sm + 10 * C;
    float* zs = red + kBlock * 8;
    float* dl = zs + kReadoutMaxK;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t* prow = a.pos + (int64_t)b * a.Smax;
    // Everything that depends on no other load is requested FIRST, so that the barrier inside valid_count drains one
    // round trip instead of this kernel paying four in a row: the final GraphNorm's accumulators / statistics and
    // parameters, the thread's first subgraph entries, the head weights of the wave's first class, the target.
    const bool early = 2 * C <= kBlock;
    GnCoefEarly E;
    if (early) gn_coef_early_issue(a.src, a.saved, C, E);
    const int TCp = 1 << a.tc_log2, rpbp = kBlock >> a.tc_log2;
    int64_t node_pre[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = (tid >> a.tc_log2) + u * rpbp;
        node_pre[u] = j < a.Smax ? prow[j] : -1;
    }
    float wh_pre[4] = {0.f, 0.f, 0.f, 0.f};
    const bool wh_early = C <= 4 * kWave && (tid >> 6) < K;
    if (wh_early)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = (tid & 63) + kWave * u;
            if (c < C) wh_pre[u] = a.Wh[(int64_t)(tid >> 6) * C + c];
        }
    (void)TCp;
    int* ids = reinterpret_cast<int*>(dl + kReadoutMaxK);  // [Smax] node id or -1 (a.stash)
    int cnt = 0;
    for (int j0 = 0; j0 < a.Smax; j0 += kBlock) {  // (valid_count, keeping what it read)
        const int j = j0 + tid;
        const int64_t pj = j < a.Smax ? prow[j] : -1;
        const bool okj = pj >= 0 && pj < a.n_nodes;
        if (a.stash && j < a.Smax) ids[j] = okj ? (int)pj : -1;
        cnt += __syncthreads_count(okj);
    }
    // the final GraphNorm's coefficients: copied from `saved`, or derived here from the accumulators the comb kernels' epilogues
    // added to (workgroup 0 writes `saved` for the two launches behind this one)
    if (early)
        gn_coef_early_finish(a.src, C, a.n_nodes, E, sums, coef_s, mu_rstd_s);
    else
        gn_fwd_coef_block(a.src, a.saved, C, a.n_nodes, sums, coef_s, mu_rstd_s);
    const float sc = readout_pool_scale(a.mode, cnt);
    // ---- pool the normalised rows; also sum xhat over the subgraph's rows ----
    const int TC = 1 << a.tc_log2, rpb = kBlock >> a.tc_log2;
    const int tc = tid & (TC - 1), tr = tid >> a.tc_log2;
    const int c0 = tc * VW;
    const bool ok = c0 < C;
    float accy[4] = {0.f, 0.f, 0.f, 0.f}, acch[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        float mu[VW], rstd[VW], scale[VW], shift[VW], al[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            mu[k] = mu_rstd_s[c0 + k];
            rstd[k] = mu_rstd_s[C + c0 + k];
            scale[k] = coef_s[c0 + k];
            shift[k] = coef_s[C + c0 + k];
            al[k] = a.alpha[c0 + k];
        }
        if (a.Smax <= 2 * rpb) {  // at most two entries per row slot: both were requested before the first barrier
            int it = 0;
            for (int j = tr; j < a.Smax; j += rpb, ++it) {
                const int64_t node = node_pre[it < 1 ? 0 : 1];
                if (node < 0 || node >= a.n_nodes) continue;
                float x[VW];
                if (VW == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(a.jk + node * a.ldj + c0);
                    x[0] = v.x; x[VW > 1 ? 1 : 0] = v.y; x[VW > 2 ? 2 : 0] = v.z; x[VW > 3 ? 3 : 0] = v.w;
                } else {
                    x[0] = a.jk[node * a.ldj + c0];
                }
#pragma unroll
                for (int k = 0; k < VW; ++k) {
                    accy[k] += fmaf(x[k], scale[k], shift[k]);
                    acch[k] += (x[k] - al[k] * mu[k]) * rstd[k];
                }
            }
        } else {
            // long subgraphs (em_user: 155 .. 499 nodes): a slot walks many entries, each a dependent pos -> row pair of round
            // trips — four entries per round, ids first, then the four rows (unconditional loads: a padding entry reads row
            // 0 and is dropped), added in the same (ascending j) order as the plain loop
            constexpr int U = 4;
            for (int j0 = tr; j0 < a.Smax; j0 += U * rpb) {
                int64_t nd[U];
                bool live[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * rpb;
                    nd[u] = j >= a.Smax ? -1 : a.stash ? (int64_t)ids[j] : (j0 == tr && u < 2) ? node_pre[u] : prow[j];
                }
                float x[U][VW];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    live[u] = nd[u] >= 0 && nd[u] < a.n_nodes;
                    const float* src = a.jk + (live[u] ? nd[u] : 0) * a.ldj + c0;
                    if (VW == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(src);
                        x[u][0] = v.x; x[u][VW > 1 ? 1 : 0] = v.y; x[u][VW > 2 ? 2 : 0] = v.z; x[u][VW > 3 ? 3 : 0] = v.w;
                    } else {
                        x[u][0] = src[0];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!live[u]) continue;
#pragma unroll
                    for (int k = 0; k < VW; ++k) {
                        accy[k] += fmaf(x[u][k], scale[k], shift[k]);
                        acch[k] += (x[u][k] - al[k] * mu[k]) * rstd[k];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid * 8 + k] = accy[k];
        red[tid * 8 + 4 + k] = acch[k];
    }
    __syncthreads();
    if (tr == 0 && ok) {
        for (int r = 1; r < rpb; ++r)
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                accy[k] += red[(r * TC + tc) * 8 + k];
                acch[k] += red[(r * TC + tc) * 8 + 4 + k];
            }
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            pooled_s[c0 + k] = accy[k] * sc;
            xh_s[c0 + k] = acch[k];
            a.pooled[(int64_t)b * C + c0 + k] = accy[k] * sc;
        }
    }
    __syncthreads();
    // ---- logits: wave w takes classes w, w+4, ... ----
    const int lane = tid & 63, w = tid >> 6;
    for (int k = w; k < K; k += kBlock / kWave) {
        const float* wr = a.Wh + (int64_t)k * C;
        float s = 0.f;
        if (wh_early && k == w) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = lane + kWave * u;
                if (c < C) s = fmaf(pooled_s[c], wh_pre[u], s);
            }
        } else {
            for (int c = lane; c < C; c += kWave) s = fmaf(pooled_s[c], wr[c], s);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) zs[k] = s + a.bh[k];
    }
    __syncthreads();
    // ---- loss term and d logits (mean reduction over B rows, or B*K elements for BCE) ----
    if (tid == 0) {
        const float gscale = a.gl[0] / (a.loss_mode == kLossCE ? (float)a.B : (float)a.B * (float)K);
        float term = 0.f;
        if (a.loss_mode == kLossCE) {
            float m = zs[0];
            for (int k = 1; k < K; ++k) m = fmaxf(m, zs[k]);
            float se = 0.f;
            for (int k = 0; k < K; ++k) se += expf(zs[k] - m);
            const float lse = m + logf(se);
            const int64_t t = ((const int64_t*)a.target)[b];
            for (int k = 0; k < K; ++k) dl[k] = gscale * (expf(zs[k] - lse) - (t == k ? 1.f : 0.f));
            term = (t >= 0 && t < K) ? lse - zs[t] : 0.f;
        } else {
            const float* y = (const float*)a.target + (int64_t)b * K;
            for (int k = 0; k < K; ++k) {
                const float z = zs[k];
                dl[k] = gscale * (1.f / (1.f + expf(-z)) - y[k]);
                term += fmaxf(z, 0.f) - z * y[k] + log1pf(expf(-fabsf(z)));  // torch's stable BCE-with-logits
            }
        }
        a.ws.loss_rows[b] = term;
    }
    __syncthreads();
    for (int k = tid; k < K; k += kBlock) {
        a.logits[(int64_t)b * K + k] = zs[k];
        a.ws.dlogits[(int64_t)b * K + k] = dl[k];
    }
    // ---- d pooled -> gradient of each pooled row, and this subgraph's share of the GraphNorm column sums ----
    double* part = a.ws.partial + (size_t)b * 2 * C;
    for (int c = tid; c < C; c += kBlock) {
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < K; ++k) s = fmaf(dl[k], a.Wh[(int64_t)k * C + c], s);
        const float dy = sc * s;
        a.ws.dys[(int64_t)b * C + c] = dy;
        if (a.bwd_acc) {
            gn_acc_add(a.bwd_acc, b % a.bwd_rep, 0, c, C, (double)cnt * (double)dy, kAccScaleBwd);
            gn_acc_add(a.bwd_acc, b % a.bwd_rep, 1, c, C, (double)dy * (double)xh_s[c], kAccScaleBwd);
        } else {
            part[c] = (double)cnt * (double)dy;
            part[C + c] = (double)dy * (double)xh_s[c];
        }
    }
}

struct R2Args {
    const float* pooled; ReadoutWs ws; int B, C, K, loss_mode;
    float *dWh, *dbh; int acc_head;
    float* loss;
    float* loss_sum;  // running sum of the step losses (nullable): += this step's loss, same thread, fixed order
    int64_t n_nodes;
    const float *gamma, *alpha, *saved;
    float *dgamma, *dbeta, *dalpha; int acc_gn;
};

// blocks [0,K): head gradient row k;  block K: mean loss;  blocks (K, K + ceil(C/4)]: GraphNorm backward finalize.
// dynamic LDS: max(B floats, kBlock*2 doubles)
__global__ __launch_bounds__(kBlock) void readout_reduce_kernel(R2Args a) {
    extern __shared__ double smd[];
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < a.K) {
        float* dl = reinterpret_cast<float*>(smd);
        for (int b = tid; b < a.B; b += kBlock) dl[b] = a.ws.dlogits[(int64_t)b * a.K + blk];
        __syncthreads();
        for (int c = tid; c < a.C; c += kBlock) {
            float s = 0.f;
#pragma unroll 8
            for (int b = 0; b < a.B; ++b) s = fmaf(dl[b], a.pooled[(int64_t)b * a.C + c], s);
            float* d = a.dWh + (int64_t)blk * a.C + c;
            *d = a.acc_head ? *d + s : s;
        }
        if (tid == 0) {
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s += dl[b];
            a.dbh[blk] = a.acc_head ? a.dbh[blk] + s : s;
        }
        return;
    }
    if (blk == a.K) {
        double part = 0.0;
        for (int b = tid; b < a.B; b += kBlock) part += (double)a.ws.loss_rows[b];
        smd[tid] = part;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if (tid < s) smd[tid] += smd[tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            const float l = (float)(smd[0] / (a.loss_mode == kLossCE ? (double)a.B : (double)a.B * (double)a.K));
            a.loss[0] = l;
            if (a.loss_sum) a.loss_sum[0] += l;
        }
        return;
    }
    gn_finalize_bwd_block(blk - a.K - 1, a.ws.partial, a.B, a.C, a.n_nodes, a.gamma, a.alpha, a.saved, a.dgamma, a.dbeta,
                          a.dalpha, a.acc_gn, a.ws.coef, smd);
}

// R3: d jk = Bx*x + K over all nodes (the part of the GraphNorm backward that does not depend on the row's own g)
__global__ __launch_bounds__(kBlock) void readout_dense_kernel(const float* __restrict__ x, int64_t ldx,
                                                               float* __restrict__ dx, int64_t lddx, int64_t N, int C,
                                                               int tc_log2, const float* __restrict__ coef) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int c0 = tc * 4;
    if (c0 >= C) return;
    const float4 Bx = *reinterpret_cast<const float4*>(coef + C + c0);
    const float4 K = *reinterpret_cast<const float4*>(coef + 2 * C + c0);
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t rr = r + u * stride;
            if (rr < N) v[u] = *reinterpret_cast<const float4*>(x + rr * ldx + c0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t rr = r + u * stride;
            if (rr >= N) continue;
            *reinterpret_cast<float4*>(dx + rr * lddx + c0) =
                make_float4(fmaf(Bx.x, v[u].x, K.x), fmaf(Bx.y, v[u].y, K.y), fmaf(Bx.z, v[u].z, K.z),
                            fmaf(Bx.w, v[u].w, K.w));
        }
    }
}

// R4 (ordered): d jk[n] += A * sum of g over the subgraphs holding n, WITHOUT atomics.  The whole pos matrix is staged
// in LDS (int32 node ids); one wave per entry: if an EARLIER entry names the same node the wave skips (that entry
// owns the node), otherwise it adds up every occurrence in (b, s) order (64 entries per ballot) and does the single
// read-modify-write of that row.  Bitwise repeatable however many subgraphs share a node; B*Smax <= 16 384.
constexpr int kOrdBlock = 1024;  // 16 waves: one entry per wave for Smax <= 16, so the per-entry latency chains overlap

__global__ __launch_bounds__(kOrdBlock) void readout_scatter_ordered_kernel(const int64_t* __restrict__ pos, int Smax,
                                                                         int n_pos, const float* __restrict__ dys,
                                                                         const float* __restrict__ coef,
                                                                         float* __restrict__ dx, int64_t lddx,
                                                                         int64_t n_nodes, int C) {
    extern __shared__ int32_t nodes[];
    for (int j = threadIdx.x; j < n_pos; j += kOrdBlock) {
        const int64_t p = pos[j];
        nodes[j] = (p >= 0 && p < n_nodes) ? (int32_t)p : -1;
    }
    __syncthreads();
    // grid = B * ceil(Smax / 16): workgroup (b, part) takes entries part*16 .. part*16+15 of subgraph b, one per wave
    constexpr int kWaves = kOrdBlock / kWave;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int parts = (Smax + kWaves - 1) / kWaves;
    const int b = blockIdx.x / parts;
    for (int s = (blockIdx.x % parts) * kWaves + w; s < Smax; s += Smax) {  // at most one entry per wave
        const int j = b * Smax + s;
        const int node = nodes[j];
        if (node < 0) continue;  // wave-uniform
        bool owned = false;      // an earlier entry names this node
        for (int j0 = 0; j0 < j && !owned; j0 += kWave) {
            const int jj = j0 + lane;
            owned = __any(jj < j && nodes[jj] == node);
        }
        if (owned) continue;
        // every lane takes part in the ballots (entries are spread over all 64 lanes); a lane owns the columns
        // lane*4 + 256*t, t < 4 (C <= 1024)
        float4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j0 = j; j0 < n_pos; j0 += kWave) {
            const int jj = j0 + lane;
            unsigned long long hits = __ballot(jj < n_pos && nodes[jj] == node);
            while (hits) {
                const int bit = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const float* grow = dys + (int64_t)((j0 + bit) / Smax) * C;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c = lane * 4 + kWave * 4 * t;
                    if (c < C) {
                        const float4 A = *reinterpret_cast<const float4*>(coef + c);
                        const float4 g = *reinterpret_cast<const float4*>(grow + c);
                        acc[t].x = fmaf(A.x, g.x, acc[t].x);
                        acc[t].y = fmaf(A.y, g.y, acc[t].y);
                        acc[t].z = fmaf(A.z, g.z, acc[t].z);
                        acc[t].w = fmaf(A.w, g.w, acc[t].w);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = lane * 4 + kWave * 4 * t;
            if (c < C) {
                float4* dst = reinterpret_cast<float4*>(dx + (int64_t)node * lddx + c);
                float4 o = *dst;
                o.x += acc[t].x; o.y += acc[t].y; o.z += acc[t].z; o.w += acc[t].w;
                *dst = o;
            }
        }
    }
}

// R3 + R4 as ONE launch when the batch's unique pooled rows are listed (glass_batch_labels: the pooled rows of a step ARE
// its labeled rows): workgroups [0, n_dense) write  d jk = Bx*x + K  on every row that is NOT pooled (label byte 0);
// the workgroups behind them take 16 listed rows each, one per wave, and write the row's full value
// Bx*x + K + A * (sum of g over the subgraphs holding it, in (b, s) order) — every row written once, no atomics, the same
// arithmetic in the same order as R3 followed by the ordered R4 (bitwise equal).
// Two-launch form (fin.acc != nullptr): the GraphNorm backward finalize is folded into this launch — every dense / sparse
// workgroup folds the exact accumulators the subgraph kernel added to and derives A | Bx | K itself (workgroup 0 writes the
// parameter gradients) — and the other roles of the reduce launch (head weight / bias gradient rows, mean loss) are extra
// workgroups behind the sparse ones.  Dynamic LDS: nodes[n_pos] int32 | (two-launch form) sums[2C] doubles | coef[3C].
struct BackfillFin {
    const long long* acc;  // nullptr: coefficients come from `coef` (three-launch form)
    int n_rep;
    const float *gamma, *alpha, *saved;
    float *dgamma, *dbeta, *dalpha;
    int acc_gn;
    // head / loss roles
    const float* pooled; const float* dlogits; const float* loss_rows;
    float *dWh, *dbh, *loss;
    int B, K, loss_mode, acc_head;
    float* loss_sum;  // nullable: += the step's loss
};

__global__ __launch_bounds__(kOrdBlock) void readout_backfill_kernel(const float* __restrict__ x, int64_t ldx,
                                                                    float* __restrict__ dx, int64_t lddx, int64_t N, int C,
                                                                    int tc_log2, const float* __restrict__ coef,
                                                                    const uint8_t* __restrict__ mask,
                                                                    const int64_t* __restrict__ pos, int Smax, int n_pos,
                                                                    const float* __restrict__ dys,
                                                                    const int32_t* __restrict__ lab_rows,
                                                                    const int32_t* __restrict__ lab_count, int n_dense, int n_sparse,
                                                                    BackfillFin fin) {
    extern __shared__ int32_t nodes[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n_dense + n_sparse) {  // head gradient row k, or the mean loss (two-launch form)
        const int blk = (int)blockIdx.x - n_dense - n_sparse;
        if (blk < fin.K) {
            float* dl = reinterpret_cast<float*>(nodes);  // [B]
            for (int b = tid; b < fin.B; b += kOrdBlock) dl[b] = fin.dlogits[(int64_t)b * fin.K + blk];
            __syncthreads();
            for (int c = tid; c < C; c += kOrdBlock) {
                float sum = 0.f;
#pragma unroll 8
                for (int b = 0; b < fin.B; ++b) sum = fmaf(dl[b], fin.pooled[(int64_t)b * C + c], sum);
                float* d = fin.dWh + (int64_t)blk * C + c;
                *d = fin.acc_head ? *d + sum : sum;
            }
            if (tid == 0) {
                float sum = 0.f;
                for (int b = 0; b < fin.B; ++b) sum += dl[b];
                fin.dbh[blk] = fin.acc_head ? fin.dbh[blk] + sum : sum;
            }
        } else if (tid < kWave) {  // mean loss: one wave, fixed order
            double part = 0.0;
            for (int b = tid; b < fin.B; b += kWave) part += (double)fin.loss_rows[b];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
            if (tid == 0) {
                const float l = (float)(part / (fin.loss_mode == kLossCE ? (double)fin.B : (double)fin.B * (double)fin.K));
                fin.loss[0] = l;
                if (fin.loss_sum) fin.loss_sum[0] += l;
            }
        }
        return;
    }
    if (fin.acc) {
        double* sums = reinterpret_cast<double*>(nodes + ((n_pos + 1) & ~1));
        float* coef_s = reinterpret_cast<float*>(sums + 2 * C);
        gn_acc_fold(fin.acc, C, 1, sums, kAccScaleBwd, fin.n_rep);
        for (int c = tid; c < C; c += kOrdBlock) {
            float A, Bx, K, da;
            gn_bwd_coeffs(sums[c], sums[C + c], (double)N, fin.gamma[c], fin.alpha[c], fin.saved[c], fin.saved[C + c], A, Bx, K, da);
            coef_s[c] = A;
            coef_s[C + c] = Bx;
            coef_s[2 * C + c] = K;
            if (blockIdx.x == 0) {
                if (fin.dgamma) fin.dgamma[c] = (fin.acc_gn ? fin.dgamma[c] : 0.f) + (float)sums[C + c];
                if (fin.dbeta) fin.dbeta[c] = (fin.acc_gn ? fin.dbeta[c] : 0.f) + (float)sums[c];
                if (fin.dalpha) fin.dalpha[c] = (fin.acc_gn ? fin.dalpha[c] : 0.f) + da;
            }
        }
        __syncthreads();
    }
    const float* cf = fin.acc ? reinterpret_cast<const float*>(reinterpret_cast<double*>(nodes + ((n_pos + 1) & ~1)) + 2 * C) : coef;  // (LDS or global)
    if ((int)blockIdx.x < n_dense) {
        const int TC = 1 << tc_log2, rpb = kOrdBlock >> tc_log2;
        const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
        const int c0 = tc * 4;
        if (c0 >= C) return;
        const float4 Bx = *reinterpret_cast<const float4*>(cf + C + c0);
        const float4 K = *reinterpret_cast<const float4*>(cf + 2 * C + c0);
        const int64_t stride = (int64_t)n_dense * rpb;
        for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < N; r += stride * 4) {
            float4 v[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + u * stride;
                live[u] = rr < N && mask[rr] == 0;
                if (live[u]) v[u] = *reinterpret_cast<const float4*>(x + rr * ldx + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + u * stride;
                if (!live[u]) continue;
                *reinterpret_cast<float4*>(dx + rr * lddx + c0) =
                    make_float4(fmaf(Bx.x, v[u].x, K.x), fmaf(Bx.y, v[u].y, K.y), fmaf(Bx.z, v[u].z, K.z),
                                fmaf(Bx.w, v[u].w, K.w));
            }
        }
        return;
    }
    constexpr int kWaves = kOrdBlock / kWave;
    const int n_lab = lab_count[0];
    const int first = ((int)blockIdx.x - n_dense) * kWaves;
    if (first >= n_lab) return;  // (workgroup-uniform)
    for (int j = threadIdx.x; j < n_pos; j += kOrdBlock) {
        const int64_t p = pos[j];
        nodes[j] = (p >= 0 && p < N) ? (int32_t)p : -1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (first + w >= n_lab) return;
    const int node = lab_rows[first + w];
    float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = 0; j0 < n_pos; j0 += kWave) {
        const int jj = j0 + lane;
        unsigned long long hits = __ballot(jj < n_pos && nodes[jj] == node);
        while (hits) {
            const int bit = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const float* grow = dys + (int64_t)((j0 + bit) / Smax) * C;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = lane * 4 + kWave * 4 * t;
                if (c < C) {
                    const float4 A = *reinterpret_cast<const float4*>(cf + c);
                    const float4 g = *reinterpret_cast<const float4*>(grow + c);
                    acc[t].x = fmaf(A.x, g.x, acc[t].x);
                    acc[t].y = fmaf(A.y, g.y, acc[t].y);
                    acc[t].z = fmaf(A.z, g.z, acc[t].z);
                    acc[t].w = fmaf(A.w, g.w, acc[t].w);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = lane * 4 + kWave * 4 * t;
        if (c < C) {
            const float4 Bx = *reinterpret_cast<const float4*>(cf + C + c);
            const float4 K = *reinterpret_cast<const float4*>(cf + 2 * C + c);
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)node * ldx + c);
            float4 o = make_float4(fmaf(Bx.x, xv.x, K.x), fmaf(Bx.y, xv.y, K.y), fmaf(Bx.z, xv.z, K.z), fmaf(Bx.w, xv.w, K.w));
            o.x += acc[t].x; o.y += acc[t].y; o.z += acc[t].z; o.w += acc[t].w;
            *reinterpret_cast<float4*>(dx + (int64_t)node * lddx + c) = o;
        }
    }
}

// R3 + R4 for ANY C (scalar accesses; the 17-wide layers of config/component.yml): one wave per node, lanes over the
// columns; a labeled (= pooled) node walks the padded pos matrix in (b, s) order and adds A * g of every subgraph that holds
// it — atomic-free, bitwise repeatable, no size limit on pos.
__global__ __launch_bounds__(kBlock) void readout_backfill_scalar_kernel(const float* __restrict__ x, int64_t ldx,
                                                                        float* __restrict__ dx, int64_t lddx, int64_t N, int C,
                                                                        const float* __restrict__ coef,
                                                                        const uint8_t* __restrict__ mask,
                                                                        const int64_t* __restrict__ pos, int Smax, int n_pos,
                                                                        const float* __restrict__ dys) {
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (node >= N) return;
    const bool lab = mask[node] != 0;  // wave-uniform
    for (int c = lane; c < C; c += kWave) {
        float acc = 0.f;
        if (lab)
            for (int j = 0; j < n_pos; ++j)
                if (pos[j] == node) acc = fmaf(coef[c], dys[(int64_t)(j / Smax) * C + c], acc);
        dx[node * lddx + c] = fmaf(coef[C + c], x[node * ldx + c], coef[2 * C + c]) + acc;
    }
}

// R4 beyond the ordered scatter's LDS staging, still without float atomics: the entries are bucketed by node (bucket.h) and a
// pooled node's row gets  d jk[n] += sum over its entries of A * g[subgraph]  with the sum taken in exact fixed point — the
// lists' arbitrary order does not reach the result.  One lane group (16 B per lane) per node; long lists by the whole
// workgroup.  The dense part (readout_dense_kernel) wrote every row before.
__global__ __launch_bounds__(kBlock) void readout_gather_add_kernel(const int32_t* __restrict__ off, const int32_t* __restrict__ list,
                                                                    const float* __restrict__ dys, const float* __restrict__ coef,
                                                                    float* __restrict__ dx, int64_t lddx, int64_t n_nodes, int C,
                                                                    int tc_log2) {
    __shared__ long long red[kBlock * 4 * 2];
    const int G = 1 << tc_log2, li = threadIdx.x & (G - 1), slot = threadIdx.x >> tc_log2, n_slot = kBlock >> tc_log2;
    const int64_t node0 = (int64_t)blockIdx.x * n_slot;
    const int c0 = li * 4;
    const bool ok = c0 < C;
    const float4 A = ok ? *reinterpret_cast<const float4*>(coef + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int kLong = 64;
    {
        const int64_t node = node0 + slot;
        const int beg = node < n_nodes ? off[node] : 0, end = node < n_nodes ? off[node + 1] : 0;
        if (ok && end > beg && end - beg < kLong) {
            ExactSum s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k].hi = s[k].lo = 0;
            for (int i = beg; i < end; ++i) {
                const float4 g = *reinterpret_cast<const float4*>(dys + (int64_t)list[i] * C + c0);
                s[0].add(A.x * g.x); s[1].add(A.y * g.y); s[2].add(A.z * g.z); s[3].add(A.w * g.w);
            }
            float4* d = reinterpret_cast<float4*>(dx + node * lddx + c0);
            float4 o = *d;
            o.x += s[0].value(); o.y += s[1].value(); o.z += s[2].value(); o.w += s[3].value();
            *d = o;
        }
    }
    for (int t = 0; t < n_slot; ++t) {
        const int64_t node = node0 + t;
        if (node >= n_nodes) break;
        const int beg = off[node], end = off[node + 1];
        if (end - beg < kLong) continue;  // workgroup-uniform
        ExactSum s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k].hi = s[k].lo = 0;
        if (ok)
            for (int i = beg + slot; i < end; i += n_slot) {
                const float4 g = *reinterpret_cast<const float4*>(dys + (int64_t)list[i] * C + c0);
                s[0].add(A.x * g.x); s[1].add(A.y * g.y); s[2].add(A.z * g.z); s[3].add(A.w * g.w);
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(threadIdx.x * 4 + k) * 2] = s[k].hi;
            red[(threadIdx.x * 4 + k) * 2 + 1] = s[k].lo;
        }
        __syncthreads();
        if (slot == 0 && ok) {
            float add[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ExactSum tsum{0, 0};
                for (int r = 0; r < n_slot; ++r) {
                    tsum.hi += red[(((r << tc_log2) + li) * 4 + k) * 2];
                    tsum.lo += red[(((r << tc_log2) + li) * 4 + k) * 2 + 1];
                }
                add[k] = tsum.value();
            }
            float4* d = reinterpret_cast<float4*>(dx + node * lddx + c0);
            float4 o = *d;
            o.x += add[0]; o.y += add[1]; o.z += add[2]; o.w += add[3];
            *d = o;
        }
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_readout_supported(int64_t C, int64_t K, int pool_mode) {
    return (C > 0 && C <= 4 * kBlock && K > 0 && K <= kReadoutMaxK &&
            (pool_mode == GLASS_POOL_SUM || pool_mode == GLASS_POOL_MEAN || pool_mode == GLASS_POOL_SIZE)) ? 1 : 0;
}

extern "C" int64_t glass_readout_ws_bytes(int64_t B, int64_t C, int64_t K) {
    if (B <= 0 || C <= 0 || K <= 0) return GLASS_E_ARG;
    return (int64_t)sizeof(double) * 2 * B * C + (int64_t)sizeof(float) * (4 * C + B * C + B * K + B) + 64;
}

// scratch of the exact large-batch scatter (0: the batch fits the ordered scatter's LDS staging, nothing needed)
extern "C" int64_t glass_readout_scatter_ws_bytes(int64_t n_nodes, int64_t B, int64_t Smax) {
    if (n_nodes <= 0 || B <= 0 || Smax <= 0) return GLASS_E_ARG;
    if (B * Smax <= kReadoutOrderedMax) return 0;
    return (int64_t)sizeof(int32_t) * bucket_ws_words(n_nodes, B, Smax, false);
}

extern "C" int glass_readout_train_f32(const float* jk, int64_t ldj, const float* gn_saved, const float* gamma,
                                       const float* alpha, const int64_t* pos, int64_t B, int64_t Smax, int pool_mode,
                                       const float* Wh, const float* bh, const void* target, int loss_mode, int64_t K,
                                       const float* grad_loss, float* pooled, float* logits, float* loss, float* djk,
                                       int64_t lddj, float* dWh, float* dbh, int acc_head, float* dgamma, float* dbeta,
                                       float* dalpha, int acc_gn, void* ws, int64_t n_nodes, int64_t C,
                                       const uint8_t* mask, const int32_t* lab_rows, const int32_t* lab_count,
                                       const glass_gn_src* gn_src, int64_t* gn_bwd_acc, int gn_bwd_rep, void* scatter_ws,
                                       float* loss_sum, void* stream) {
    GLASS_REQUIRE(jk && gn_saved && gamma && alpha && pos && Wh && bh && target && grad_loss && pooled && logits && loss &&
                      djk && dWh && dbh && ws,
                  "readout_train: null pointer");
    GLASS_REQUIRE(B > 0 && Smax > 0 && Smax < (1ll << 31) && n_nodes > 0 && ldj >= C && lddj >= C && B * K < (1ll << 30),
                  "readout_train: bad sizes");
    if (!glass_readout_supported(C, K, pool_mode) || (loss_mode != kLossCE && loss_mode != kLossBCE)) {
        set_error("readout_train: unsupported shape/mode (C=%lld K=%lld pool=%d loss=%d): C %% 4 == 0, C <= 1024, "
                  "K <= 256, pool sum|mean|size", (long long)C, (long long)K, pool_mode, loss_mode);
        return GLASS_E_UNSUPPORTED;
    }
    const bool vec = C % 4 == 0 && ldj % 4 == 0 && lddj % 4 == 0 && aligned16(jk) && aligned16(djk) && aligned16(pooled);
    GLASS_REQUIRE(aligned16(ws) && (vec || mask),
                  "readout_train: operands must be 16-B aligned with C %% 4 == 0 and ld %% 4 == 0 — or the label bytes given "
                  "(scalar form for any C)");
    if (B * Smax > kReadoutOrderedMax && !scatter_ws) {
        // (this entry promises bitwise repeatable results: it never falls back to float atomics)
        set_error("readout_train: B*Smax = %lld exceeds the ordered scatter's LDS staging (%d entries): pass scatter_ws "
                  "(glass_readout_scatter_ws_bytes bytes)", (long long)(B * Smax), (int)kReadoutOrderedMax);
        return GLASS_E_WS;
    }
    hipStream_t st = (hipStream_t)stream;
    const ReadoutWs w = carve_ws(ws, B, C, K);
    const int tc = pow2_ceil_cap(vec ? C / 4 : C, kBlock);
    int tc_log2 = 0;
    while ((1 << tc_log2) < tc) ++tc_log2;
    GnExactSrc esrc{nullptr, 1, kAccRep, nullptr, nullptr, nullptr, 0.f, nullptr};
    if (gn_src) {
        GLASS_REQUIRE(gn_src->acc && gn_src->gamma && gn_src->beta && gn_src->alpha && gn_src->n_src >= 1 &&
                          C % gn_src->n_src == 0 && gn_src->n_rep >= 1 && gn_src->n_rep <= kAccRep,
                      "readout_train: bad gn_src (n_src accumulator blocks of C / n_src columns, all pointers set)");
        esrc = GnExactSrc{reinterpret_cast<const long long*>(gn_src->acc), (int)gn_src->n_src, (int)gn_src->n_rep, gn_src->gamma, gn_src->beta,
                          gn_src->alpha, gn_src->eps, const_cast<float*>(gn_saved)};
    }
    // two-launch form: the subgraph kernel adds its backward column sums to exact accumulators, the one-launch backfill folds
    // them and carries the head-gradient / loss roles — no reduce launch (needs the listed pooled rows: see below)
    const bool two = gn_bwd_acc != nullptr && vec && mask && lab_rows && lab_count && B * Smax <= kReadoutOrderedMax;
    GLASS_REQUIRE(!gn_bwd_acc || (gn_bwd_rep >= 1 && gn_bwd_rep <= kAccRep), "readout_train: gn_bwd_rep = replicas of gn_bwd_acc (1 .. 16)");
    // long subgraphs: the row's node ids staged in LDS by the counting pass (while everything stays within 64 KB)
    const int stash = Smax > 2 * (kBlock >> tc_log2) && n_nodes < (1ll << 31) && sizeof(float) * (size_t)(10 * C + kBlock * 8 + 2 * kReadoutMaxK + Smax) <= 64 * 1024;
    R1Args a1{jk, ldj, gn_saved, alpha, pos, (int)Smax, pool_mode, Wh, bh, target, loss_mode, (int)B, (int)C, (int)K,
              grad_loss, pooled, logits, w, n_nodes, tc_log2, esrc, stash, two ? reinterpret_cast<long long*>(gn_bwd_acc) : nullptr,
              gn_bwd_rep};
    const size_t lds1 = sizeof(float) * (size_t)(10 * C + kBlock * 8 + 2 * kReadoutMaxK + (stash ? Smax : 0));
    if (vec)
        hipLaunchKernelGGL(readout_subgraph_kernel<4>, dim3((unsigned)B), dim3(kBlock), lds1, st, a1);
    else
        hipLaunchKernelGGL(readout_subgraph_kernel<1>, dim3((unsigned)B), dim3(kBlock), lds1, st, a1);
    R2Args a2{pooled, w, (int)B, (int)C, (int)K, loss_mode, dWh, dbh, acc_head, loss, loss_sum, n_nodes, gamma, alpha, gn_saved,
              dgamma, dbeta, dalpha, acc_gn};
    size_t lds2 = sizeof(double) * kBlock * 2;
    if (sizeof(float) * (size_t)B > lds2) lds2 = sizeof(float) * (size_t)B;
    GLASS_REQUIRE(lds2 <= 64 * 1024, "readout_train: batch too large for the LDS staging");
    if (!two) hipLaunchKernelGGL(readout_reduce_kernel, dim3((unsigned)(K + 1 + ceil_div(C, kFinCols))), dim3(kBlock), lds2, st, a2);
    if (!vec) {  // any C: scalar dense + sparse part as one launch (needs the label bytes = the pooled rows of this pos)
        hipLaunchKernelGGL(readout_backfill_scalar_kernel, dim3((unsigned)ceil_div(n_nodes, kBlock / kWave)), dim3(kBlock), 0, st,
                           jk, ldj, djk, lddj, n_nodes, (int)C, w.coef, mask, pos, (int)Smax, (int)(B * Smax), w.dys);
        return launch_status("glass_readout_train_f32");
    }
    if (mask && lab_rows && lab_count && B * Smax <= kReadoutOrderedMax) {
        // the pooled rows are listed (glass_batch_labels on this pos): dense and sparse part as ONE launch
        const int rpb1 = kOrdBlock / tc;
        int64_t n_dense = ceil_div(n_nodes, (int64_t)rpb1 * 4);
        if (n_dense > 2048) n_dense = 2048;
        const int64_t n_sparse = ceil_div(B * Smax, (int64_t)(kOrdBlock / kWave));
        BackfillFin fin{};
        size_t lds3 = sizeof(int32_t) * (size_t)(B * Smax);
        unsigned extra = 0;
        if (two) {
            fin = BackfillFin{reinterpret_cast<const long long*>(gn_bwd_acc), gn_bwd_rep, gamma, alpha, gn_saved, dgamma, dbeta, dalpha,
                              acc_gn, pooled, w.dlogits, w.loss_rows, dWh, dbh, loss, (int)B, (int)K, loss_mode, acc_head, loss_sum};
            lds3 = sizeof(int32_t) * (size_t)((B * Smax + 1) & ~1ll) + sizeof(double) * 2 * (size_t)C + sizeof(float) * 3 * (size_t)C;
            if (lds3 < sizeof(float) * (size_t)B) lds3 = sizeof(float) * (size_t)B;
            extra = (unsigned)K + 1;
            if (lds3 > 64 * 1024)
                (void)hipFuncSetAttribute((const void*)readout_backfill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        }
        hipLaunchKernelGGL(readout_backfill_kernel, dim3((unsigned)(n_dense + n_sparse) + extra), dim3(kOrdBlock), lds3, st, jk, ldj, djk,
                           lddj, n_nodes, (int)C, tc_log2, w.coef, mask, pos, (int)Smax, (int)(B * Smax), w.dys, lab_rows, lab_count,
                           (int)n_dense, (int)n_sparse, fin);
        return launch_status("glass_readout_train_f32");
    }
    const int rpb = kBlock / tc;
    int64_t blocks = ceil_div(n_nodes, (int64_t)rpb * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(readout_dense_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, jk, ldj, djk, lddj, n_nodes, (int)C,
                       tc_log2, w.coef);
    if (B * Smax <= kReadoutOrderedMax) {
        const int64_t parts = ceil_div(Smax, (int64_t)(kOrdBlock / kWave));
        hipLaunchKernelGGL(readout_scatter_ordered_kernel, dim3((unsigned)(B * parts)), dim3(kOrdBlock), sizeof(int32_t) * (size_t)(B * Smax),
                           st, pos, (int)Smax, (int)(B * Smax), w.dys, w.coef, djk, lddj, n_nodes, (int)C);
    } else if (scatter_ws) {
        // exact, atomic-free: entries bucketed by node, then one gather-add per pooled node (glass_readout_scatter_ws_bytes)
        BucketLists bl;
        int rc = bucket_build(pos, B, Smax, -1, false, n_nodes, scatter_ws, st, &bl);
        if (rc) return rc;
        hipLaunchKernelGGL(readout_gather_add_kernel, dim3((unsigned)ceil_div(n_nodes, (int64_t)(kBlock / tc))), dim3(kBlock), 0, st,
                           bl.off, bl.list, w.dys, w.coef, djk, lddj, n_nodes, (int)C, tc_log2);
    }
    return launch_status("glass_readout_train_f32");
}
