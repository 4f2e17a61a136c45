// K7: subgraph pooling straight from the padded node matrix.  Replaces pad2batch + emb[pos] +
// global_{add,mean,max}_pool / GraphSizeNorm (reference impl/models.py:346-350, 294-319;
// impl/utils.py:18-29).  One workgroup per subgraph: TC lanes x 16 B cover an embedding row,
// 256/TC row slots walk the padded node list; slots are combined through LDS in fixed order.
#include "bucket.h"
#include "common.h"

#include <float.h>

namespace glass {

template <int VW> struct P;
template <> struct P<4> {
    float a[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 v = *reinterpret_cast<const float4*>(p);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    }
};
template <> struct P<1> {
    float a[1];
    __device__ __forceinline__ void load(const float* p) { a[0] = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = a[0]; }
};

// number of non-padding entries in row b of pos (block-wide, every thread gets the result)
__device__ __forceinline__ int count_valid(const int64_t* __restrict__ prow, int Smax, int64_t n_nodes) {
    int cnt = 0;
    for (int j0 = 0; j0 < Smax; j0 += kBlock) {
        const int j = j0 + threadIdx.x;
        const bool ok = j < Smax && prow[j] >= 0 && prow[j] < n_nodes;
        cnt += __syncthreads_count(ok);
    }
    return cnt;
}

__device__ __forceinline__ float pool_scale(int mode, int cnt) {
    if (mode == GLASS_POOL_MEAN) return 1.0f / (float)(cnt > 0 ? cnt : 1);   // scatter_mean: count clamped >= 1
    if (mode == GLASS_POOL_SIZE) return cnt > 0 ? 1.0f / sqrtf((float)cnt) : 0.f;  // GraphSizeNorm: n_b^-1/2
    return 1.0f;
}

template <int VW>
__global__ __launch_bounds__(kBlock) void pool_fwd_kernel(const float* __restrict__ emb, int64_t lde,
                                                          const int64_t* __restrict__ pos, int Smax, int mode,
                                                          float* __restrict__ out, int64_t ldo,
                                                          int32_t* __restrict__ argmax, int64_t n_nodes, int C,
                                                          int tc_log2) {
    __shared__ float lds_v[kBlock * VW];
    __shared__ int lds_j[kBlock * VW];
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int b = blockIdx.x;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    const bool ok = c0 < C;
    const int64_t* prow = pos + (int64_t)b * Smax;
    const int cnt = count_valid(prow, Smax, n_nodes);
    float acc[VW];
    int best[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        acc[k] = (mode == GLASS_POOL_MAX) ? -FLT_MAX : 0.f;
        best[k] = INT32_MAX;
    }
    for (int j = tr; j < Smax; j += rpb) {
        const int64_t node = prow[j];
        if (node < 0 || node >= n_nodes || !ok) continue;
        P<VW> v;
        v.load(emb + node * lde + c0);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            if (mode == GLASS_POOL_MAX) {
                if (v.a[k] > acc[k]) {  // strict: first occurrence wins a tie
                    acc[k] = v.a[k];
                    best[k] = j;
                }
            } else {
                acc[k] += v.a[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        lds_v[threadIdx.x * VW + k] = acc[k];
        lds_j[threadIdx.x * VW + k] = best[k];
    }
    __syncthreads();
    if (tr != 0 || !ok) return;
    for (int r = 1; r < rpb; ++r) {
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const float ov = lds_v[(r * TC + tc) * VW + k];
            const int oj = lds_j[(r * TC + tc) * VW + k];
            if (mode == GLASS_POOL_MAX) {
                if (ov > acc[k] || (ov == acc[k] && oj < best[k])) {
                    acc[k] = ov;
                    best[k] = oj;
                }
            } else {
                acc[k] += ov;
            }
        }
    }
    const float sc = pool_scale(mode, cnt);
    P<VW> o;
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        if (mode == GLASS_POOL_MAX) {
            const bool any = best[k] != INT32_MAX;
            o.a[k] = any ? acc[k] : 0.f;  // empty segment -> 0 (torch_scatter)
            if (argmax && c0 + k < C) argmax[(int64_t)b * C + c0 + k] = any ? (int32_t)prow[best[k]] : -1;
        } else {
            o.a[k] = acc[k] * sc;
        }
    }
    if (VW == 1 || c0 + VW <= C) o.store(out + (int64_t)b * ldo + c0);
}

template <int VW>
__global__ __launch_bounds__(kBlock) void pool_bwd_kernel(const float* __restrict__ dout, int64_t ldd,
                                                          const int64_t* __restrict__ pos, int Smax, int mode,
                                                          const int32_t* __restrict__ argmax,
                                                          float* __restrict__ demb, int64_t lde, int64_t n_nodes,
                                                          int C, int tc_log2) {
    const int TC = 1 << tc_log2, rpb = kBlock >> tc_log2;
    const int tc = threadIdx.x & (TC - 1), tr = threadIdx.x >> tc_log2;
    const int b = blockIdx.x;
    const int c0 = (blockIdx.y * TC + tc) * VW;
    const int64_t* prow = pos + (int64_t)b * Smax;
    const int cnt = count_valid(prow, Smax, n_nodes);
    if (c0 >= C) return;
    P<VW> g;
    g.load(dout + (int64_t)b * ldd + c0);
    if (mode == GLASS_POOL_MAX) {
        if (tr != 0) return;
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const int32_t node = argmax[(int64_t)b * C + c0 + k];
            if (node >= 0) atomicAdd(demb + (int64_t)node * lde + c0 + k, g.a[k]);
        }
        return;
    }
    const float sc = pool_scale(mode, cnt);
#pragma unroll
    for (int k = 0; k < VW; ++k) g.a[k] *= sc;
    for (int j = tr; j < Smax; j += rpb) {
        const int64_t node = prow[j];
        if (node < 0 || node >= n_nodes) continue;
        float* dst = demb + node * lde + c0;
#pragma unroll
        for (int k = 0; k < VW; ++k) atomicAdd(dst + k, g.a[k]);
    }
}

// Ordered, atomic-free backward for sum / mean / size pooling (demb zero-filled by the caller): the whole pos
// matrix is staged in LDS as int32 node ids together with every subgraph's scale; one wave per entry: if an EARLIER
// entry names the same node the wave skips (that entry owns the node), otherwise it adds up every occurrence in
// (b, s) order (64 entries per ballot) and stores the row.  Bitwise repeatable however many subgraphs share a node.
__global__ __launch_bounds__(kBlock) void pool_bwd_ordered_kernel(const float* __restrict__ dout, int64_t ldd,
                                                                  const int64_t* __restrict__ pos, int Smax, int B,
                                                                  int mode, float* __restrict__ demb, int64_t lde,
                                                                  int64_t n_nodes, int C) {
    extern __shared__ int32_t sm_i[];
    int32_t* nodes = sm_i;                                         // [B*Smax]
    float* scale = reinterpret_cast<float*>(sm_i + (size_t)B * Smax);  // [B]
    const int n_pos = B * Smax;
    for (int j = threadIdx.x; j < n_pos; j += kBlock) {
        const int64_t p = pos[j];
        nodes[j] = (p >= 0 && p < n_nodes) ? (int32_t)p : -1;
    }
    __syncthreads();
    for (int bb = threadIdx.x; bb < B; bb += kBlock) {
        int cnt = 0;
        for (int s = 0; s < Smax; ++s) cnt += nodes[bb * Smax + s] >= 0;
        scale[bb] = pool_scale(mode, cnt);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    for (int s = w; s < Smax; s += kBlock / kWave) {
        const int j = b * Smax + s;
        const int node = nodes[j];
        if (node < 0) continue;  // wave-uniform
        bool owned = false;      // an earlier entry names this node
        for (int j0 = 0; j0 < j && !owned; j0 += kWave) {
            const int jj = j0 + lane;
            owned = __any(jj < j && nodes[jj] == node);
        }
        if (owned) continue;
        for (int c0 = 0; c0 < C; c0 += kWave) {  // 64 columns per pass, one per lane (any C, any alignment)
            const int c = c0 + lane;
            float acc = 0.f;
            for (int j0 = j; j0 < n_pos; j0 += kWave) {
                const int jj = j0 + lane;
                unsigned long long hits = __ballot(jj < n_pos && nodes[jj] == node);
                while (hits) {
                    const int bit = __ffsll((long long)hits) - 1;
                    hits &= hits - 1;
                    const int bb = (j0 + bit) / Smax;
                    if (c < C) acc = fmaf(dout[(int64_t)bb * ldd + c], scale[bb], acc);
                }
            }
            if (c < C) demb[(int64_t)node * lde + c] = acc;
        }
    }
}

// ---- node pairs (Smax = 2): the link-prediction batches of the pre-training path (reference GNNEmb.py:144: 131 072 edge /
// non-edge pairs per step; impl/models.py:498-503 pools them with the same mean).  One workgroup per subgraph is the wrong
// shape for two entries: here G lanes cover an embedding row and a workgroup takes 256 / G pairs.
template <int VW>
__global__ __launch_bounds__(kBlock) void pair_fwd_kernel(const float* __restrict__ emb, int64_t lde,
                                                          const int64_t* __restrict__ pairs, int64_t B, int mode,
                                                          float* __restrict__ out, int64_t ldo, int64_t n_nodes, int C,
                                                          int g_log2) {
    const int G = 1 << g_log2, li = threadIdx.x & (G - 1);
    const int64_t p = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> g_log2;
    if (p >= B) return;
    const int64_t a = pairs[2 * p], b = pairs[2 * p + 1];
    const bool va = a >= 0 && a < n_nodes, vb = b >= 0 && b < n_nodes;
    const float sc = pool_scale(mode, (int)va + (int)vb);
    for (int c0 = li * VW; c0 < C; c0 += G * VW) {
        P<VW> x, y, o;
#pragma unroll
        for (int k = 0; k < VW; ++k) x.a[k] = y.a[k] = 0.f;
        if (va) x.load(emb + a * lde + c0);
        if (vb) y.load(emb + b * lde + c0);
#pragma unroll
        for (int k = 0; k < VW; ++k) o.a[k] = (x.a[k] + y.a[k]) * sc;
        o.store(out + p * ldo + c0);
    }
}

// Backward of the pair pool without a float atomic: the entries (pair, side) are bucketed by node — counts by integer
// atomics, offsets by a one-workgroup scan, entry lists filled through integer cursors (their order inside a node's list
// is arbitrary) — and each node's row is the sum of its entries' scaled gradient rows taken in EXACT fixed point
// (hi * 2^-20 + lo * 2^-60 in two 64-bit integers per column: integer addition commutes, so the arbitrary list order does
// not reach the result).  Bitwise repeatable; every row of demb is written (rows without an entry: zeros).
__global__ __launch_bounds__(kBlock) void pair_zero_kernel(int32_t* __restrict__ off, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) off[i] = 0;
}

// counts per node (off[node + 1]) and, from the same returning atomic, the entry's rank inside its node's list
// dedup_smax > 0 (= Smax): a node named several times by one padded row is listed once for that row (its first entry;
// the others get rank -1) — the max-pool backward asks "which subgraphs hold this node", not "how often"
__global__ __launch_bounds__(kBlock) void pair_rank_kernel(const int64_t* __restrict__ pairs, int64_t n_entries,
                                                           int64_t n_nodes, int32_t* __restrict__ off,
                                                           int32_t* __restrict__ rank, int dedup_smax) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_entries) return;
    const int64_t node = pairs[e];
    if (node < 0 || node >= n_nodes) return;
    if (dedup_smax > 0) {
        const int64_t row0 = e - e % dedup_smax;
        for (int64_t k = row0; k < e; ++k)
            if (pairs[k] == node) {
                rank[e] = -1;
                return;
            }
    }
    rank[e] = atomicAdd(off + node + 1, 1);
}

// off[0 .. n_nodes] (off[0] = 0, off[i + 1] = entries of node i) -> inclusive prefix sums in place: one workgroup, 4096
// elements per round (b128 per thread, wave scans by lane shuffles, the 16 wave totals through LDS, a running carry)
__global__ __launch_bounds__(1024) void pair_scan_kernel(int32_t* __restrict__ off, int64_t n) {
    __shared__ int32_t wsum[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int32_t carry = 0;
    for (int64_t base = 0; base < n; base += 4096) {
        const int64_t i = base + 4 * threadIdx.x;
        int32_t v[4] = {0, 0, 0, 0};
        if (i + 3 < n) {
            const int4 q = *reinterpret_cast<const int4*>(off + i);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else {
            for (int k = 0; k < 4; ++k)
                if (i + k < n) v[k] = off[i + k];
        }
        v[1] += v[0];
        v[2] += v[1];
        v[3] += v[2];
        int32_t incl = v[3];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int32_t t = wsum[k];
            if (k < w) before += t;
            total += t;
        }
        const int32_t excl = carry + before + incl - v[3];
        if (i + 3 < n) {
            *reinterpret_cast<int4*>(off + i) = make_int4(excl + v[0], excl + v[1], excl + v[2], excl + v[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (i + k < n) off[i + k] = excl + v[k];
        }
        carry += total;
        __syncthreads();
    }
}

// list[off[node] + rank] = (pair << 1) | (both entries of the pair valid): what the gather needs without reading pairs.
// Subgraphs of any padded width (Smax != 2): the entry's subgraph index; its scale comes from the scale array.
__global__ __launch_bounds__(kBlock) void pair_fill_kernel(const int64_t* __restrict__ pairs, int64_t n_entries, int Smax,
                                                           int pair_form, int64_t n_nodes, const int32_t* __restrict__ off,
                                                           const int32_t* __restrict__ rank, int32_t* __restrict__ list) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n_entries) return;
    const int64_t node = pairs[e];
    if (node < 0 || node >= n_nodes || rank[e] < 0) return;
    if (pair_form) {
        const int64_t other = pairs[e ^ 1];
        list[off[node] + rank[e]] = (int32_t)((e >> 1) << 1) | (int32_t)(other >= 0 && other < n_nodes);
    } else {
        list[off[node] + rank[e]] = (int32_t)(e / Smax);
    }
}

// scale[b] of every subgraph (one wave per padded row)
__global__ __launch_bounds__(kBlock) void pool_scale_kernel(const int64_t* __restrict__ pos, int B, int Smax, int mode,
                                                            int64_t n_nodes, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t* prow = pos + (int64_t)b * Smax;
    int cnt = 0;
    for (int j0 = 0; j0 < Smax; j0 += kWave) {
        const int j = j0 + lane;
        cnt += __popcll(__ballot(j < Smax && prow[j] >= 0 && prow[j] < n_nodes));
    }
    if (lane == 0) scale[b] = pool_scale(mode, cnt);
}

constexpr int kPairLongList = 64;  // entries from which a node's list is summed by the whole workgroup

template <int VW>
__global__ __launch_bounds__(kBlock) void pair_gather_kernel(const float* __restrict__ dout, int64_t ldd,
                                                             int mode, const float* __restrict__ scale,
                                                             const int32_t* __restrict__ off, const int32_t* __restrict__ list,
                                                             float* __restrict__ demb, int64_t lde, int64_t n_nodes, int C,
                                                             int g_log2) {
    __shared__ long long red[kBlock * VW * 2];
    const int G = 1 << g_log2, li = threadIdx.x & (G - 1), slot = threadIdx.x >> g_log2, n_slot = kBlock >> g_log2;
    const int64_t node0 = (int64_t)blockIdx.x * n_slot;
    const float sc1 = pool_scale(mode, 1), sc2 = pool_scale(mode, 2);
    auto scaled_row = [&](int e, int c0, float (&v)[VW]) __attribute__((always_inline)) {
        P<VW> g;
        const int b = scale ? e : (e >> 1);  // (workgroup-uniform choice: pairs carry their count in bit 0)
        g.load(dout + (int64_t)b * ldd + c0);
        const float sc = scale ? scale[b] : ((e & 1) ? sc2 : sc1);
#pragma unroll
        for (int k = 0; k < VW; ++k) v[k] = g.a[k] * sc;
    };
    // short lists: one lane group per node
    {
        const int64_t node = node0 + slot;
        const int beg = node < n_nodes ? off[node] : 0, end = node < n_nodes ? off[node + 1] : 0;
        if (node < n_nodes && end - beg < kPairLongList) {
            for (int c0 = li * VW; c0 < C; c0 += G * VW) {
                ExactSum s[VW];
#pragma unroll
                for (int k = 0; k < VW; ++k) s[k].hi = s[k].lo = 0;
                int i = beg;
                for (; i + 3 < end; i += 4) {  // four independent chains of loads in flight
                    float v0[VW], v1[VW], v2[VW], v3[VW];
                    const int e0 = list[i], e1 = list[i + 1], e2 = list[i + 2], e3 = list[i + 3];
                    scaled_row(e0, c0, v0);
                    scaled_row(e1, c0, v1);
                    scaled_row(e2, c0, v2);
                    scaled_row(e3, c0, v3);
#pragma unroll
                    for (int k = 0; k < VW; ++k) {
                        s[k].add(v0[k]);
                        s[k].add(v1[k]);
                        s[k].add(v2[k]);
                        s[k].add(v3[k]);
                    }
                }
                for (; i < end; ++i) {
                    float v0[VW];
                    scaled_row(list[i], c0, v0);
#pragma unroll
                    for (int k = 0; k < VW; ++k) s[k].add(v0[k]);
                }
                P<VW> o;
#pragma unroll
                for (int k = 0; k < VW; ++k) o.a[k] = s[k].value();
                o.store(demb + node * lde + c0);
            }
        }
    }
    // long lists: the whole workgroup takes one node at a time (its lane groups stride over the entries)
    for (int t = 0; t < n_slot; ++t) {
        const int64_t node = node0 + t;
        if (node >= n_nodes) break;
        const int beg = off[node], end = off[node + 1];
        if (end - beg < kPairLongList) continue;  // workgroup-uniform
        for (int ch = 0; ch * G * VW < C; ++ch) {  // uniform trip count: barriers inside
            const int c0 = (ch * G + li) * VW;
            ExactSum s[VW];
#pragma unroll
            for (int k = 0; k < VW; ++k) s[k].hi = s[k].lo = 0;
            if (c0 < C)
                for (int i = beg + slot; i < end; i += n_slot) {
                    float v0[VW];
                    scaled_row(list[i], c0, v0);
#pragma unroll
                    for (int k = 0; k < VW; ++k) s[k].add(v0[k]);
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                red[(threadIdx.x * VW + k) * 2] = s[k].hi;
                red[(threadIdx.x * VW + k) * 2 + 1] = s[k].lo;
            }
            __syncthreads();
            if (slot == 0 && c0 < C) {
                P<VW> o;
#pragma unroll
                for (int k = 0; k < VW; ++k) {
                    ExactSum tsum{0, 0};
                    for (int r = 0; r < n_slot; ++r) {
                        tsum.hi += red[(((r << g_log2) + li) * VW + k) * 2];
                        tsum.lo += red[(((r << g_log2) + li) * VW + k) * 2 + 1];
                    }
                    o.a[k] = tsum.value();
                }
                o.store(demb + node * lde + c0);
            }
        }
    }
}

// Max-pool backward on the same lists (deduplicated per row): node n's row is, per column, the sum of dout[b][c] over the
// subgraphs b that hold n AND whose argmax for that column is n — exact sums again, every row written.
template <int VW>
__global__ __launch_bounds__(kBlock) void pool_max_gather_kernel(const float* __restrict__ dout, int64_t ldd,
                                                                 const int32_t* __restrict__ argmax,
                                                                 const int32_t* __restrict__ off, const int32_t* __restrict__ list,
                                                                 float* __restrict__ demb, int64_t lde, int64_t n_nodes, int C,
                                                                 int g_log2) {
    const int G = 1 << g_log2, li = threadIdx.x & (G - 1), slot = threadIdx.x >> g_log2, n_slot = kBlock >> g_log2;
    const int64_t node = (int64_t)blockIdx.x * n_slot + slot;
    if (node >= n_nodes) return;
    const int beg = off[node], end = off[node + 1];
    for (int c0 = li * VW; c0 < C; c0 += G * VW) {
        ExactSum s[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) s[k].hi = s[k].lo = 0;
        for (int i = beg; i < end; ++i) {
            const int64_t b = list[i];
            P<VW> g;
            g.load(dout + b * ldd + c0);
#pragma unroll
            for (int k = 0; k < VW; ++k)
                if (argmax[b * C + c0 + k] == (int32_t)node) s[k].add(g.a[k]);
        }
        P<VW> o;
#pragma unroll
        for (int k = 0; k < VW; ++k) o.a[k] = s[k].value();
        o.store(demb + node * lde + c0);
    }
}

constexpr int64_t kPoolOrderedMax = 12288;  // pos entries (+ B scales) staged in LDS: <= 64 KiB

}  // namespace glass

using namespace glass;

static int pool_args_ok(const void* a, const void* pos, const void* o, int64_t B, int64_t Smax, int mode, int64_t C,
                        int64_t ld1, int64_t ld2) {
    GLASS_REQUIRE(a && pos && o, "segment_pool: null pointer");
    GLASS_REQUIRE(B > 0 && Smax > 0 && C > 0 && ld1 >= C && ld2 >= C && Smax < (1ll << 31), "segment_pool: bad sizes");
    if (mode < GLASS_POOL_SUM || mode > GLASS_POOL_SIZE) {
        set_error("segment_pool: unknown mode %d", mode);
        return GLASS_E_UNSUPPORTED;  // reference raises NotImplementedError (GLASSTest.py:168-171)
    }
    return 0;
}

extern "C" int glass_segment_pool_f32(const float* emb, int64_t lde, const int64_t* pos, int64_t B, int64_t Smax,
                                      int mode, float* out, int64_t ldo, int32_t* argmax, int64_t n_nodes, int64_t C,
                                      void* stream) {
    int rc = pool_args_ok(emb, pos, out, B, Smax, mode, C, lde, ldo);
    if (rc) return rc;
    GLASS_REQUIRE(mode != GLASS_POOL_MAX || argmax, "segment_pool: max pooling needs argmax");
    const bool vec = C % 4 == 0 && lde % 4 == 0 && ldo % 4 == 0 && aligned16(emb) && aligned16(out);
    const int cw = (int)ceil_div(C, vec ? 4 : 1);
    const int tc = pow2_ceil_cap(cw, kBlock);
    int tc_log2 = 0;
    while ((1 << tc_log2) < tc) ++tc_log2;
    dim3 grid((unsigned)B, (unsigned)ceil_div(cw, tc));
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(pool_fwd_kernel<4>, grid, dim3(kBlock), 0, st, emb, lde, pos, (int)Smax, mode, out, ldo,
                           argmax, n_nodes, (int)C, tc_log2);
    else
        hipLaunchKernelGGL(pool_fwd_kernel<1>, grid, dim3(kBlock), 0, st, emb, lde, pos, (int)Smax, mode, out, ldo,
                           argmax, n_nodes, (int)C, tc_log2);
    return launch_status("glass_segment_pool_f32");
}

static int pool_bwd_launch_atomic(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax, int mode,
                                  const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes, int64_t C, hipStream_t st) {
    const bool vec = C % 4 == 0 && ldd % 4 == 0 && aligned16(dout);
    const int cw = (int)ceil_div(C, vec ? 4 : 1);
    const int tc = pow2_ceil_cap(cw, kBlock);
    int tc_log2 = 0;
    while ((1 << tc_log2) < tc) ++tc_log2;
    dim3 grid((unsigned)B, (unsigned)ceil_div(cw, tc));
    if (vec)
        hipLaunchKernelGGL(pool_bwd_kernel<4>, grid, dim3(kBlock), 0, st, dout, ldd, pos, (int)Smax, mode, argmax, demb,
                           lde, n_nodes, (int)C, tc_log2);
    else
        hipLaunchKernelGGL(pool_bwd_kernel<1>, grid, dim3(kBlock), 0, st, dout, ldd, pos, (int)Smax, mode, argmax, demb,
                           lde, n_nodes, (int)C, tc_log2);
    return 0;
}

// The bitwise-repeatable entry: ordered, atomic-free while pos fits the LDS staging.  Beyond it (and for max pooling) it
// REFUSES with GLASS_E_WS instead of silently switching to float atomics: the caller picks glass_segment_pool_bwd_exact_f32 /
// glass_segment_pool_max_bwd_exact_f32 (workspace, exact) or, knowingly, glass_segment_pool_bwd_atomic_f32.
extern "C" int glass_segment_pool_bwd_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                          int mode, const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes,
                                          int64_t C, void* stream) {
    int rc = pool_args_ok(dout, pos, demb, B, Smax, mode, C, ldd, lde);
    if (rc) return rc;
    (void)argmax;
    if (mode == GLASS_POOL_MAX) {
        set_error("glass_segment_pool_bwd_f32: max pooling has no atomic-free form here — use glass_segment_pool_max_bwd_exact_f32 "
                  "(workspace) or glass_segment_pool_bwd_atomic_f32");
        return GLASS_E_WS;
    }
    if (B * Smax + B > kPoolOrderedMax) {
        set_error("glass_segment_pool_bwd_f32: B*Smax + B exceeds the ordered scatter's LDS staging — use "
                  "glass_segment_pool_bwd_exact_f32 (workspace) or glass_segment_pool_bwd_atomic_f32");
        return GLASS_E_WS;
    }
    hipStream_t st = (hipStream_t)stream;
    // atomic-free and bitwise repeatable (demb is zero-filled by the caller; untouched rows stay zero)
    hipLaunchKernelGGL(pool_bwd_ordered_kernel, dim3((unsigned)B), dim3(kBlock), sizeof(int32_t) * (size_t)(B * Smax + B), st,
                       dout, ldd, pos, (int)Smax, (int)B, mode, demb, lde, n_nodes, (int)C);
    return launch_status("glass_segment_pool_bwd_f32");
}

// The float-atomic scatter under its own name (any size, all four modes; demb zero-filled by the caller): one launch, results
// within rounding of the exact forms but ORDER-DEPENDENT once three or more entries share a node — not bitwise repeatable.
extern "C" int glass_segment_pool_bwd_atomic_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                                 int mode, const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes,
                                                 int64_t C, void* stream) {
    int rc = pool_args_ok(dout, pos, demb, B, Smax, mode, C, ldd, lde);
    if (rc) return rc;
    GLASS_REQUIRE(mode != GLASS_POOL_MAX || argmax, "segment_pool_bwd_atomic: max pooling needs argmax");
    pool_bwd_launch_atomic(dout, ldd, pos, B, Smax, mode, argmax, demb, lde, n_nodes, C, (hipStream_t)stream);
    return launch_status("glass_segment_pool_bwd_atomic_f32");
}


// ---- pair pool (Smax = 2, sum | mean | size) ----
static int pair_group_log2(int64_t C, bool vec) {
    const int cw = (int)ceil_div(C, vec ? 4 : 1);
    const int g = pow2_ceil_cap(cw, 64);
    int l = 0;
    while ((1 << l) < g) ++l;
    return l;
}

namespace glass {
int64_t bucket_ws_words(int64_t n_nodes, int64_t B, int64_t Smax, bool pair_form) {
    // off [n_nodes + 1, padded to 4] | rank [E] | list [E] | scale [B] (plain form)
    return ((n_nodes + 1 + 3) & ~(int64_t)3) + 2 * B * Smax + (pair_form ? 0 : B);
}

int bucket_build(const int64_t* pos, int64_t B, int64_t Smax, int mode, bool pair_form, int64_t n_nodes, void* ws, hipStream_t st,
                 BucketLists* out, bool dedup) {
    GLASS_REQUIRE(pos && ws && aligned16(ws) && B > 0 && Smax > 0 && n_nodes > 0 && B * Smax < (1ll << 30) && n_nodes < (1ll << 31) &&
                      (!pair_form || Smax == 2),
                  "bucket_build: 16-B aligned workspace, fewer than 2^30 entries and 2^31 nodes");
    const int64_t E = B * Smax;
    int32_t* off = (int32_t*)ws;
    int32_t* rank = off + ((n_nodes + 1 + 3) & ~(int64_t)3);
    int32_t* list = rank + E;
    float* scale = (pair_form || mode < 0) ? nullptr : reinterpret_cast<float*>(list + E);
    const unsigned ge = (unsigned)ceil_div(E, (int64_t)kBlock);
    hipLaunchKernelGGL(pair_zero_kernel, dim3((unsigned)ceil_div(n_nodes + 1, (int64_t)kBlock)), dim3(kBlock), 0, st, off, n_nodes + 1);
    if (scale)
        hipLaunchKernelGGL(pool_scale_kernel, dim3((unsigned)ceil_div(B, (int64_t)(kBlock / kWave))), dim3(kBlock), 0, st, pos, (int)B,
                           (int)Smax, mode, n_nodes, scale);
    hipLaunchKernelGGL(pair_rank_kernel, dim3(ge), dim3(kBlock), 0, st, pos, E, n_nodes, off, rank, dedup ? (int)Smax : 0);
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, st, off, n_nodes + 1);
    hipLaunchKernelGGL(pair_fill_kernel, dim3(ge), dim3(kBlock), 0, st, pos, E, (int)Smax, pair_form ? 1 : 0, n_nodes, off, rank, list);
    *out = BucketLists{off, list, scale};
    return 0;
}
}  // namespace glass

extern "C" int64_t glass_pair_pool_ws_bytes(int64_t n_nodes, int64_t B) {
    if (n_nodes <= 0 || B <= 0) return 0;
    return (int64_t)sizeof(int32_t) * bucket_ws_words(n_nodes, B, 2, true);
}

extern "C" int64_t glass_segment_pool_bwd_exact_ws_bytes(int64_t n_nodes, int64_t B, int64_t Smax) {
    if (n_nodes <= 0 || B <= 0 || Smax <= 0) return 0;
    return (int64_t)sizeof(int32_t) * bucket_ws_words(n_nodes, B, Smax, Smax == 2);
}

extern "C" int glass_pair_pool_f32(const float* emb, int64_t lde, const int64_t* pairs, int64_t B, int mode, float* out,
                                   int64_t ldo, int64_t n_nodes, int64_t C, void* stream) {
    int rc = pool_args_ok(emb, pairs, out, B, 2, mode, C, lde, ldo);
    if (rc) return rc;
    GLASS_REQUIRE(mode != GLASS_POOL_MAX, "pair_pool: sum | mean | size only");
    const bool vec = C % 4 == 0 && lde % 4 == 0 && ldo % 4 == 0 && aligned16(emb) && aligned16(out);
    const int gl = pair_group_log2(C, vec);
    const unsigned grid = (unsigned)ceil_div(B << gl, (int64_t)kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(pair_fwd_kernel<4>, dim3(grid), dim3(kBlock), 0, st, emb, lde, pairs, B, mode, out, ldo, n_nodes,
                           (int)C, gl);
    else
        hipLaunchKernelGGL(pair_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, st, emb, lde, pairs, B, mode, out, ldo, n_nodes,
                           (int)C, gl);
    return launch_status("glass_pair_pool_f32");
}

// sum | mean | size backward of a padded node matrix of any width, bucketed by node, exact sums (see pair_gather_kernel)
static int bucketed_pool_bwd(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax, int mode, float* demb,
                             int64_t lde, int64_t n_nodes, int64_t C, void* ws, void* stream, const char* what) {
    int rc = pool_args_ok(dout, pos, demb, B, Smax, mode, C, ldd, lde);
    if (rc) return rc;
    GLASS_REQUIRE(mode != GLASS_POOL_MAX && ws && aligned16(ws) && n_nodes > 0 && B * Smax < (1ll << 30) && n_nodes < (1ll << 31),
                  "pool backward (exact): sum | mean | size only, 16-B aligned workspace, fewer than 2^30 entries and 2^31 nodes");
    const bool vec = C % 4 == 0 && ldd % 4 == 0 && lde % 4 == 0 && aligned16(dout) && aligned16(demb);
    const int gl = pair_group_log2(C, vec);
    hipStream_t st = (hipStream_t)stream;
    BucketLists bl;
    rc = bucket_build(pos, B, Smax, mode, Smax == 2, n_nodes, ws, st, &bl);
    if (rc) return rc;
    const int32_t *off = bl.off, *list = bl.list;
    const float* scale = bl.scale;
    const unsigned gg = (unsigned)ceil_div(n_nodes, (int64_t)(kBlock >> gl));
    if (vec)
        hipLaunchKernelGGL(pair_gather_kernel<4>, dim3(gg), dim3(kBlock), 0, st, dout, ldd, mode, scale, off, list, demb, lde,
                           n_nodes, (int)C, gl);
    else
        hipLaunchKernelGGL(pair_gather_kernel<1>, dim3(gg), dim3(kBlock), 0, st, dout, ldd, mode, scale, off, list, demb, lde,
                           n_nodes, (int)C, gl);
    return launch_status(what);
}

extern "C" int glass_pair_pool_bwd_f32(const float* dout, int64_t ldd, const int64_t* pairs, int64_t B, int mode, float* demb,
                                       int64_t lde, int64_t n_nodes, int64_t C, void* ws, void* stream) {
    return bucketed_pool_bwd(dout, ldd, pairs, B, 2, mode, demb, lde, n_nodes, C, ws, stream, "glass_pair_pool_bwd_f32");
}

extern "C" int glass_segment_pool_bwd_exact_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                                int mode, float* demb, int64_t lde, int64_t n_nodes, int64_t C, void* ws,
                                                void* stream) {
    return bucketed_pool_bwd(dout, ldd, pos, B, Smax, mode, demb, lde, n_nodes, C, ws, stream, "glass_segment_pool_bwd_exact_f32");
}

extern "C" int glass_segment_pool_max_bwd_exact_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                                    const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes, int64_t C,
                                                    void* ws, void* stream) {
    int rc = pool_args_ok(dout, pos, demb, B, Smax, GLASS_POOL_MAX, C, ldd, lde);
    if (rc) return rc;
    GLASS_REQUIRE(argmax && ws && aligned16(ws), "segment_pool_max_bwd_exact: argmax and a 16-B aligned workspace required");
    const bool vec = C % 4 == 0 && ldd % 4 == 0 && lde % 4 == 0 && aligned16(dout) && aligned16(demb);
    const int gl = pair_group_log2(C, vec);
    hipStream_t st = (hipStream_t)stream;
    BucketLists bl;
    rc = bucket_build(pos, B, Smax, -1, false, n_nodes, ws, st, &bl, true);
    if (rc) return rc;
    const unsigned gg = (unsigned)ceil_div(n_nodes, (int64_t)(kBlock >> gl));
    if (vec)
        hipLaunchKernelGGL(pool_max_gather_kernel<4>, dim3(gg), dim3(kBlock), 0, st, dout, ldd, argmax, bl.off, bl.list, demb, lde,
                           n_nodes, (int)C, gl);
    else
        hipLaunchKernelGGL(pool_max_gather_kernel<1>, dim3(gg), dim3(kBlock), 0, st, dout, ldd, argmax, bl.off, bl.list, demb, lde,
                           n_nodes, (int)C, gl);
    return launch_status("glass_segment_pool_max_bwd_exact_f32");
}
