// K7: subgraph pooling straight from the padded node matrix.  Replaces pad2batch + emb[pos] +
// global_{add,mean,max}_pool / GraphSizeNorm (reference impl/models.py:346-350, 294-319;
// impl/utils.py:18-29).  One workgroup per subgraph: TC lanes x 16 B cover an embedding row,
// 256/TC row slots walk the padded node list; slots are combined through LDS in fixed order.
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

extern "C" int glass_segment_pool_bwd_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                          int mode, const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes,
                                          int64_t C, void* stream) {
    int rc = pool_args_ok(dout, pos, demb, B, Smax, mode, C, ldd, lde);
    if (rc) return rc;
    GLASS_REQUIRE(mode != GLASS_POOL_MAX || argmax, "segment_pool_bwd: max pooling needs argmax");
    const bool vec = C % 4 == 0 && ldd % 4 == 0 && aligned16(dout);
    const int cw = (int)ceil_div(C, vec ? 4 : 1);
    const int tc = pow2_ceil_cap(cw, kBlock);
    int tc_log2 = 0;
    while ((1 << tc_log2) < tc) ++tc_log2;
    dim3 grid((unsigned)B, (unsigned)ceil_div(cw, tc));
    hipStream_t st = (hipStream_t)stream;
    if (mode != GLASS_POOL_MAX && B * Smax + B <= kPoolOrderedMax) {
        // atomic-free and bitwise repeatable (demb is zero-filled by the caller; untouched rows stay zero)
        hipLaunchKernelGGL(pool_bwd_ordered_kernel, dim3((unsigned)B), dim3(kBlock), sizeof(int32_t) * (size_t)(B * Smax + B), st,
                           dout, ldd, pos, (int)Smax, (int)B, mode, demb, lde, n_nodes, (int)C);
        return launch_status("glass_segment_pool_bwd_f32");
    }
    if (vec)
        hipLaunchKernelGGL(pool_bwd_kernel<4>, grid, dim3(kBlock), 0, st, dout, ldd, pos, (int)Smax, mode, argmax, demb,
                           lde, n_nodes, (int)C, tc_log2);
    else
        hipLaunchKernelGGL(pool_bwd_kernel<1>, grid, dim3(kBlock), 0, st, dout, ldd, pos, (int)Smax, mode, argmax, demb,
                           lde, n_nodes, (int)C, tc_log2);
    return launch_status("glass_segment_pool_bwd_f32");
}
