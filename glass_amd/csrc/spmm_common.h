// Pieces of K1 (spmm.hip) shared with the launches its long-row items can ride in (linear.hip: the step's batched
// weight-gradient reduction carries the selection product of the embedding backward).
#pragma once
#include "common.h"

namespace glass {

constexpr int32_t kPlanMagic = 0x474C5350;  // 'GLSP'
constexpr int32_t kPlanVersion = 2;

// Cache policy of the once-read streams (rowptr / col / val in, Y out), template parameter NT of the kernels: non-temporal
// when the launch streams more than the Infinity Cache can hold (they would only displace X rows, the one operand that
// is re-read: permutation N = 4 M 413 -> 395 us, uniform degree 3 at N = 2 M 311 -> 292 us), default policy otherwise
// — on the BASELINE graphs col/val/Y of one launch ARE re-read by the next launch from L2 / Infinity Cache, and
// marking them non-temporal costs 13.4 -> 15.7 us at ppi_bp-shape and 75 -> 89 us at hpo_neuro-shape.
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
typedef float k1_f32x4 __attribute__((ext_vector_type(4)));
// header word indices
enum { H_MAGIC, H_VER, H_NROWS, H_NNZ, H_NSWEEP, H_NLONG, H_NREDUCE, H_NSLOTS, H_LONG_THR, H_LONG_CHUNK,
       H_OFF_SWEEP, H_OFF_LONG, H_OFF_REDUCE, H_RP_FACTOR, H_MIN_ITEM_DEG, H_FLAT_SHARE };

// ---- vector helpers --------------------------------------------------------------------------
template <int VW> struct Vec;
template <> struct Vec<4> {
    float4 v;
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
    template <bool NT>
    __device__ __forceinline__ void store_out(float* p) const {  // a row of Y: written once, not read by this kernel
        if (NT) __builtin_nontemporal_store((k1_f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<k1_f32x4*>(p));
        else store(p);
    }
    __device__ __forceinline__ void fma(float a, const Vec& x) {
        v.x = fmaf(a, x.v.x, v.x); v.y = fmaf(a, x.v.y, v.y); v.z = fmaf(a, x.v.z, v.z); v.w = fmaf(a, x.v.w, v.w);
    }
    __device__ __forceinline__ void add(const Vec& x) { v.x += x.v.x; v.y += x.v.y; v.z += x.v.z; v.w += x.v.w; }
    __device__ __forceinline__ void xor_add(int s) {
        v.x += __shfl_xor(v.x, s); v.y += __shfl_xor(v.y, s); v.z += __shfl_xor(v.z, s); v.w += __shfl_xor(v.w, s);
    }
};
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
    template <bool NT>
    __device__ __forceinline__ void store_out(float* p) const {
        if (NT) __builtin_nontemporal_store(v, p);
        else store(p);
    }
    __device__ __forceinline__ void fma(float a, const Vec& x) { v = fmaf(a, x.v, v); }
    __device__ __forceinline__ void add(const Vec& x) { v += x.v; }
    __device__ __forceinline__ void xor_add(int s) { v += __shfl_xor(v, s); }
};

// Accumulate edges [e0,e1) of one row into `acc` (per lane-group partial sums).
// All 64 lanes execute this together; e0/e1 are wave-uniform.
template <int VW, int LPR, int U, bool NT>
__device__ __forceinline__ void gather_edges(Vec<VW>& acc, const int32_t* __restrict__ col,
                                             const float* __restrict__ val, const float* __restrict__ Xc, int64_t ldx,
                                             int e0, int e1, int lane, int grp, bool col_ok) {
    constexpr int G = kWave / LPR;
    for (int eb = e0; eb < e1; eb += kWave) {
        const int cnt = min(kWave, e1 - eb);
        int my_c = 0;
        float my_v = 0.f;
        if (lane < cnt) {
            my_c = ld_stream<NT>(col + eb + lane);
            my_v = ld_stream<NT>(val + eb + lane);
        }
        for (int j = 0; j < cnt; j += U * G) {
            Vec<VW> x[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * G + grp;          // < 64 + U*G: shfl wraps mod 64, masked by `ok`
                const int c = __shfl(my_c, idx);
                v[u] = __shfl(my_v, idx);
                const bool ok = col_ok && idx < cnt;
                x[u].zero();
                if (ok) x[u].load(Xc + (int64_t)c * ldx);
                if (!ok) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc.fma(v[u], x[u]);
        }
    }
}

template <int VW, int LPR>
__device__ __forceinline__ void reduce_groups(Vec<VW>& acc) {
#pragma unroll
    for (int s = LPR; s < kWave; s <<= 1) acc.xor_add(s);
}

// ---- long rows: one workgroup per (row, chunk); 4 waves combine through LDS --------------------
template <int VW, int LPR, int U, bool NT>
__device__ __forceinline__ void long_item_body(const int32_t* __restrict__ col, const float* __restrict__ val,
                                               const float* __restrict__ X, int64_t ldx, float* __restrict__ Y,
                                               int64_t ldy, float* __restrict__ partials, int H,
                                               const int32_t* __restrict__ it, float* lds) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    const int grp = lane / LPR, sub = lane % LPR;
    const int coff = (blockIdx.y * LPR + sub) * VW;
    const bool col_ok = coff < H;
    const int row = it[0], eb = it[1], ee = it[2], slot = it[3];
    // wave w takes the w-th quarter of the chunk, rounded to whole 64-edge batches
    const int per = ((ee - eb + 4 * kWave - 1) / (4 * kWave)) * kWave;
    const int e0 = min(eb + w * per, ee), e1 = min(e0 + per, ee);
    Vec<VW> acc;
    acc.zero();
    gather_edges<VW, LPR, U, NT>(acc, col, val, X + coff, ldx, e0, e1, lane, grp, col_ok);
    reduce_groups<VW, LPR>(acc);
    if (grp == 0) acc.store(&lds[(w * LPR + sub) * VW]);
    __syncthreads();
    if (w == 0 && grp == 0 && col_ok) {
        Vec<VW> s, t;
        s.load(&lds[sub * VW]);
#pragma unroll
        for (int k = 1; k < kBlock / kWave; ++k) {
            t.load(&lds[(k * LPR + sub) * VW]);
            s.add(t);
        }
        float* dst = slot < 0 ? Y + (int64_t)row * ldy : partials + (int64_t)slot * H;
        s.store(dst + coff);
    }
}

}  // namespace glass
