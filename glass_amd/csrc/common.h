// Shared host/device helpers for libglass_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/glass_hip.h"

namespace glass {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

void set_error(const char* fmt, ...);

// Every launch function ends with this: report a launch failure without synchronising.
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define GLASS_REQUIRE(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            glass::set_error(__VA_ARGS__); \
            return GLASS_E_ARG;         \
        }                               \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Smallest power of two >= v (v >= 1), capped at `cap`.
inline int pow2_ceil_cap(int64_t v, int cap) {
    int p = 1;
    while (p < v && p < cap) p <<= 1;
    return p;
}

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float elu_f(float h) { return h > 0.f ? h : expm1f(h); }
__device__ __forceinline__ float elu_grad_f(float h) { return h > 0.f ? 1.f : __expf(h); }
// exp(h) - 1 with the hardware exponential: absolute error ~1e-7 (fine against the 1e-5 rel-inf bar), a
// handful of instructions instead of expm1f's ~50 — used where 32 ELUs per lane sit in a kernel epilogue.
__device__ __forceinline__ float elu_fast_f(float h) { return h > 0.f ? h : __expf(h) - 1.f; }
// Activation by code (wave-uniform): GLASS_ACT_NONE | GLASS_ACT_ELU (alpha = 1: GLASSTest.py:143) | GLASS_ACT_RELU (the
// reference's constructor default, impl/models.py:125,192, and the pre-training path's nn.ReLU, GNNEmb.py:90).
// act_fast: hardware exponential (the dense kernels' prologues / epilogues); act_exact: expm1f (graphnorm.hip's apply);
// act_grad: the derivative from the PRE-activation value (relu'(0) = 0, as torch's).
#ifdef GLASS_NO_RELU  // laboratory build: what the ELU-only code cost (A/B of the activation-code dispatch)
__device__ __forceinline__ float act_fast(int act, float h) { return act == GLASS_ACT_ELU ? elu_fast_f(h) : h; }
__device__ __forceinline__ float act_exact(int act, float h) { return act == GLASS_ACT_ELU ? elu_f(h) : h; }
__device__ __forceinline__ float act_grad(int act, float h) { return act == GLASS_ACT_ELU ? elu_grad_f(h) : 1.f; }
#else
// A three-way select per element (ELU ? .. : RELU ? .. : h) evaluated both activations and cost the benchmarked ELU step
// 3-4 us at ppi_bp-shape and 15 us at em_user-shape (A/B against an ELU-only build, same box).  Instead: NONE leaves through
// a wave-uniform branch as the ELU-only code did, and ELU / ReLU share ONE formula whose negative side is scaled by a
// uniform factor e (1: ELU, 0: ReLU) — the ELU path executes the instruction count it always had, bit-identical results
// (fmaf(1, exp(h), -1) rounds like exp(h) - 1), ReLU pays an exponential it does not need.
__device__ __forceinline__ float act_fast(int act, float h) {
    if (act == GLASS_ACT_NONE) return h;
    const float e = act == GLASS_ACT_ELU ? 1.f : 0.f;
    return h > 0.f ? h : fmaf(e, __expf(h), -e);
}
__device__ __forceinline__ float act_exact(int act, float h) {
    if (act == GLASS_ACT_NONE) return h;
    return h > 0.f ? h : (act == GLASS_ACT_ELU ? expm1f(h) : 0.f);
}
__device__ __forceinline__ float act_grad(int act, float h) {
    if (act == GLASS_ACT_NONE) return 1.f;
    const float e = act == GLASS_ACT_ELU ? 1.f : 0.f;
    return h > 0.f ? 1.f : e * __expf(h);
}
#endif
__host__ __device__ inline bool act_code_ok(int act) { return act == GLASS_ACT_NONE || act == GLASS_ACT_ELU || act == GLASS_ACT_RELU; }

// keep-scale of one element from its 32-bit word: 4 consecutive columns share one rand4() call.
__device__ __forceinline__ float keep_scale(uint32_t word, float p_drop, float inv_keep) {
    // uniform in [0,1) from the top 24 bits
    float u = (float)(word >> 8) * (1.0f / 16777216.0f);
    return u >= p_drop ? inv_keep : 0.f;
}

// Inverted-dropout configuration shared by every kernel that draws or re-draws a mask: the mask of element
// (row, col) is word col%4 of rand4(seed, step, call_id, row*cw4 + col/4), independent of vector width / ld.
struct Drop {
    float p, inv_keep;
    uint64_t seed, step, call_id;
    int cw4;  // ceil(C/4)
};

inline Drop make_drop(float p, uint64_t call_id, int64_t C) {
    Drop d;
    d.p = p;
    d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    d.seed = d.step = 0;
    d.call_id = call_id;
    d.cw4 = (int)ceil_div(C, 4);
    return d;
}

// 32-bit avalanche hash (Wellons' "lowbias32": 2 multiplies, bias 0.17 on the strict avalanche test)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// Four uniform words for counter idx under key (seed, step, call_id).  Counter-based (the backward regenerates the
// forward's mask from (seed, step, call_id, element) instead of storing it), like the Philox4x32-10 this replaced,
// but 9 integer multiplies per call instead of 40: v_mul_{lo,hi}_u32 are quarter-rate on CDNA and the mask
// generation sits on the critical path of latency-bound kernels (one wave per SIMD) — 0.406 -> 0.400 ms/step at C2.
// Statistical quality is far beyond what a dropout mask needs; not a cryptographic generator.
__device__ __forceinline__ void rand4(uint64_t seed, uint64_t step, uint64_t call_id, uint64_t idx, uint32_t (&out)[4]) {
    const uint32_t key = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) ^ mix32((uint32_t)step ^ mix32((uint32_t)call_id + 0x9E3779B9u))));
    const uint32_t x = mix32((uint32_t)idx ^ key) + (uint32_t)(idx >> 32) * 0x85EBCA6Bu;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = mix32(x + (uint32_t)(j + 1) * 0x632BE5ABu);
}

template <int VW>
__device__ __forceinline__ void drop_scales(const Drop& d, int64_t row, int c0, float (&s)[VW]) {
    uint32_t w[4];
    rand4(d.seed, d.step, d.call_id, (uint64_t)row * d.cw4 + (c0 >> 2), w);
    if (VW == 4) {
#pragma unroll
        for (int k = 0; k < VW; ++k) s[k] = keep_scale(w[k], d.p, d.inv_keep);
    } else {
        s[0] = keep_scale(w[c0 & 3], d.p, d.inv_keep);
    }
}

// ---- Adam (torch.optim.Adam, single-tensor formulation, amsgrad off) --------------------------------------------------
//   m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// Shared by the stand-alone optimizer launch (linear.hip) and the step's last launch, where it rides behind the
// embedding-table backward (embnorm.hip).
struct AdamCoef {
    float step_size, bc2_sqrt, w1, w2, beta2, eps, weight_decay;
};

// (the two double-precision pow() are evaluated by ONE thread per workgroup and broadcast through LDS: as per-thread code
// they were a visible share of a launch whose useful work is a few loads and a dozen flops per element)
__device__ __forceinline__ AdamCoef adam_coef(int64_t step_now, float lr, float beta1, float beta2, float eps, float weight_decay) {
    __shared__ float s_bc[2];
    if (threadIdx.x == 0) {
        const double t = (double)step_now;
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        s_bc[0] = (float)((double)lr / bc1);
        s_bc[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    AdamCoef c;
    c.step_size = s_bc[0];
    c.bc2_sqrt = s_bc[1];
    c.w1 = 1.f - beta1;
    c.w2 = 1.f - beta2;
    c.beta2 = beta2;
    c.eps = eps;
    c.weight_decay = weight_decay;
    return c;
}

// one element, on values: (pk, mk, vk) in -> updated in place
__device__ __forceinline__ void adam_element(const AdamCoef& c, float& pk, float gk, float& mk, float& vk) {
    if (c.weight_decay != 0.f) gk = fmaf(c.weight_decay, pk, gk);
    mk = mk + c.w1 * (gk - mk);
    vk = vk * c.beta2 + c.w2 * gk * gk;
    const float denom = sqrtf(vk) / c.bc2_sqrt + c.eps;
    pk = pk - c.step_size * (mk / denom);
}

__device__ __forceinline__ void adam_update(const AdamCoef& c, float* p, float gk, float* __restrict__ m,
                                            float* __restrict__ v, int64_t k) {
    float pk = p[k], mk = m[k], vk = v[k];
    adam_element(c, pk, gk, mk, vk);
    m[k] = mk;
    v[k] = vk;
    p[k] = pk;
}

// step_dev = int64[2]: (steps completed, ticket).  Every workgroup reads the count first and takes a ticket last; the
// workgroup that takes the last ticket publishes count + 1 and clears the ticket.
__device__ __forceinline__ void adam_ticket(int64_t* __restrict__ step_dev, int64_t step_now) {
    __syncthreads();  // every thread of this workgroup has read step_dev[0]
    if (threadIdx.x == 0) {
        const unsigned long long taken = atomicAdd(reinterpret_cast<unsigned long long*>(step_dev + 1), 1ull) + 1ull;
        if (taken == gridDim.x) {
            step_dev[1] = 0;
            step_dev[0] = step_now;
        }
    }
}

// ---- LDS-only barrier and buffer addressing (the staged dense kernels) -----------------------------------------------
// lds_barrier: s_waitcnt lgkmcnt(0) + s_barrier — what a stage hand-over through LDS needs.  __syncthreads() also carries a
// release fence that waits for vmcnt(0), i.e. it DRAINS the wave's outstanding global loads: a prefetch across it is no
// prefetch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Buffer addressing for the stage loop: an out-of-range offset makes a load return 0 and drops a store, so row validity
// costs no branch — and the loop holds no CONDITIONAL memory instruction.  That matters for more than the branch: vmcnt
// counts loads and stores in issue order, and when a younger memory instruction may or may not have been issued the
// compiler has to wait with vmcnt(0) for an older load — i.e. for every store in flight (the first version of this kernel
// stalled ~1 us per two stages on its own output stores; ISA: `global_store_dwordx4; s_waitcnt vmcnt(0); ds_write_b128`).
using buf_rsrc = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x4 __attribute__((__vector_size__(16)));  // (the builtin's own type; an ext_vector_type took one dword and splat it)
constexpr int kBufOOB = 0x7fffffff;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(buf_rsrc r, int off) {
    // (whole-vector copy: __builtin_bit_cast on the ELEMENTS of the vector compiled to one dword load splat four times)
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    float4 f;
    __builtin_memcpy(&f, &v, sizeof(f));
    return f;
}
// glass_pin(x): every later use of x is ordered behind this point (an empty volatile asm that "rewrites" the register), so
// the compiler cannot hoist a use — and with it the s_waitcnt for the load that produced x — above the loads issued before.
#define glass_pin(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ int buf_load1i(buf_rsrc r, int off) { return (int)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ float buf_load1f(buf_rsrc r, int off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store4(buf_rsrc r, int off, const float4& f) {
    u32x4 v;
    __builtin_memcpy(&v, &f, sizeof(v));
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}
__device__ __forceinline__ void buf_store1(buf_rsrc r, int off, float f) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), r, off, 0, 0);
}


}  // namespace glass
