// Prediction head + loss of the training step in two launches: logits = pooled @ W^T + b (the bare
// nn.Linear head of reference GLASSTest.py:159-160), then CrossEntropyLoss (multi-class,
// GLASSTest.py:69) or BCEWithLogitsLoss on the flattened logits (binary / multi-label,
// GLASSTest.py:57-58), both with mean reduction; and their whole backward.
// The operands are tiny (B ~ 80 subgraphs, C = H*L = 128 features, K <= ~50 classes) but as separate
// library calls they are ~12 launches of ~4 us each (GEMM, softmax, nll, fills, three backward GEMMs,
// bias reduction, gradient accumulations): pure launch latency on this path.
#include "common.h"

namespace glass {

constexpr int GLASS_LOSS_CE = 0;
constexpr int GLASS_LOSS_BCE = 1;

constexpr int kMaxK = 256;  // classes handled by the per-subgraph workgroup

// One workgroup per subgraph b: its pooled row is staged in LDS, each wave computes logits[b,k] for
// k = wave, wave+4, ... as a 64-lane dot product (one round trip to memory), then lane 0 of wave 0 does the
// K-term softmax / sigmoid and the row's loss term.  A second one-workgroup launch averages the B terms
// in index order (deterministic; a float atomic would not be).
__global__ __launch_bounds__(kBlock) void head_logits_kernel(const float* __restrict__ pooled, int64_t ldp,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             const void* __restrict__ target, int mode, int C, int K,
                                                             float* __restrict__ logits, float* __restrict__ prob,
                                                             float* __restrict__ loss_rows) {
    __shared__ float zs[kMaxK];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = pooled + (int64_t)b * ldp;
    for (int k = w; k < K; k += kBlock / kWave) {
        const float* wr = W + (int64_t)k * C;
        float s = 0.f;
        for (int c = lane; c < C; c += kWave) s = fmaf(p[c], wr[c], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) zs[k] = s + bias[k];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float term = 0.f;
    if (mode == GLASS_LOSS_CE) {
        float m = zs[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, zs[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(zs[k] - m);
        const float lse = m + logf(se);
        for (int k = 0; k < K; ++k) {
            logits[(int64_t)b * K + k] = zs[k];
            prob[(int64_t)b * K + k] = expf(zs[k] - lse);
        }
        const int64_t t = ((const int64_t*)target)[b];
        term = (t >= 0 && t < K) ? lse - zs[t] : 0.f;
    } else {
        const float* y = (const float*)target + (int64_t)b * K;
        for (int k = 0; k < K; ++k) {
            const float z = zs[k];
            logits[(int64_t)b * K + k] = z;
            prob[(int64_t)b * K + k] = 1.f / (1.f + expf(-z));
            // max(z,0) - z*y + log(1 + exp(-|z|))   (torch's stable BCE-with-logits)
            term += fmaxf(z, 0.f) - z * y[k] + log1pf(expf(-fabsf(z)));
        }
    }
    loss_rows[b] = term;
}

// The head alone (evaluation: no target, no loss): logits[b, k] = pooled[b, :] . W[k, :] + bias[k], one wave per (b, k) pair
// strided over the workgroup — any K.
__global__ __launch_bounds__(kBlock) void head_linear_kernel(const float* __restrict__ pooled, int64_t ldp,
                                                             const float* __restrict__ W, const float* __restrict__ bias, int C,
                                                             int K, float* __restrict__ logits, int64_t ldl) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = pooled + (int64_t)b * ldp;
    for (int k = w; k < K; k += kBlock / kWave) {
        const float* wr = W + (int64_t)k * C;
        float s = 0.f;
        for (int c = lane; c < C; c += kWave) s = fmaf(p[c], wr[c], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) logits[(int64_t)b * ldl + k] = s + (bias ? bias[k] : 0.f);
    }
}

__global__ __launch_bounds__(kBlock) void head_loss_mean_kernel(const float* __restrict__ loss_rows, int B, float denom,
                                                                float* __restrict__ loss) {
    __shared__ double red[kBlock];
    double part = 0.0;
    for (int b = threadIdx.x; b < B; b += kBlock) part += (double)loss_rows[b];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)denom);
}

__device__ __forceinline__ float dlogit(const float* prob, const void* target, int mode, int b, int k, int K, float scale) {
    const float p = prob[(int64_t)b * K + k];
    if (mode == GLASS_LOSS_CE) return scale * (p - (((const int64_t*)target)[b] == k ? 1.f : 0.f));
    return scale * (p - ((const float*)target)[(int64_t)b * K + k]);
}

// Workgroups 0..B-1: dpooled[b,:] = dlogits[b,:] @ W.   Workgroups B..B+K-1: dW[k,:] (+)= sum_b dlogits[b,k] *
// pooled[b,:], db[k] (+)= sum_b dlogits[b,k] (fixed b order).  dlogits = gl * (prob - target) / (B or B*K).
// The dlogits a workgroup needs are staged in LDS; the inner loops keep 8 independent loads in flight.
__global__ __launch_bounds__(kBlock) void head_loss_bwd_kernel(const float* __restrict__ pooled, int64_t ldp,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ prob,
                                                               const void* __restrict__ target, int mode,
                                                               const float* __restrict__ gl, int B, int C, int K,
                                                               float* __restrict__ dpooled, int64_t lddp,
                                                               float* __restrict__ dW, float* __restrict__ db,
                                                               int accumulate) {
    extern __shared__ float dl[];  // max(B, K) floats
    const float scale = gl[0] / (mode == GLASS_LOSS_CE ? (float)B : (float)B * (float)K);
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < B) {
        for (int k = tid; k < K; k += kBlock) dl[k] = dlogit(prob, target, mode, blk, k, K, scale);
        __syncthreads();
        for (int c = tid; c < C; c += kBlock) {
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < K; ++k) s = fmaf(dl[k], W[(int64_t)k * C + c], s);
            dpooled[(int64_t)blk * lddp + c] = s;
        }
        return;
    }
    const int k = blk - B;
    for (int b = tid; b < B; b += kBlock) dl[b] = dlogit(prob, target, mode, b, k, K, scale);
    __syncthreads();
    for (int c = tid; c < C; c += kBlock) {
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s = fmaf(dl[b], pooled[(int64_t)b * ldp + c], s);
        float* d = dW + (int64_t)k * C + c;
        *d = accumulate ? *d + s : s;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dl[b];
        db[k] = accumulate ? db[k] + s : s;
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_head_loss_fwd_f32(const float* pooled, int64_t ldp, const float* W, const float* bias,
                                       const void* target, int mode, int64_t B, int64_t C, int64_t K, float* logits,
                                       float* prob, float* loss, void* stream) {
    GLASS_REQUIRE(pooled && W && bias && target && logits && prob && loss, "head_loss_fwd: null pointer");
    GLASS_REQUIRE(B > 0 && C > 0 && K > 0 && ldp >= C && B * K < (1ll << 30), "head_loss_fwd: bad sizes");
    if (mode != GLASS_LOSS_CE && mode != GLASS_LOSS_BCE) {
        set_error("head_loss_fwd: unknown loss mode %d", mode);
        return GLASS_E_UNSUPPORTED;
    }
    if (K > kMaxK) {
        set_error("head_loss_fwd: at most %d classes", kMaxK);
        return GLASS_E_UNSUPPORTED;
    }
    // loss_rows lives in the tail of `prob`'s sibling buffer: the caller passes prob with B*K + B floats
    float* loss_rows = prob + B * K;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head_logits_kernel, dim3((unsigned)B), dim3(kBlock), 0, st, pooled, ldp, W, bias, target, mode, (int)C,
                       (int)K, logits, prob, loss_rows);
    hipLaunchKernelGGL(head_loss_mean_kernel, dim3(1), dim3(kBlock), 0, st, loss_rows, (int)B,
                       mode == GLASS_LOSS_CE ? (float)B : (float)B * (float)K, loss);
    return launch_status("glass_head_loss_fwd_f32");
}

extern "C" int glass_head_loss_bwd_f32(const float* pooled, int64_t ldp, const float* W, const float* prob,
                                       const void* target, int mode, const float* grad_loss, int64_t B, int64_t C,
                                       int64_t K, float* dpooled, int64_t lddp, float* dW, float* db, int accumulate,
                                       void* stream) {
    GLASS_REQUIRE(pooled && W && prob && target && grad_loss && dpooled && dW && db, "head_loss_bwd: null pointer");
    GLASS_REQUIRE(B > 0 && C > 0 && K > 0 && ldp >= C && lddp >= C, "head_loss_bwd: bad sizes");
    if (mode != GLASS_LOSS_CE && mode != GLASS_LOSS_BCE) {
        set_error("head_loss_bwd: unknown loss mode %d", mode);
        return GLASS_E_UNSUPPORTED;
    }
    const size_t lds = sizeof(float) * (size_t)(B > K ? B : K);
    GLASS_REQUIRE(lds <= 64 * 1024, "head_loss_bwd: batch too large for the LDS staging");
    hipLaunchKernelGGL(head_loss_bwd_kernel, dim3((unsigned)(B + K)), dim3(kBlock), lds, (hipStream_t)stream, pooled, ldp, W,
                       prob, target, mode, grad_loss, (int)B, (int)C, (int)K, dpooled, lddp, dW, db, accumulate);
    return launch_status("glass_head_loss_bwd_f32");
}

extern "C" int glass_head_linear_f32(const float* pooled, int64_t ldp, const float* W, const float* bias, int64_t B, int64_t C,
                                     int64_t K, float* logits, int64_t ldl, void* stream) {
    GLASS_REQUIRE(pooled && W && logits && B > 0 && C > 0 && K > 0 && ldp >= C && ldl >= K && B < (1ll << 31), "head_linear: bad arguments");
    hipLaunchKernelGGL(head_linear_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, pooled, ldp, W, bias, (int)C,
                       (int)K, logits, ldl);
    return launch_status("glass_head_linear_f32");
}
