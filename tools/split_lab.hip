// lab: accuracy of fp32 products formed from 3-way bf16 splits on the bf16 matrix cores (6 partial products) against the fp32
// matrix-core product and an fp64 host product.  C[32 x 32] tiles, A [M][K], B [N][K] (both row-major, K contiguous).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
    float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, bf16x2));
    ra -= __builtin_bit_cast(float, mid << 16);
    rb -= __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, bf16x2));
}
__device__ __forceinline__ void split8(const float* p, bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh, um, ul;
    for (int i = 0; i < 4; ++i) { unsigned a, b, c; split2(p[2 * i], p[2 * i + 1], a, b, c); uh[i] = a; um[i] = b; ul[i] = c; }
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}
// mode 0: fp32 matrix core; 1: six products (small first); 2: six products (large first); 3: three products (hh, hm, mh)
__global__ void prod(const float* A, const float* B, float* C, int M, int N, int K, int mode) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int tm = blockIdx.x, tn = blockIdx.y;
    f32x16 acc = {0};
    const float* a = A + (size_t)(tm * 32 + j) * K;
    const float* b = B + (size_t)(tn * 32 + j) * K;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + h], b[k + h], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            split8(a + k + 8 * h, ah, am, al);
            split8(b + k + 8 * h, bh, bm, bl);
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            } else if (mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            }
        }
    }
    for (int i = 0; i < 16; ++i) C[(size_t)(tm * 32 + 8 * (i >> 2) + 4 * h + (i & 3)) * N + tn * 32 + j] = acc[i];
}
static double rnd() { return (double)rand() / RAND_MAX; }
static double gauss() { return sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd()); }
int main() {
    const int M = 256, N = 256;
    for (int dist = 0; dist < 3; ++dist)
        for (int K : {64, 512, 4096}) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            for (auto& v : A) v = (float)(dist == 0 ? gauss() : dist == 1 ? fabs(gauss()) + 0.5 : gauss() * exp(4.0 * gauss()));
            for (auto& v : B) v = (float)(dist == 0 ? gauss() : dist == 1 ? fabs(gauss()) + 0.5 : gauss() * exp(4.0 * gauss()));
            std::vector<double> R((size_t)M * N), S((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double s = 0, t = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)m * K + k] * B[(size_t)n * K + k]; s += p; t += fabs(p); }
                    R[(size_t)m * N + n] = s; S[(size_t)m * N + n] = t;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            std::vector<float> C((size_t)M * N);
            printf("dist %d K %5d:", dist, K);
            for (int mode = 0; mode < 4; ++mode) {
                hipLaunchKernelGGL(prod, dim3(M / 32, N / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K, mode);
                hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                double worst = 0, rms = 0;  // error relative to sum |a b| (the bound fp32 accumulation itself is stated in)
                for (size_t i = 0; i < C.size(); ++i) { const double e = fabs((double)C[i] - R[i]) / S[i]; worst = fmax(worst, e); rms += e * e; }
                printf("  mode %d max %.3e rms %.3e", mode, worst, sqrt(rms / C.size()));
            }
            printf("\n");
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    return 0;
}
