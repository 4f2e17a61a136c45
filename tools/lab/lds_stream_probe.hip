// Feasibility probe (laboratory): how fast can every workgroup of a launch stream the SAME 3.7 MB table through its LDS?
// 228 workgroups x 1024 threads, 64 KB blocks, double buffered, global_load_lds (DMA); optional LDS gather per block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr int kThreads = 1024, kBC = 256, kH = 64;  // block = 256 rows x 64 floats = 64 KB

template <int GATHERS>
__global__ __launch_bounds__(kThreads) void stream_kernel(const float* __restrict__ X, int n_blocks, float* __restrict__ out,
                                                          const int* __restrict__ idx) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x;
    const int wbase = __builtin_amdgcn_readfirstlane((tid >> 6) * 64);
    auto dma = [&](int b, int buf) {
        const float4* src = reinterpret_cast<const float4*>(X) + (size_t)b * (kBC * kH / 4);
        float4* dst = lds + buf * (kBC * kH / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(src + i * kThreads + tid, (lds_void_ptr)(uintptr_t)(dst + i * kThreads + wbase), 16, 0, 0);
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    dma(0, 0);
    for (int b = 0; b < n_blocks; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (b + 1 < n_blocks) dma(b + 1, (b + 1) & 1);
        const float4* cur = lds + (b & 1) * (kBC * kH / 4);
        const int grp = tid >> 4, sub = tid & 15;
#pragma unroll
        for (int g = 0; g < GATHERS; ++g) {
            const int r = idx[(b * GATHERS + g) * 64 + grp] & (kBC - 1);
            const float4 x = cur[r * 16 + sub];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
    }
    if (acc.x == 12345.f) out[blockIdx.x * kThreads + tid] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
    const int N = 14587, n_blocks = (N + kBC - 1) / kBC, wgs = argc > 1 ? atoi(argv[1]) : 228;
    float* X; float* out; int* idx;
    HIP_OK(hipMalloc(&X, (size_t)n_blocks * kBC * kH * 4));
    HIP_OK(hipMemset(X, 0, (size_t)n_blocks * kBC * kH * 4));
    HIP_OK(hipMalloc(&out, (size_t)wgs * kThreads * 4));
    HIP_OK(hipMalloc(&idx, (size_t)n_blocks * 16 * 64 * 4));
    HIP_OK(hipMemset(idx, 0, (size_t)n_blocks * 16 * 64 * 4));
    hipStream_t st; HIP_OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const size_t lds = 2 * kBC * kH * 4;
    auto run = [&](auto kern, const char* name) {
        HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), lds, st, X, n_blocks, out, idx);
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), lds, st, X, n_blocks, out, idx);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %d workgroups, %d blocks of 64 KB each: %.1f us per launch (%.1f TB/s into LDS)\n", name, wgs, n_blocks,
               ms * 1e3 / 50, (double)wgs * n_blocks * 65536 / (ms / 50 * 1e-3) / 1e12);
    };
    run(stream_kernel<0>, "stream only");
    run(stream_kernel<8>, "stream + 8 gathers/group/block (444 edges per 64 rows per block ~ 7)");
    run(stream_kernel<16>, "stream + 16 gathers/group/block");
    return 0;
}
