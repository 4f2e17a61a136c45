// Laboratory probe (round 6): what does a DEVICE-SCOPE BARRIER inside one launch cost on MI355X, against
// the kernel boundary it would replace?  The step at hidden 64 is 22 launches of ~11 us; between a producer of per-column
// sums (exact 64-bit integer accumulators, gn_acc.h) and their consumer sits either a launch boundary (today: a statistics
// launch, then the consumer folds the accumulators in its prologue) or — if the consumer's workgroups all fit the chip at
// once — a barrier inside the consumer: every workgroup reads ITS rows once, adds its partial sums, arrives, waits, folds.
//
// Modes (all inside one captured hipGraph of `chain` nodes, replayed; time per node):
//   empty      : kernels that do nothing (the launch floor of a chain)
//   two        : stats kernel (reads rows, 128 column sums -> 2 x 64-bit atomics each, 16 replicas) + consumer kernel (reads the
//                same rows again, folds 16 replicas x 2 KB, writes rows)          = today's two launches
//   fused      : ONE kernel: rows -> registers, sums -> atomics, arrive / wait, fold, write rows
//   barrier K  : one kernel with K bare barriers back to back (arrive + spin), nothing else: the cost of one barrier
// Only the accumulators and the arrival counter cross workgroups; both are touched by device-scope atomics / sc1 loads that
// bypass the per-XCD L2, so no cache write-back or invalidate is needed around the barrier (a release / acquire at agent
// scope would write back and invalidate the whole L2 of the XCD: DESIGN.md K1, "in-launch reduction").
// The spin is BOUNDED: a workgroup that waits longer than ~20 ms sets an error word and goes on — never a hang.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kThreads = 256, kH = 64, kRep = 16, kRowsWg = 80;

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_sc1(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// arrive + wait on word `ctr` (zeroed before the launch); returns false on timeout
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned n_wg, unsigned* err) {
    __builtin_amdgcn_s_waitcnt(0);  // this wave's atomics have been acknowledged (device-scope atomics return after the memory-side unit ran them)
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = wall_clock64();
        while (ld_sc1(ctr) < n_wg) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 2000000) {  // 100 MHz clock: 20 ms
                ok = false;
                *err = 1;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(kThreads) void empty_kernel(int) {}

__global__ __launch_bounds__(kThreads) void barrier_kernel(unsigned* ctrs, int K, unsigned* err) {
    for (int k = 0; k < K; ++k) grid_barrier(ctrs + 32 * k, gridDim.x, err);
}

// thread (rs, ga): rows rs + 16 st of the workgroup's tile, columns 4 ga .. + 3
__device__ __forceinline__ void load_rows(const float* X, int64_t N, float4 (&v)[kRowsWg / 16]) {
    const int rs = threadIdx.x >> 4, ga = threadIdx.x & 15;
#pragma unroll
    for (int st = 0; st < kRowsWg / 16; ++st) {
        const int64_t r = (int64_t)blockIdx.x * kRowsWg + 16 * st + rs;
        v[st] = r < N ? reinterpret_cast<const float4*>(X + r * kH)[ga] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ void add_sums(const float4 (&v)[kRowsWg / 16], long long* acc, double (*red)[2 * kH]) {
    const int rs = threadIdx.x >> 4, ga = threadIdx.x & 15, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < kRowsWg / 16; ++st) {
        const float a[4] = {v[st].x, v[st].y, v[st].z, v[st].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += a[k]; q[k] += (double)a[k] * a[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s[k] += __shfl_xor(s[k], 16); q[k] += __shfl_xor(q[k], 16);
        s[k] += __shfl_xor(s[k], 32); q[k] += __shfl_xor(q[k], 32);
    }
    (void)rs;
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[w][4 * ga + k] = s[k]; red[w][kH + 4 * ga + k] = q[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kH) {
        const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const double sv = t * 4096.0, fl = floor(sv);
        unsigned long long* p = reinterpret_cast<unsigned long long*>(acc + (((size_t)(blockIdx.x % kRep) * 2 * kH + threadIdx.x) * 2));
        atomicAdd(p, (unsigned long long)(long long)fl);
        atomicAdd(p + 1, (unsigned long long)(long long)((sv - fl) * 1099511627776.0));
    }
}

template <bool SC1>
__device__ __forceinline__ void fold(const long long* acc, float* coef_s) {
    if (threadIdx.x < 2 * kH) {
        long long hi = 0, lo = 0;
        long long h[kRep], l[kRep];
#pragma unroll
        for (int r = 0; r < kRep; ++r) {
            const long long* p = acc + (((size_t)r * 2 * kH + threadIdx.x) * 2);
            h[r] = SC1 ? ld_sc1(p) : p[0];
            l[r] = SC1 ? ld_sc1(p + 1) : p[1];
        }
#pragma unroll
        for (int r = 0; r < kRep; ++r) { hi += h[r]; lo += l[r]; }
        coef_s[threadIdx.x] = (float)(((double)hi + (double)lo * (1.0 / 1099511627776.0)) * (1.0 / 4096.0));
    }
    __syncthreads();
}

__device__ __forceinline__ void write_rows(float* Y, int64_t N, const float4 (&v)[kRowsWg / 16], const float* coef_s) {
    const int rs = threadIdx.x >> 4, ga = threadIdx.x & 15;
    const float4 c = *reinterpret_cast<const float4*>(coef_s + 4 * ga), d = *reinterpret_cast<const float4*>(coef_s + kH + 4 * ga);
#pragma unroll
    for (int st = 0; st < kRowsWg / 16; ++st) {
        const int64_t r = (int64_t)blockIdx.x * kRowsWg + 16 * st + rs;
        if (r < N) reinterpret_cast<float4*>(Y + r * kH)[ga] = make_float4(v[st].x * c.x + d.x, v[st].y * c.y + d.y, v[st].z * c.z + d.z, v[st].w * c.w + d.w);
    }
}

__global__ __launch_bounds__(kThreads) void stats_kernel(const float* X, int64_t N, long long* acc) {
    __shared__ double red[4][2 * kH];
    float4 v[kRowsWg / 16];
    load_rows(X, N, v);
    add_sums(v, acc, red);
}
__global__ __launch_bounds__(kThreads) void consumer_kernel(const float* X, float* Y, int64_t N, const long long* acc) {
    __shared__ float coef_s[2 * kH];
    float4 v[kRowsWg / 16];
    load_rows(X, N, v);
    fold<false>(acc, coef_s);
    write_rows(Y, N, v, coef_s);
}
__global__ __launch_bounds__(kThreads) void fused_kernel(const float* X, float* Y, int64_t N, long long* acc, unsigned* ctr, unsigned* err) {
    __shared__ double red[4][2 * kH];
    __shared__ float coef_s[2 * kH];
    float4 v[kRowsWg / 16];
    load_rows(X, N, v);
    add_sums(v, acc, red);
    grid_barrier(ctr, gridDim.x, err);
    fold<true>(acc, coef_s);
    write_rows(Y, N, v, coef_s);
}
__global__ void zero_kernel(long long* acc, int n_acc, unsigned* ctrs, int n_ctr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_acc; i += gridDim.x * blockDim.x) acc[i] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ctr; i += gridDim.x * blockDim.x) ctrs[i] = 0;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 17080;
    const int chain = argc > 2 ? atoi(argv[2]) : 16, reps = 200;
    const int wgs = (int)((N + kRowsWg - 1) / kRowsWg);
    float *X, *Y;
    long long* acc;
    unsigned *ctrs, *err;
    const int n_acc = chain * kRep * 2 * kH * 2, n_ctr = 32 * 64 * chain;
    HIP_OK(hipMalloc(&X, N * kH * 4)); HIP_OK(hipMalloc(&Y, N * kH * 4));
    std::vector<float> hx(N * kH);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    HIP_OK(hipMemcpy(X, hx.data(), N * kH * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&acc, (size_t)n_acc * 8)); HIP_OK(hipMalloc(&ctrs, (size_t)n_ctr * 4)); HIP_OK(hipMalloc(&err, 4));
    HIP_OK(hipMemset(err, 0, 4));
    hipStream_t st; HIP_OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    printf("N = %lld rows x %d floats, %d workgroups of %d threads, chain of %d per graph, %d replays\n", (long long)N, kH, wgs, kThreads, chain, reps);
    auto timed = [&](const char* name, auto body, int per_chain_units) {
        hipGraph_t g; hipGraphExec_t ge;
        HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(zero_kernel, dim3(64), dim3(256), 0, st, acc, n_acc, ctrs, n_ctr);
        for (int i = 0; i < chain; ++i) body(i);
        HIP_OK(hipStreamEndCapture(st, &g));
        HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 10; ++i) HIP_OK(hipGraphLaunch(ge, st));
        HIP_OK(hipStreamSynchronize(st));
        float best = 1e30f, sum = 0;
        for (int b = 0; b < 5; ++b) {
            HIP_OK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) HIP_OK(hipGraphLaunch(ge, st));
            HIP_OK(hipEventRecord(e1, st));
            HIP_OK(hipStreamSynchronize(st));
            float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        unsigned herr; HIP_OK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-28s %8.2f us per graph (best of 5; mean %.2f)  -> %6.2f us per unit%s\n", name, best * 1000.f / reps, sum * 1000.f / reps / 5,
               best * 1000.f / reps / per_chain_units, herr ? "   ** barrier TIMEOUT **" : "");
        HIP_OK(hipGraphExecDestroy(ge)); HIP_OK(hipGraphDestroy(g));
        return best * 1000.f / reps;
    };
    const float t_zero = timed("zero kernel alone", [&](int) {}, 1);
    (void)t_zero;
    timed("empty kernels", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(kThreads), 0, st, 0); }, chain);
    timed("two launches: stats+consumer", [&](int i) {
        hipLaunchKernelGGL(stats_kernel, dim3(wgs), dim3(kThreads), 0, st, X, N, acc + (size_t)i * kRep * 2 * kH * 2);
        hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(kThreads), 0, st, X, Y, N, acc + (size_t)i * kRep * 2 * kH * 2);
    }, chain);
    timed("consumer alone", [&](int i) {
        hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(kThreads), 0, st, X, Y, N, acc + (size_t)i * kRep * 2 * kH * 2);
    }, chain);
    timed("stats alone", [&](int i) {
        hipLaunchKernelGGL(stats_kernel, dim3(wgs), dim3(kThreads), 0, st, X, N, acc + (size_t)i * kRep * 2 * kH * 2);
    }, chain);
    timed("fused: barrier inside", [&](int i) {
        hipLaunchKernelGGL(fused_kernel, dim3(wgs), dim3(kThreads), 0, st, X, Y, N, acc + (size_t)i * kRep * 2 * kH * 2, ctrs + 32 * i, err);
    }, chain);
    for (int K : {1, 4, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "bare barriers, K = %d / launch", K);
        timed(nm, [&](int i) { hipLaunchKernelGGL(barrier_kernel, dim3(wgs), dim3(kThreads), 0, st, ctrs + 32 * 64 * i, K, err); }, chain);
    }
    // correctness of the fused form: Y = X * mean + meansq-ish coefficients, same as the two-launch form
    std::vector<float> y1(N * kH), y2(N * kH);
    HIP_OK(hipMemset(acc, 0, (size_t)n_acc * 8)); HIP_OK(hipMemset(ctrs, 0, (size_t)n_ctr * 4));
    hipLaunchKernelGGL(stats_kernel, dim3(wgs), dim3(kThreads), 0, st, X, N, acc);
    hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(kThreads), 0, st, X, Y, N, acc);
    HIP_OK(hipStreamSynchronize(st)); HIP_OK(hipMemcpy(y1.data(), Y, N * kH * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemset(acc, 0, (size_t)n_acc * 8)); HIP_OK(hipMemset(Y, 0, N * kH * 4));
    hipLaunchKernelGGL(fused_kernel, dim3(wgs), dim3(kThreads), 0, st, X, Y, N, acc, ctrs, err);
    HIP_OK(hipStreamSynchronize(st)); HIP_OK(hipMemcpy(y2.data(), Y, N * kH * 4, hipMemcpyDeviceToHost));
    printf("fused == two-launch result: %s\n", memcmp(y1.data(), y2.data(), N * kH * 4) == 0 ? "bitwise equal" : "DIFFERENT");
    return 0;
}
