// Feasibility probe (laboratory, round 6): K1 at hpo_neuro-shape (N = 14 587, nnz = 6.48 M, hidden 64) with a COLUMN slice of X
// resident in LDS and the edges streamed — the dual of tools/lab/lds_stream_probe.hip (row groups, X streamed through LDS).
//   form 2: workgroup (row group rg of 8, column pair cp of 32) keeps X[:, 2cp .. 2cp+1] for ALL rows (114 KB) and walks the
//           edges of its row group: per edge one (col, val) pair from global (coalesced, 8 B), one ds_read_b64 gather, 2 FMAs.
//   form 4: workgroup (rg of 8, column quad cq of 16, source half sh of 2) keeps X[half, 4cq .. 4cq+3] (114 KB) and walks the edges
//           of its row group whose source lies in its half: one ds_read_b128 gather and 4 FMAs per edge; the two halves' partial
//           rows would be added by the consumer.
// Rows are padded to whole 64-edge chunks (a wave takes a row's chunks one after the other, every lane accumulating its own
// edges; ONE cross-lane reduction per row).  Measures the launch only (synthetic uniform columns, val = 1): the bound a real
// kernel (plan format, ragged rows, A^T) could approach.  Compare: the committed workgroup kernel 75-78 us per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <random>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kThreads = 1024, kH = 64;

struct Item { int row, first_chunk, n_chunks, pad; };   // one row of a row group: its chunks of 64 (col, val) pairs

template <int W>  // W = columns per workgroup (2 or 4)
__global__ __launch_bounds__(kThreads) void colres_kernel(const float* __restrict__ X, int n_src0, int n_src, const Item* __restrict__ items,
                                                          const int* __restrict__ item_ptr, const int2* __restrict__ edges,
                                                          float* __restrict__ Y, int n_col_groups) {
    extern __shared__ float xs[];  // [n_src][W]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware: consecutive workgroup ids go round the 8 XCDs, so the row group (8 of them: one per XCD) is the id's low
    // part — all column groups of a row group then stream the SAME edge range through ONE L2 (with the plain order every XCD
    // streamed every row group's edges: 346 us per launch)
    const int n_parts = gridDim.x / n_col_groups;
    const int part = blockIdx.x % n_parts, cg = blockIdx.x / n_parts;   // part = row group (x source half for W = 4)
    // X slice -> LDS (strided global reads: 4 W bytes per row)
    for (int r = tid; r < n_src; r += kThreads) {
        if (W == 2) {
            const float2 v = *reinterpret_cast<const float2*>(X + (size_t)(n_src0 + r) * kH + 2 * cg);
            *reinterpret_cast<float2*>(xs + 2 * r) = v;
        } else {
            const float4 v = *reinterpret_cast<const float4*>(X + (size_t)(n_src0 + r) * kH + 4 * cg);
            *reinterpret_cast<float4*>(xs + 4 * r) = v;
        }
    }
    __syncthreads();
    const int i0 = item_ptr[part], i1 = item_ptr[part + 1];
    for (int it = i0 + wv; it < i1; it += kThreads / 64) {   // one row per wave at a time
        const Item I = items[it];
        float acc[W];
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = 0.f;
        const int2* e = edges + (size_t)I.first_chunk * 64 + lane;
        int2 ch[8];   // the row's chunks requested together (a row has <= 8 chunks here): 8 coalesced loads in flight per wave
#pragma unroll
        for (int c = 0; c < 8; ++c) ch[c] = c < I.n_chunks ? e[(size_t)c * 64] : make_int2(0, 0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int2 cur = ch[c];
            const float val = __int_as_float(cur.y);
            if (W == 2) {
                const float2 x = *reinterpret_cast<const float2*>(xs + 2 * cur.x);
                acc[0] = fmaf(val, x.x, acc[0]);
                acc[1] = fmaf(val, x.y, acc[1]);
            } else {
                const float4 x = *reinterpret_cast<const float4*>(xs + 4 * cur.x);
                acc[0] = fmaf(val, x.x, acc[0]);
                acc[1] = fmaf(val, x.y, acc[1]);
                acc[2] = fmaf(val, x.z, acc[2]);
                acc[3] = fmaf(val, x.w, acc[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) acc[k] += __shfl_xor(acc[k], s);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < W; ++k) Y[(size_t)I.row * kH + W * cg + k] = acc[k];
        }
    }
}

int main(int argc, char** argv) {
    const int N = 14587;
    const long long nnz = argc > 1 ? atoll(argv[1]) : 6476348;
    const int deg = (int)(nnz / N);
    std::mt19937 rng(1);
    std::vector<float> hx((size_t)N * kH);
    for (auto& v : hx) v = (float)(rng() % 1000) / 1000.f;
    float *X, *Y;
    HIP_OK(hipMalloc(&X, hx.size() * 4)); HIP_OK(hipMalloc(&Y, hx.size() * 4 * 2));
    HIP_OK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; HIP_OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int W : {2, 4}) {
        const int halves = W == 4 ? 2 : 1, n_col_groups = kH / W, n_rg = 8;
        const int n_src = (N + halves - 1) / halves;
        // items: every row once per (row group, half); a row's edges of one half: deg / halves, padded to 64
        std::vector<Item> items; std::vector<int> item_ptr; std::vector<int2> edges;
        const int rows_per_rg = (N + n_rg - 1) / n_rg;
        for (int h = 0; h < halves; ++h)
            for (int g = 0; g < n_rg; ++g) {
                item_ptr.push_back((int)items.size());
                for (int r = g * rows_per_rg; r < std::min(N, (g + 1) * rows_per_rg); ++r) {
                    const int d = deg / halves + (int)(rng() % 17) - 8;
                    const int nch = (d + 63) / 64;
                    items.push_back({r, (int)(edges.size() / 64), nch, 0});
                    for (int k = 0; k < nch * 64; ++k) {
                        const float one = k < d ? 1.f : 0.f;
                        edges.push_back(make_int2((int)(rng() % n_src), *reinterpret_cast<const int*>(&one)));
                    }
                }
            }
        item_ptr.push_back((int)items.size());
        Item* d_items; int* d_ptr; int2* d_edges;
        HIP_OK(hipMalloc(&d_items, items.size() * sizeof(Item))); HIP_OK(hipMalloc(&d_ptr, item_ptr.size() * 4));
        HIP_OK(hipMalloc(&d_edges, edges.size() * 8));
        HIP_OK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_ptr, item_ptr.data(), item_ptr.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_edges, edges.data(), edges.size() * 8, hipMemcpyHostToDevice));
        const size_t lds = (size_t)n_src * W * 4;
        const int grid = n_col_groups * n_rg * halves;
        auto launch = [&]() {
            // (W = 4: the half's partial rows go to Y + half * N * kH; n_src0 = the half's first source row — the probe uses part % ... below)
            if (W == 2) hipLaunchKernelGGL(colres_kernel<2>, dim3(grid), dim3(kThreads), lds, st, X, 0, n_src, d_items, d_ptr, d_edges, Y, n_col_groups);
            else hipLaunchKernelGGL(colres_kernel<4>, dim3(grid), dim3(kThreads), lds, st, X, 0, n_src, d_items, d_ptr, d_edges, Y, n_col_groups);
        };
        if (W == 2) HIP_OK(hipFuncSetAttribute((const void*)colres_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else HIP_OK(hipFuncSetAttribute((const void*)colres_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int i = 0; i < 5; ++i) launch();
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < 30; ++i) launch();
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        printf("W = %d columns per workgroup (%d workgroups, %zu KB of LDS, %zu padded edges = %.2f x nnz): %.1f us per launch\n", W, grid,
               lds >> 10, edges.size(), (double)edges.size() / (double)(nnz * (W == 4 ? 1 : 1)), ms * 1000.f / 30);
        HIP_OK(hipFree(d_items)); HIP_OK(hipFree(d_ptr)); HIP_OK(hipFree(d_edges));
    }
    return 0;
}
