// K4b: the labels of one subgraph batch for a replayed training step.
// reference: utils.MaxZOZ (impl/utils.py:32-45: z = 0; z[pos[pos >= 0]] = 1) and the batch hand-over of
// ZGDataloader (impl/SubGDataset.py:75-96).
//
// One launch per step, ONE workgroup (the work is a few thousand entries; what it costs is its chain of dependent
// round trips, not bandwidth):
//   * the batch (pos, target) is copied into the step's fixed buffers (what glass_copy_pair did);
//   * the label bytes are maintained INCREMENTALLY: the nodes the previous batch named are cleared, the new ones set
//     — no pass over the N bytes (incremental == 0: the N bytes are zero-filled here first);
//   * the UNIQUE labeled rows are listed in first-occurrence order (lab_rows, lab_count): the entry with the lowest
//     index naming a node owns it (integer atomicMin on a scratch word per named node: the result does not depend on
//     the order of the atomics), owners are compacted in entry order by ballots — deterministic.  The scratch words hold
//     INT32_MAX between calls (the caller fills them once, every call restores the words it touched), so setting the
//     label bytes and the atomicMin share a phase: three dependent round trips instead of four (6.7 -> 5.4 us).
// The list is what lets the comb pair of GLASSConv run ONE product per row (effective per-label weights, dense.hip):
// every row tile multiplies the unlabeled-row weight, the listed rows are recomputed with the labeled-row weight by a
// few extra workgroups of the same launch.
#include "common.h"

namespace glass {

constexpr int kLabThreads = 1024;

// Where the batch comes from.  Plain: pos_src / y_src ARE the batch.  Gather (kGather): they are the data set's whole
// padded node matrix [n_all, smax] and target matrix [n_all, y_row_words]; the batch is the rows idx[0 .. n_pos / smax) —
// what ZGDataloader's `pos[perm], y[perm]` (impl/SubGDataset.py:69-72, 92-96) materialised with two index kernels and a
// copy is read in place (one more dependent load per entry; a row index outside [0, n_all) reads as padding).
struct BatchSrc {
    const int64_t* pos;
    const uint32_t* y;
    const int64_t* idx;
    int smax, y_row_words;
    int64_t n_all;
};

template <bool kGather>
__device__ __forceinline__ int64_t batch_entry(const BatchSrc& s, int e) {
    if (!kGather) return s.pos[e];
    const int r = e / s.smax;
    const int64_t row = s.idx[r];
    return (row >= 0 && row < s.n_all) ? s.pos[row * s.smax + (e - r * s.smax)] : -1;
}

template <bool kGather>
__global__ __launch_bounds__(kLabThreads) void batch_labels_kernel(const BatchSrc src, int n_pos,
                                                                   int64_t* __restrict__ pos_dst,
                                                                   uint32_t* __restrict__ y_dst, int64_t y_words,
                                                                   uint8_t* __restrict__ mask,
                                                                   int32_t* __restrict__ lab_rows,
                                                                   int32_t* __restrict__ lab_count,
                                                                   int32_t* __restrict__ owner, int64_t N,
                                                                   int incremental) {
    __shared__ int wave_cnt[kLabThreads / kWave];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // this thread's first entry of the NEW batch is requested before anything else: its round trip runs beside phase 1's
    // (the barrier between the phases would otherwise put the two loads in a row)
    const int64_t p_first = tid < n_pos ? batch_entry<kGather>(src, tid) : -1;
    // 1. labels of the previous batch off (or all N bytes)
    if (incremental) {
        if (pos_dst)
            for (int e = tid; e < n_pos; e += kLabThreads) {
                const int64_t p = pos_dst[e];
                if (p >= 0 && p < N) mask[p] = 0;
            }
    } else {
        const int64_t head = (16 - (reinterpret_cast<uintptr_t>(mask) & 15u)) & 15u;  // bytes up to 16-B alignment
        for (int64_t k = tid; k < head && k < N; k += kLabThreads) mask[k] = 0;
        const int64_t vecs = N > head ? (N - head) / 16 : 0;
        uint4* mv = reinterpret_cast<uint4*>(mask + head);
        for (int64_t k = tid; k < vecs; k += kLabThreads) mv[k] = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t k = head + vecs * 16 + tid; k < N; k += kLabThreads) mask[k] = 0;
    }
    __syncthreads();
    // 2. the new batch: fixed buffers, label bytes; the lowest entry index naming a node owns it (owner words are
    //    INT32_MAX on entry)
    for (int e = tid; e < n_pos; e += kLabThreads) {
        const int64_t p = e == tid ? p_first : batch_entry<kGather>(src, e);
        if (pos_dst) pos_dst[e] = p;
        if (p >= 0 && p < N) {
            mask[p] = 1;
            atomicMin(owner + p, e);
        }
    }
    for (int64_t k = tid; k < y_words; k += kLabThreads) {
        if (!kGather) {
            y_dst[k] = src.y[k];
        } else {
            const int64_t r = k / src.y_row_words;
            const int64_t row = src.idx[r];
            y_dst[k] = (row >= 0 && row < src.n_all) ? src.y[row * src.y_row_words + (k - r * src.y_row_words)] : 0u;
        }
    }
    __syncthreads();
    // 3. owners, compacted in entry order
    int base = 0;
    for (int e0 = 0; e0 < n_pos; e0 += kLabThreads) {
        const int e = e0 + tid;
        const int64_t p = e0 == 0 ? p_first : (e < n_pos ? batch_entry<kGather>(src, e) : -1);
        const bool own = p >= 0 && p < N && __hip_atomic_load(owner + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e;
        const unsigned long long bal = __ballot(own);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        int off = base, total = 0;
        for (int ww = 0; ww < kLabThreads / kWave; ++ww) {
            if (ww < w) off += wave_cnt[ww];
            total += wave_cnt[ww];
        }
        if (own) lab_rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)p;
        base += total;
        __syncthreads();
    }
    if (tid == 0) lab_count[0] = base;
    // 4. the owner words of the named nodes back to INT32_MAX (every read of them lies before the loop's last barrier)
    for (int e = tid; e < n_pos; e += kLabThreads) {
        const int64_t p = e == tid ? p_first : batch_entry<kGather>(src, e);
        if (p >= 0 && p < N) owner[p] = INT32_MAX;
    }
}

}  // namespace glass

using namespace glass;

extern "C" int64_t glass_batch_labels_ws_bytes(int64_t n_nodes) {
    return n_nodes > 0 ? n_nodes * (int64_t)sizeof(int32_t) : GLASS_E_ARG;
}

extern "C" int glass_batch_labels(const int64_t* pos_src, int64_t n_pos, int64_t* pos_dst, const void* y_src, void* y_dst,
                                  int64_t y_bytes, uint8_t* mask, int32_t* lab_rows, int32_t* lab_count, void* ws,
                                  int64_t n_nodes, int incremental, void* stream) {
    GLASS_REQUIRE(pos_src && mask && lab_rows && lab_count && ws, "batch_labels: null pointer");
    GLASS_REQUIRE(n_pos > 0 && n_pos < (1ll << 30) && n_nodes > 0 && n_nodes < (1ll << 31), "batch_labels: bad sizes");
    GLASS_REQUIRE(y_bytes == 0 || (y_src && y_dst && y_bytes > 0 && y_bytes % 4 == 0 &&
                                   ((reinterpret_cast<uintptr_t>(y_src) | reinterpret_cast<uintptr_t>(y_dst)) & 3u) == 0),
                  "batch_labels: the target copy has 4-byte granularity");
    const BatchSrc src{pos_src, (const uint32_t*)y_src, nullptr, 1, 1, 0};
    hipLaunchKernelGGL(batch_labels_kernel<false>, dim3(1), dim3(kLabThreads), 0, (hipStream_t)stream, src, (int)n_pos, pos_dst,
                       (uint32_t*)y_dst, y_bytes / 4, mask, lab_rows, lab_count, (int32_t*)ws, n_nodes, incremental);
    return launch_status("glass_batch_labels");
}

extern "C" int glass_batch_labels_gather(const int64_t* pos_all, int64_t n_all, int64_t smax, const void* y_all,
                                         int64_t y_row_bytes, const int64_t* idx, int64_t n_idx, int64_t* pos_dst, void* y_dst,
                                         uint8_t* mask, int32_t* lab_rows, int32_t* lab_count, void* ws, int64_t n_nodes,
                                         int incremental, void* stream) {
    GLASS_REQUIRE(pos_all && idx && mask && lab_rows && lab_count && ws, "batch_labels_gather: null pointer");
    GLASS_REQUIRE(n_all > 0 && smax > 0 && n_idx > 0 && n_idx * smax < (1ll << 30) && smax < (1ll << 30) && n_nodes > 0 &&
                      n_nodes < (1ll << 31),
                  "batch_labels_gather: bad sizes");
    GLASS_REQUIRE(y_row_bytes == 0 || (y_all && y_dst && y_row_bytes > 0 && y_row_bytes % 4 == 0 && y_row_bytes < (1ll << 30) &&
                                       ((reinterpret_cast<uintptr_t>(y_all) | reinterpret_cast<uintptr_t>(y_dst)) & 3u) == 0),
                  "batch_labels_gather: the target rows have 4-byte granularity");
    const BatchSrc src{pos_all, (const uint32_t*)y_all, idx, (int)smax, y_row_bytes ? (int)(y_row_bytes / 4) : 1, n_all};
    hipLaunchKernelGGL(batch_labels_kernel<true>, dim3(1), dim3(kLabThreads), 0, (hipStream_t)stream, src, (int)(n_idx * smax),
                       pos_dst, (uint32_t*)y_dst, n_idx * (y_row_bytes / 4), mask, lab_rows, lab_count, (int32_t*)ws, n_nodes,
                       incremental);
    return launch_status("glass_batch_labels_gather");
}
