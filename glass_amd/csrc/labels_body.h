// K4b, the body of the label launch as a device function (see labels.hip for the algorithm) — shared by the stand-alone
// kernels there and by the label workgroup of the step's head launch (dense.hip: glass_step_head_f32).
#pragma once
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

// The body of the label launch: every thread of a kLabThreads-wide workgroup calls it (the stand-alone kernels of labels.hip; the
// label workgroup of the step's head launch, dense.hip step_head_kernel).
template <bool kGather>
__device__ __forceinline__ void batch_labels_body(const BatchSrc& src, int n_pos, int64_t* __restrict__ pos_dst,
                                                  uint32_t* __restrict__ y_dst, int64_t y_words, uint8_t* __restrict__ mask,
                                                  int32_t* __restrict__ lab_rows, int32_t* __restrict__ lab_count,
                                                  int32_t* __restrict__ owner, int64_t N, int incremental) {
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

template <bool kGather>
__global__ __launch_bounds__(kLabThreads) void batch_labels_kernel(const BatchSrc src, int n_pos,
                                                                   int64_t* __restrict__ pos_dst,
                                                                   uint32_t* __restrict__ y_dst, int64_t y_words,
                                                                   uint8_t* __restrict__ mask,
                                                                   int32_t* __restrict__ lab_rows,
                                                                   int32_t* __restrict__ lab_count,
                                                                   int32_t* __restrict__ owner, int64_t N,
                                                                   int incremental) {
    batch_labels_body<kGather>(src, n_pos, pos_dst, y_dst, y_words, mask, lab_rows, lab_count, owner, N, incremental);
}

}  // namespace glass
