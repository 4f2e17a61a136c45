// Node-bucketed entry lists of a padded node matrix (pool.hip) and the exact fixed-point sum their consumers use, shared
// with the fused readout's large-batch scatter (readout.hip).
#pragma once
#include "common.h"

namespace glass {

// hi * 2^-20 + lo * 2^-60 in two 64-bit integers: integer addition commutes, so a sum over a list whose order is
// arbitrary (filled through atomics) is the same bits every run.  Quantum 2^-60 per addend (smaller contributions are
// truncated: the sum is deterministic and exact in that fixed point, not in real arithmetic), range |addend| < 2^41.
// The fraction limb is kept below 2^40 by carrying into hi after every add (the total hi * 2^40 + lo is what counts, so
// when the carry happens does not change the result), which leaves bits 50.. of lo free for a STICKY MARK: a NaN /
// infinite / out-of-range addend does not saturate into finite garbage (fmax(NaN, x) = x did exactly that) but sets the
// mark, and value() — also after the limbs of up to 1024 lanes were summed through LDS (their fractions stay below 2^50,
// their marks add up to at most 2^60) — returns NaN when a mark is there, as the reference's float scatter-add propagates it
// (tests/test_gpu_kernels.py::test_pool_backward_propagates_nonfinite).
struct ExactSum {
    long long hi, lo;
    static constexpr long long kMark = 1ll << 50;
    __device__ __forceinline__ void add(float v) {
        const double sv = (double)v * 1048576.0;  // 2^20
        if (!(fabs(sv) <= 4.0e18)) {
            lo |= kMark;
            return;
        }
        const double fl = floor(sv);
        lo += (long long)((sv - fl) * 1099511627776.0);  // 2^40
        hi += (long long)fl + ((lo >> 40) & 1);
        lo &= ~(1ll << 40);
    }
    __device__ __forceinline__ float value() const {
        if (lo >= kMark) return __int_as_float(0x7fc00000);
        return (float)(((double)hi + (double)lo * (1.0 / 1099511627776.0)) * (1.0 / 1048576.0));
    }
};

struct BucketLists {
    const int32_t* off;   // [n_nodes + 1]: entries of node i are list[off[i] .. off[i + 1])
    const int32_t* list;  // subgraph index of each entry (plain form) — or (pair << 1) | both-valid for Smax == 2 pairs
    float* scale;         // [B] (plain form with a pooling mode) or nullptr
};

// words (int32) of workspace for bucket_build
int64_t bucket_ws_words(int64_t n_nodes, int64_t B, int64_t Smax, bool pair_form);
// Four small launches on `st` (zero, rank, scan, fill; + the subgraph scales when mode >= 0 and not the pair form).
// pair_form (Smax == 2 only): the list carries the pair's count in bit 0 instead of a scale array.
int bucket_build(const int64_t* pos, int64_t B, int64_t Smax, int mode, bool pair_form, int64_t n_nodes, void* ws, hipStream_t st,
                 BucketLists* out, bool dedup = false);  // dedup: a node named twice by one row is listed once for it

}  // namespace glass
