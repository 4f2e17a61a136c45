// One-shot peer all-reduce of the small gradient bucket, fused with Adam (round 6; SURVEY.md 8e: "one RCCL all-reduce of
// gradients over xGMI", reference impl/train.py:10-16 + torch.optim.Adam at GLASSTest.py:213).
//
// The benchmarked data-parallel step exchanges ~0.2 MB of gradients (hidden 64, two layers).  A ring all-reduce of that
// payload is all latency: 2 (N - 1) dependent hops at 8 ranks, ~33 us under the model of glass_amd/dist.py against a 232 us
// step — and it sits between the last gradient and Adam, fully exposed.  xGMI is point to point: every rank can READ every
// peer's gradient arena directly (peer mappings of the other GPUs' memory: hipIpc handles here, hipDeviceEnablePeerAccess
// inside one process), N - 1 links side by side.  So the exchange is ONE launch per rank and step:
//   1. publish "my gradients of step s are final" (a 64-bit sequence number in this rank's flag block, system-scope release);
//   2. every workgroup waits until every peer's flag shows s (bounded spin, system-scope loads), then acquires;
//   3. element k of the mean gradient = (g_0[k] + g_1[k] + ... + g_{N-1}[k]) / N, summed in RANK order on every rank — the
//      same bits everywhere —, fed straight to Adam on this rank's replica (the arithmetic of adam_kernel, linear.hip);
//   4. the last workgroup to finish publishes "I am done reading step s" and waits for every peer's done flag before the
//      launch ends: a rank's next backward pass may then overwrite its arena (stream order puts it behind this launch).
// No collective library call, no host round trip, capturable in the step's hipGraph like any kernel of this library.
// A flag that does not arrive within `spin_limit` polls sets *status (sticky, non-zero) and the launch ends WITHOUT
// touching parameters or optimizer state: an error the host sees at its next synchronisation point — never a hang.
// RCCL (dist.GradExchange) stays the default exchange; this form is opt-in (glass_amd/peer.py) until a multi-GPU run has
// measured it.  Functional test: two processes on ONE GPU through hipIpc handles (tests/test_gpu_peer.py).
#include "common.h"
#include <string.h>

namespace glass {

constexpr int kPeerMax = 8;
struct PeerArgs {
    const float* grad[kPeerMax];            // every rank's gradient arena as THIS process addresses it (own: plain pointer)
    unsigned long long* flags[kPeerMax];    // every rank's flag block: [0] ready sequence, [1] done sequence
    int world, rank;
};

__device__ __forceinline__ unsigned long long peer_flag_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wait until flags[p][which] >= seq for every peer; false on timeout
__device__ __forceinline__ bool peer_wait_all(const PeerArgs& a, int which, unsigned long long seq, long long spin_limit) {
    for (int p = 0; p < a.world; ++p) {
        if (p == a.rank) continue;
        long long it = 0;
        while (peer_flag_load(a.flags[p] + which) < seq) {
            if (++it > spin_limit) return false;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    return true;
}

__global__ __launch_bounds__(kBlock) void peer_allreduce_adam_kernel(PeerArgs a, int64_t n, float* __restrict__ p,
                                                                     float* __restrict__ m, float* __restrict__ v,
                                                                     const float* __restrict__ lr_dev, float beta1, float beta2,
                                                                     float eps, float weight_decay, int64_t* __restrict__ step_dev,
                                                                     unsigned long long* __restrict__ seq_dev,
                                                                     int* __restrict__ status, long long spin_limit,
                                                                     float* __restrict__ mean_out) {
    __shared__ int s_ok;
    // seq_dev = [sequence of the last finished exchange, ticket]: every workgroup reads [0] first; the last one advances it
    const unsigned long long seq = seq_dev[0] + 1ull;
    const int64_t step_now = step_dev[0] + 1;
    const AdamCoef c = adam_coef(step_now, lr_dev[0], beta1, beta2, eps, weight_decay);
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0)  // (the gradients were written by earlier launches of this stream: final and written back)
            __hip_atomic_store(a.flags[a.rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool ok = *reinterpret_cast<volatile int*>(status) == 0 && peer_wait_all(a, 0, seq, spin_limit);
        if (!ok) atomicOr(status, 1);
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    if (ok) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: no stale line of a peer's arena (read in the previous step) survives
        const float inv = 1.f / (float)a.world;
        for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
            float g = 0.f;
            for (int r = 0; r < a.world; ++r) g += a.grad[r][k];   // rank order: the same sum on every rank
            g *= inv;
            if (mean_out) mean_out[k] = g;
            adam_update(c, p, g, m, v, k);
        }
    }
    // the last workgroup: done flag, wait for the peers' done flags, advance the counters
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long taken = atomicAdd(seq_dev + 1, 1ull) + 1ull;
        if (taken == gridDim.x) {
            seq_dev[1] = 0;
            __hip_atomic_store(a.flags[a.rank] + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!(ok && peer_wait_all(a, 1, seq, spin_limit))) atomicOr(status, 1);
            seq_dev[0] = seq;                    // (advanced even after a timeout: the next launch pairs with the peers' next one)
            if (*reinterpret_cast<volatile int*>(status) == 0) step_dev[0] = step_now;
        }
    }
}

}  // namespace glass

using namespace glass;

extern "C" int glass_peer_allreduce_adam_f32(const glass_peer_group* grp, int64_t n, float* param, float* exp_avg, float* exp_avg_sq,
                                             const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                                             int64_t* step_dev, uint64_t* seq_dev, int32_t* status_dev, int64_t spin_limit,
                                             float* mean_out, void* stream) {
    GLASS_REQUIRE(grp && param && exp_avg && exp_avg_sq && lr_dev && step_dev && seq_dev && status_dev && n > 0 && spin_limit > 0,
                  "peer_allreduce_adam: bad arguments");
    GLASS_REQUIRE(grp->world >= 1 && grp->world <= kPeerMax && grp->rank >= 0 && grp->rank < grp->world,
                  "peer_allreduce_adam: world size 1 .. %d", kPeerMax);
    PeerArgs a{};
    a.world = grp->world;
    a.rank = grp->rank;
    for (int r = 0; r < grp->world; ++r) {
        GLASS_REQUIRE(grp->grad[r] && grp->flags[r] && (reinterpret_cast<uintptr_t>(grp->flags[r]) & 7u) == 0,
                      "peer_allreduce_adam: rank %d has no arena / flag mapping", r);
        a.grad[r] = grp->grad[r];
        a.flags[r] = reinterpret_cast<unsigned long long*>(grp->flags[r]);
    }
    int64_t blocks = ceil_div(n, kBlock);
    if (blocks > 256) blocks = 256;  // every workgroup polls the peers' flags: resident together, one per CU at most
    hipLaunchKernelGGL(peer_allreduce_adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, n, param, exp_avg,
                       exp_avg_sq, lr_dev, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, step_dev,
                       reinterpret_cast<unsigned long long*>(seq_dev), status_dev, (long long)spin_limit, mean_out);
    return launch_status("glass_peer_allreduce_adam_f32");
}

// ---- set-up helpers (NOT on the step's path; the only entries of this library that allocate): the arena and the flag block a
// rank exposes to its peers must be whole allocations of the HIP runtime (hipIpcGetMemHandle does not take a sub-range of a
// framework's pooled block).
extern "C" int glass_peer_alloc(int64_t bytes, void** out) {
    GLASS_REQUIRE(out && bytes > 0, "peer_alloc: bad arguments");
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)bytes) != hipSuccess || hipMemset(p, 0, (size_t)bytes) != hipSuccess) {
        set_error("peer_alloc: hipMalloc of %lld bytes failed", (long long)bytes);
        return GLASS_E_UNSUPPORTED;
    }
    *out = p;
    return 0;
}
extern "C" int glass_peer_free(void* p) { return (p == nullptr || hipFree(p) == hipSuccess) ? 0 : GLASS_E_ARG; }
extern "C" int glass_peer_export(void* p, void* handle64) {
    GLASS_REQUIRE(p && handle64, "peer_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), p) != hipSuccess) {
        set_error("peer_export: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on this pool)");
        return GLASS_E_UNSUPPORTED;
    }
    return 0;
}
extern "C" int glass_peer_import(const void* handle64, void** out) {
    GLASS_REQUIRE(handle64 && out, "peer_import: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        set_error("peer_import: hipIpcOpenMemHandle failed");
        return GLASS_E_UNSUPPORTED;
    }
    *out = p;
    return 0;
}
extern "C" int glass_peer_close(void* p) { return (p == nullptr || hipIpcCloseMemHandle(p) == hipSuccess) ? 0 : GLASS_E_ARG; }
