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
#include "labels_body.h"



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
