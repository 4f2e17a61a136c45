/* glass_hip.h — C ABI of libglass_hip.so: the MI355X (gfx950) kernels behind GLASS's labeled
 * message-passing hot path.
 *
 * This is the LOWER drop-in boundary of SURVEY.md §8(b): the entry points a binding of the
 * reference would call in place of the third-party kernels its Python dispatches to today
 * (ATen sparse addmm, nn.Embedding, index_put, PyG GraphNorm / global_*_pool → torch_scatter).
 * Each function cites the reference call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers owned by the caller, sizes as int64_t, `stream` is a
 *    hipStream_t passed as void* (NULL = default stream). No torch types, no exceptions.
 *  - every function returns 0 on success, a positive hipError_t, or a negative GLASS_E_* code;
 *    glass_last_error_string() describes the last failure on the calling thread.
 *  - functions only ENQUEUE work on `stream`: no allocation, no host synchronisation, no global
 *    state (hipGraph-capturable). Scratch is caller-provided; its size comes from a *_ws_bytes()
 *    query that is pure host arithmetic.
 *  - all floating data is fp32, row-major, with an explicit leading dimension (`ld*`, in
 *    elements) so outputs can land inside wider buffers (the [g || x_] concat, the JK concat).
 *  - results are deterministic (bitwise repeatable run to run) unless a function says otherwise.
 */
#ifndef GLASS_HIP_H
#define GLASS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {

typedef struct glass_gn_bwd_src glass_gn_bwd_src;
typedef struct glass_gn_src glass_gn_src; /* exact GraphNorm accumulators as a kernel input: defined with K5's entries */
#endif

#define GLASS_ABI_VERSION 6

#define GLASS_E_ARG (-1)       /* bad argument (null pointer, negative size, misaligned ld) */
#define GLASS_E_PLAN (-2)      /* plan blob does not match the call (magic / sizes) */
#define GLASS_E_UNSUPPORTED (-3)
#define GLASS_E_WS (-4)        /* this size needs the workspace form of the call (see its *_ws_bytes query): the entry point
                                  promises bitwise repeatable results and does not fall back to float atomics */

/* pool modes: AddPool / MeanPool / MaxPool / SizePool (impl/models.py:294-319) */
#define GLASS_POOL_SUM 0
#define GLASS_POOL_MEAN 1
#define GLASS_POOL_MAX 2
#define GLASS_POOL_SIZE 3

/* activation fused into a kernel: none, or ELU(alpha=1) (GLASSTest.py:143) */
/* Activation codes — and the OPTIONS of a dense call: the `act` argument of the four glass_dual_linear_{fwd,bwd,dgrad,wgrad}_f32
 * entries (and the `gn_act` argument of glass_comb_eff_fwd_f32) is a word, activation code in bits 0..7 (GLASS_ACT_MASK), options above.  The library keeps NO mutable state: no
 * environment variable is read, nothing is process-global; what a call does follows from its arguments alone, so two threads
 * on two streams with different options each get their own behaviour. */
#define GLASS_ACT_MASK 0xff
#define GLASS_DENSE_F32_PRODUCTS 0x100 /* hidden 128 / 256 / 512 (LDS-tiled family) and, since round 6, the staged hidden-64 kernels
                                          (trans / comb forward, both fused backward launches; also read from the gn_act word of
                                          glass_comb_eff_fwd_f32 / _bwd_f32): form fp32 products with the f32-input MFMA
                                          (an fmaf chain, 1/16 of the bf16 matrix rate) instead of the default six bf16 partial
                                          products of 3-way split operands (glass_dense_caps.product_form).  Same operand images
                                          either way.  Differences of the default form at the edge of fp32's range: a +-Inf
                                          operand gives NaN where the f32 form gives +-Inf (pieces Inf, NaN, NaN); finite
                                          |x| >= 2^127 * (2 - 2^-8) rounds its first piece to Inf -> NaN; below |x| ~ 2^-117 the
                                          low pieces leave bf16's normal range and the product keeps ~16 instead of 24
                                          significant bits (measured against fp64, hidden 256: rel-inf <= 1.6e-6 for operands
                                          scaled down to 2^-115, 3e-5 .. 8e-5 at 2^-120; tests/test_gpu_hardening.py) — a
                                          caller with such operands passes this bit.  Ignored by the other families. */
#define GLASS_ACT_NONE 0
#define GLASS_ACT_ELU 1
#define GLASS_ACT_RELU 2 /* hidden-64 kernels, the GraphNorm kernels and the stand-alone mix (the reference's constructor default nn.ReLU(), impl/models.py:125,192; the pre-training path, GNNEmb.py:90) */

int glass_version(void);
const char* glass_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * K1  CSR aggregation  Y = A @ X         replaces `self.adj @ x` (impl/models.py:164) and its
 *     autograd backward `adj^T @ g` (call it again with the CSR of A^T).
 *
 * The launch schedule ("plan") depends only on the row pointer: short rows are dealt to wavefronts as
 * items (r0, r1, e0, e1) — runs of <= 64 consecutive rows holding <= 256 edges, edge-balanced, each
 * carrying its own edge range so that the kernel's index loads hang off one plan read; rows longer
 * than a threshold are cut into chunks that a whole workgroup reduces through LDS (the trailing
 * workgroups of the same launch), and rows longer
 * than one chunk are summed from per-chunk partial rows in a fixed order (no float atomics ->
 * bitwise repeatable).  Header words: magic, version (2), n_rows, nnz, #items, #long chunks,
 * #reduce rows, #partial slots, long threshold, long chunk, offsets of the three sections, and the
 * flat-mode factor (an item whose mean degree is <= factor * lane-groups-per-wave is walked as one
 * flat edge stream per lane group instead of row by row).
 *
 *   glass_spmm_plan_build: HOST function. `rowptr_host` = int32[n_rows+1] in host memory.
 *     Writes the plan into `plan_host` (int32 words; pass NULL to only query) and returns the
 *     number of int32 words needed in *plan_words. The caller copies the blob to the device.
 *     The first GLASS_PLAN_HEADER_WORDS words are the header the launch reads on the host.
 *   glass_spmm_ws_bytes: scratch bytes needed by glass_spmm_csr_f32 for feature width H.
 * ---------------------------------------------------------------------------------------- */
#define GLASS_PLAN_HEADER_WORDS 16
int glass_spmm_plan_build(const int32_t* rowptr_host, int64_t n_rows, int32_t* plan_host, int64_t* plan_words);
int64_t glass_spmm_ws_bytes(const int32_t* plan_header_host, int64_t H);
int glass_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, /* CSR of A, device */
                       const float* X, int64_t ldx,                                 /* [n_cols, H] */
                       float* Y, int64_t ldy,                                       /* [n_rows, H] */
                       int64_t n_rows, int64_t H,
                       const int32_t* plan_header_host, const int32_t* plan_dev, void* ws, void* stream);

/* K2  normalised edge values of buildAdj (impl/models.py:83-111) on an (row,col)-sorted COO:
 *     deg = segment-sum of w per row, deg<0.5 -> +1; aggr 0=mean w/deg[row], 1=sum w,
 *     2=gcn deg[row]^-1/2 * w * deg[col]^-1/2.  Duplicate (row,col) entries stay separate CSR
 *     entries (they act additively in the product, as in the reference's uncoalesced COO).
 *     Unknown aggr -> GLASS_E_UNSUPPORTED (the reference raises NotImplementedError). */
int glass_adj_values_f32(const int32_t* rowptr, const int32_t* col, const float* w, int64_t n_rows, int aggr,
                         float* deg_ws /*[n_rows], receives deg*/, float* val, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  max-zero-one label   replaces utils.MaxZOZ (impl/utils.py:32-45):  z[n]=0; z[pos>=0]=1
 * ---------------------------------------------------------------------------------------- */
int glass_maxzoz_i64(const int64_t* pos, int64_t n_pos /* B*Smax, -1 = padding */, int64_t* z, int64_t n_nodes,
                     void* stream);

/* K4b labels of one subgraph batch for a replayed training step: utils.MaxZOZ (impl/utils.py:32-45) as label BYTES plus
 *     the list of the unique labeled rows, and the batch hand-over of ZGDataloader (impl/SubGDataset.py:75-96), in one
 *     single-workgroup launch.  pos_src int64[n_pos] (-1 padding) is copied to pos_dst (may be NULL), y_bytes of target to
 *     y_dst (4-byte granularity; 0 = none).  mask uint8[n_nodes]: incremental != 0 — it holds the labels of the batch
 *     currently in pos_dst (all zero before the first call, pos_dst all -1): those are cleared, the new ones set, no pass
 *     over the n_nodes bytes; incremental == 0 — zero-filled here first.  lab_rows int32[n_pos] receives the UNIQUE labeled
 *     node ids in first-occurrence order, lab_count[0] their number (the entry with the lowest index naming a node owns
 *     it: integer atomicMin on a scratch word per named node, ordered compaction by ballots — deterministic).
 *     ws = int32[n_nodes] scratch (glass_batch_labels_ws_bytes): every word INT32_MAX before the first call — the caller
 *     fills it once; a call touches the named nodes' words only and restores them. */
int64_t glass_batch_labels_ws_bytes(int64_t n_nodes);
int glass_batch_labels(const int64_t* pos_src, int64_t n_pos, int64_t* pos_dst, const void* y_src, void* y_dst,
                       int64_t y_bytes, uint8_t* mask, int32_t* lab_rows, int32_t* lab_count, void* ws, int64_t n_nodes,
                       int incremental, void* stream);
/*     The same with the batch SELECTED in place — replaces `self.get_pos()[perm], self.get_y()[perm]` of the loaders
 *     (impl/SubGDataset.py:69-72, 92-96: two index kernels + the copy into the step's fixed buffers): pos_all int64[n_all, smax]
 *     and y_all [n_all, y_row_bytes] are the data set's whole matrices, idx int64[n_idx] the batch's rows (a row index outside
 *     [0, n_all) reads as an all-padding row with zero target).  The batch has n_idx * smax entries; pos_dst / y_dst (may be
 *     NULL / y_row_bytes == 0) receive exactly pos_all[idx], y_all[idx].  Everything else as glass_batch_labels. */
int glass_batch_labels_gather(const int64_t* pos_all, int64_t n_all, int64_t smax, const void* y_all, int64_t y_row_bytes,
                              const int64_t* idx, int64_t n_idx, int64_t* pos_dst, void* y_dst, uint8_t* mask,
                              int32_t* lab_rows, int32_t* lab_count, void* ws, int64_t n_nodes, int incremental,
                              void* stream);

/* K3+K4 fused label + embedding   replaces `mask=(z>0.5)` and `input_emb(x)`
 *     (impl/models.py:242-248):  out[n,:] = W[x[n],:],  mask[n] = label of node n.
 *     The label comes from `z` (int64[N], as MaxZOZ produced it) when z != NULL, else from
 *     `pos` (int64[n_pos], -1 pad) scattered here; both NULL -> every node labeled
 *     (impl/models.py:243-244) — unless n_pos < 0: then mask is an INPUT (written by glass_batch_labels)
 *     and is left alone (the same convention holds in glass_embed_norm_fwd_f32). An index of x outside [0,V) yields a zero row here; callers
 *     validate x once per dataset (nn.Embedding would raise IndexError). */
int glass_embed_label_f32(const int64_t* x, const float* W, int64_t V, const int64_t* z, const int64_t* pos,
                          int64_t n_pos, float* out, int64_t ldo, uint8_t* mask, int64_t n_nodes, int64_t H,
                          void* stream);
/*     backward of the embedding gather (ATen embedding_dense_backward, a scatter-add):
 *     dW[v,:] = sum_{n: x[n]=v} dout[n,:]  is  dW = S^T @ dout with S the [N,V] one-hot selection
 *     matrix, so it runs on K1: call glass_spmm_csr_f32 with the CSR of S^T (rows = table rows,
 *     cols = node ids in ascending order, val = 1), built once per dataset because x is static.
 *     That keeps the scatter-add atomic-free and bitwise repeatable, and a hot table row
 *     (use_one: every node -> row 1) is split over workgroups by the K1 plan. */

/* ------------------------------------------------------------------------------------------
 * K5' label-conditioned mix (impl/models.py:158-162 and 169-173)
 *     T = [T1 | T0] is the [N, 2H] output of the two stacked Linears (f1 then f0).
 *     a = act(T);  out = mask ? zr*a1 + (1-zr)*a0 : zr*a0 + (1-zr)*a1
 *     backward: dT1 = dout * (mask ? zr : 1-zr) * act'(T1), dT0 likewise with the roles swapped.
 *     z_ratio is a double so that 1 - z_ratio is formed as the reference forms it (Python float)
 *     before both factors are rounded to fp32.
 * ---------------------------------------------------------------------------------------- */
int glass_mix_fwd_f32(const float* T, int64_t ldt, const uint8_t* mask, double z_ratio, int act, float* out,
                      int64_t ldo, int64_t n_nodes, int64_t H, void* stream);
int glass_mix_bwd_f32(const float* dout, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask, double z_ratio,
                      int act, float* dT, int64_t lddt, int64_t n_nodes, int64_t H, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6  whole-graph GraphNorm (PyG GraphNorm with batch=None; impl/models.py:165,249,257,266,271)
 *     mu = mean_rows(x); o = x - alpha*mu; y = gamma*o*rsqrt(mean_rows(o^2)+eps) + beta
 *     optionally followed by ELU and by inverted dropout (the reference applies F.dropout right
 *     after every GraphNorm(+act): impl/models.py:166,251,259).
 *     Column sums are accumulated in fp64 (one pass: sum and sum of squares), reduced in a fixed
 *     order. `saved` = float[4*C]: mean, rstd, scale, shift (needed by the backward).
 *     Dropout: keep-mask from a counter-based hash keyed by (rng_state[0]=seed, rng_state[1]=step,
 *     call_id, element index); rng_state is DEVICE memory so captured graphs see new masks each
 *     replay once glass_rng_advance has run. p_drop = 0 disables it (rng_state may be NULL).
 * ---------------------------------------------------------------------------------------- */
int64_t glass_graphnorm_ws_bytes(int64_t n_rows, int64_t C);
int glass_graphnorm_fwd_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int64_t C,
                            const float* gamma, const float* beta, const float* alpha, float eps, float* saved,
                            int act, float p_drop, const uint64_t* rng_state, uint64_t call_id, void* ws,
                            void* stream);
/* backward: dx = dGraphNorm(dy) (+ addend when addend != NULL: a second gradient flowing into the same tensor,
 * e.g. the jumping-knowledge slice next to the next layer's input gradient — saves an elementwise launch). */
int glass_graphnorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx,
                            const float* addend, int64_t ldadd,
                            int64_t n_rows, int64_t C, const float* gamma, const float* alpha, const float* saved,
                            float* dgamma, float* dbeta, float* dalpha, int accumulate /* != 0: add into d* */,
                            int act, float p_drop, const uint64_t* rng_state, uint64_t call_id, void* ws,
                            void* stream);
/*     GraphNorm in pieces, for statistics produced elsewhere (the epilogue of the kernel that wrote x):
 *     finalize: saved[4*C] from n_src partial buffers (HOST array of device pointers), each [nblk][2][C_each]
 *     doubles, covering consecutive column blocks (C = n_src * C_each; several sources = the jumping-knowledge
 *     concatenation);  apply: y = dropout(act(x*scale + shift)) from saved. */
int glass_graphnorm_finalize_f32(const double* const* partials, int64_t n_src, int64_t nblk, int64_t C_each,
                                 int64_t n_rows, const float* gamma, const float* beta, const float* alpha, float eps,
                                 float* saved, void* stream);
int glass_graphnorm_apply_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows, int64_t C,
                              const float* saved, int act, float p_drop, const uint64_t* rng_state, uint64_t call_id,
                              void* stream);
/*     backward from partial sums produced elsewhere (glass_dual_linear_dgrad_f32 with gn_partial): finalize
 *     (parameter gradients, coefficients) + apply; ws as glass_graphnorm_ws_bytes. */
int glass_graphnorm_bwd_from_stats_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx,
                                       int64_t lddx, const float* addend, int64_t ldadd, int64_t n_rows, int64_t C,
                                       const float* gamma, const float* alpha, const float* saved,
                                       const double* partial, int64_t nblk, float* dgamma, float* dbeta,
                                       float* dalpha, int accumulate, int act, float p_drop, const uint64_t* rng_state,
                                       uint64_t call_id, void* ws, void* stream);
int glass_rng_advance(uint64_t* rng_state, void* stream); /* rng_state[1] += 1 */
/*     Measurement aid (bench.py `step_floor`): launch a kernel that does nothing with the given grid / block / dynamic-LDS
 *     geometry — the training step's chain of launches replayed with these is the latency floor of that chain. */
int glass_empty_launch(int64_t grid_x, int64_t grid_y, int64_t grid_z, int64_t block, int64_t lds_bytes, void* stream);
/*     Checker's hook: the keep-scales (0 or 1/(1-p)) the dropout with `call_id` draws for an [n_rows, C] tensor under the
 *     CURRENT rng_state words, written to out — so a test can hand the very masks of a dropout-on step to the CPU oracle
 *     (the masks are regenerated from (seed, step, call id, element) everywhere, never stored). */
int glass_dropout_scales_f32(const uint64_t* rng_state, uint64_t call_id, float p_drop, int64_t n_rows, int64_t C, float* out,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * K3n  embedding lookup + emb_gn + dropout through the embedding TABLE   (replaces the chain
 *      input_emb -> emb_gn -> dropout at impl/models.py:246-251 when the table is small)
 *     h0 = W[x] has only V distinct rows, so the whole-graph GraphNorm statistics are count-weighted sums
 *     over W (class_rowptr = int32[V+1], row lengths = nodes per table row, class_rowptr[V] = N: the row
 *     pointer of the selection CSR whose product on K1 is the embedding backward).
 *   fwd: saved[4H] = mean, rstd, scale, shift; table[V,H] (scratch) = W*scale + shift;
 *        out[n] = dropout(table[x[n]]); mask[n] as glass_embed_label_f32.  Two launches instead of
 *        gather + statistics + finalize + apply over [N,H].
 *   bwd: G[V,H] = sum over the nodes of each table row of the (dropout-masked) gradient of `out`
 *        (glass_spmm_csr_f32 with the selection CSR).  dW (+)= GraphNorm-and-gather backward,
 *        dgamma/dbeta/dalpha (+)=.  One launch instead of backward statistics + finalize + apply over [N,H].
 * ---------------------------------------------------------------------------------------- */
#define GLASS_EMBED_NORM_MAX_ROWS 8192
int glass_embed_norm_fwd_f32(const int64_t* x, const float* W, int64_t V, const int32_t* class_rowptr,
                             const float* gamma, const float* beta, const float* alpha, float eps, float* saved,
                             float* table, const int64_t* z, const int64_t* pos, int64_t n_pos, float p_drop,
                             const uint64_t* rng_state, uint64_t call_id, float* out, int64_t ldo, uint8_t* mask,
                             int64_t n_nodes, int64_t H, void* stream);
int glass_embed_norm_bwd_f32(const float* G, const float* W, int64_t V, const int32_t* class_rowptr,
                             const float* gamma, const float* alpha, const float* saved, float* dW, int accumulate_w,
                             float* dgamma, float* dbeta, float* dalpha, int accumulate, int64_t H, void* stream);

/*   bwd_adam: the step's last launch on the table path — (a) rows of the selection product G that K1 cut into several
 *        chunks are summed here from its partial rows (partials = the product's scratch, reduce_rows = the plan's reduce
 *        list on the device: (row, first slot, count) triples; call glass_spmm_csr_f32 with a header copy whose reduce
 *        count is 0), (b) glass_embed_norm_bwd_f32, (c) Adam (glass_adam_step_f32's update and step counter) over the whole
 *        arena param / grad / exp_avg / exp_avg_sq [n_param]: the table workgroups update exactly the elements whose
 *        gradients they produced (the table at arena offset off_W, emb_gn's weight / bias / mean_scale at off_gamma /
 *        off_beta / off_alpha — dW, dgamma, ... must be those views), the other workgroups the rest.  param == NULL: (a) +
 *        (b) only.  Three dependent launches (K1's reduce, the table backward, Adam) as one. */
int glass_embed_norm_bwd_adam_f32(float* G, const float* W, int64_t V, const int32_t* class_rowptr, const float* gamma,
                                  const float* alpha, const float* saved, float* dW, int accumulate_w, float* dgamma,
                                  float* dbeta, float* dalpha, int accumulate, int64_t H, const float* partials,
                                  const int32_t* reduce_rows, int64_t n_reduce, float* param, float* grad, float* exp_avg,
                                  float* exp_avg_sq, int64_t n_param, const float* lr_dev, double beta1, double beta2,
                                  double eps, double weight_decay, int64_t* step_dev, int64_t off_W, int64_t off_gamma,
                                  int64_t off_beta, int64_t off_alpha, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7  subgraph pooling   replaces pad2batch + emb[pos] + global_{add,mean,max}_pool /
 *     GraphSizeNorm (impl/models.py:346-350, 294-319; impl/utils.py:18-29)
 *     out[b,:] = reduce_{j: pos[b,j]>=0} emb[pos[b,j],:]   (sum | mean | max | sum * n_b^-1/2)
 *     argmax (int32 [B,C], only for max, may be NULL otherwise) records the contributing node.
 *     An all-padding row gives 0 (torch_scatter semantics).
 *     Backward scatters into demb (must be ZEROED by the caller).  sum / mean / size with B*Smax + B <= 12 288:
 *     ordered and atomic-free (pos staged in LDS; the first entry naming a node sums all its occurrences in
 *     (b, s) order) -> bitwise repeatable.  Larger batches and max pooling: glass_segment_pool_bwd_f32 returns
 *     GLASS_E_WS (it never switches to float atomics by itself) — call glass_segment_pool_bwd_exact_f32 /
 *     glass_segment_pool_max_bwd_exact_f32 below (workspace, exact fixed-point sums, bitwise repeatable), or, knowingly,
 *     glass_segment_pool_bwd_atomic_f32: the one-launch float-atomic scatter (any size, all four modes), whose result is
 *     within rounding of the exact forms but depends on the order of the atomics once three entries share a node.
 * ---------------------------------------------------------------------------------------- */
int glass_segment_pool_f32(const float* emb, int64_t lde, const int64_t* pos, int64_t B, int64_t Smax, int mode,
                           float* out, int64_t ldo, int32_t* argmax, int64_t n_nodes, int64_t C, void* stream);
int glass_segment_pool_bwd_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax, int mode,
                               const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes, int64_t C,
                               void* stream);
int glass_segment_pool_bwd_atomic_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax, int mode,
                                      const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes, int64_t C,
                                      void* stream);

/* Node pairs (Smax = 2; sum | mean | size): the link-prediction batches of the pre-training path — replaces
 * emb[subG_node] + torch.mean(emb, dim=1) (impl/models.py:497-498, 501-503; 131 072 pairs per step, GNNEmb.py).  Same
 * values as glass_segment_pool_f32 on a [B, 2] matrix (a -1 entry is padding).  The backward writes EVERY row of demb (no
 * zero-fill needed) and uses no float atomic: entries are bucketed by node (integer counts, a scan, integer cursors) and
 * a node's row is the sum of its entries' scaled gradient rows taken in exact fixed point (two 64-bit integers per
 * column), so the result does not depend on the order of the lists -> bitwise repeatable.  `ws`: glass_pair_pool_ws_bytes
 * bytes of scratch (need not be initialised). */
/* The same backward for a padded node matrix of ANY width (sum | mean | size): what glass_segment_pool_bwd_f32 computes,
 * without its LDS staging limit (12 288 entries) and without float atomics beyond it — five small launches instead of
 * one, every row of demb written, bitwise repeatable.  `ws`: glass_segment_pool_bwd_exact_ws_bytes bytes. */
int64_t glass_segment_pool_bwd_exact_ws_bytes(int64_t n_nodes, int64_t B, int64_t Smax);
int glass_segment_pool_bwd_exact_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax, int mode,
                                     float* demb, int64_t lde, int64_t n_nodes, int64_t C, void* ws, void* stream);
/* ... and for max pooling (argmax from glass_segment_pool_f32): node n receives, per column, the gradients of the subgraphs
 * whose maximum it is — summed exactly over the node's (per-row deduplicated) subgraph list instead of by float atomics.
 * `ws`: glass_segment_pool_bwd_exact_ws_bytes bytes. */
int glass_segment_pool_max_bwd_exact_f32(const float* dout, int64_t ldd, const int64_t* pos, int64_t B, int64_t Smax,
                                         const int32_t* argmax, float* demb, int64_t lde, int64_t n_nodes, int64_t C, void* ws,
                                         void* stream);
int64_t glass_pair_pool_ws_bytes(int64_t n_nodes, int64_t B);
int glass_pair_pool_f32(const float* emb, int64_t lde, const int64_t* pairs, int64_t B, int mode, float* out, int64_t ldo,
                        int64_t n_nodes, int64_t C, void* stream);
int glass_pair_pool_bwd_f32(const float* dout, int64_t ldd, const int64_t* pairs, int64_t B, int mode, float* demb,
                            int64_t lde, int64_t n_nodes, int64_t C, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9  link-prediction head of the pre-training path on node pairs, hidden 64 (glass_pair_head_supported)
 *     replaces EdgeGNN.Pool + MLP(hidden, hidden, 1, 2) + BCEWithLogitsLoss and their autograd
 *     (impl/models.py:497-509, 33-50; GNNEmb.py:94-99, 129-130, 144: 131 072 pairs per step):
 *         pooled[p] = (emb[pairs[p,0]] + emb[pairs[p,1]]) / 2
 *         hid[p]    = relu(dropout(pooled[p] W0^T + b0))           W0 [64,64] row-major, Linear -> Dropout -> ReLU
 *         logit[p]  = hid[p] . w1 + b1                             w1 [64], b1 [1]
 *         loss      = mean_p BCE-with-logits(logit[p], target[p])  target float [P]
 *     forward: ONE launch (pair gather inside the operand load, W0 on the fp32 matrix cores, the rest in the epilogue);
 *       writes hid [P,64] (row stride 64; its sign pattern is the ReLU / dropout mask), logits [P], dlogit [P] =
 *       grad_scale * (sigmoid(logit) - target) / P and per-workgroup loss terms into ws.  target == NULL: evaluation
 *       (logits only; hid, dlogit, ws may be NULL).  Dropout words: (seed, step) at rng_state, stream `call_id`.
 *     backward: dW0 / db0 / dw1 / db1 (accumulated when `accumulate`, else overwritten) from per-slab partial tiles summed in
 *       slab order, loss[0] = the mean loss (may be NULL), and demb [N,64] = d loss / d emb, EVERY row written: entries
 *       bucketed by node, exact fixed-point sums (no float atomic), then the 64 x 64 product with W0 — bitwise repeatable.
 *       Launches: weight partials, reduce, 4 for the buckets, gather = 7.  ws: glass_pair_head_ws_bytes, the SAME buffer in
 *       both calls (uninitialised scratch).  pairs entries must be valid node ids (an id outside [0, N) counts as a zero row).
 * ---------------------------------------------------------------------------------------- */
int glass_pair_head_supported(int64_t hidden);
int64_t glass_pair_head_ws_bytes(int64_t n_nodes, int64_t P);
int glass_pair_head_fwd_f32(const float* emb, int64_t lde, int64_t n_nodes, const int64_t* pairs, int64_t P, const float* W0,
                            const float* b0, const float* w1, const float* b1, const float* target, float p_drop,
                            const uint64_t* rng_state, uint64_t call_id, const float* grad_scale, float* hid, float* logits,
                            float* dlogit, void* ws, void* stream);
int glass_pair_head_bwd_f32(const float* emb, int64_t lde, int64_t n_nodes, const int64_t* pairs, int64_t P, const float* W0,
                            const float* w1, const float* hid, const float* dlogit, float p_drop, float* dW0, float* db0,
                            float* dw1, float* db1, int accumulate, float* loss, float* demb, int64_t ldde, void* ws,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * K5w weight / bias gradient of the stacked Linears   (autograd backward of nn.Linear at
 *     impl/models.py:158-159,169-170 — the measured dominant dense contraction of the step)
 *     dW[o,i] (+)= sum_n G[n,o] * X[n,i]      db[o] (+)= sum_n G[n,o]      (db may be NULL)
 *     Split over the node dimension on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), slab
 *     partials summed in fixed order (deterministic; exact fp32 fma-chain numerics).
 *     accumulate != 0 adds into dW / db (the flat gradient arena), else overwrites.
 *     Needs O%4==0, I%2==0, ldg%4==0, ldx%2==0, G 16-B and X 8-B aligned; otherwise returns
 *     GLASS_E_UNSUPPORTED and the caller uses a library GEMM.
 * ---------------------------------------------------------------------------------------- */
int64_t glass_linear_wgrad_ws_bytes(int64_t N, int64_t O, int64_t I);
int glass_linear_wgrad_f32(const float* G, int64_t ldg, const float* X, int64_t ldx, int64_t N, int64_t O, int64_t I,
                           float* dW, int64_t lddw, float* db, int accumulate, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  stacked Linear pair fused with the label-conditioned mix, on the fp32 matrix cores
 *     (impl/models.py:158-162 trans_fns + ELU + mix; 167-173 cat + comb_fns + mix).
 *     W = [W1; W0] ([2H, K] row-major, K = H or 2H), bias = [b1 | b0].
 *     The kernels consume the weight as a PACKED image produced by glass_dense_pack_batch_f32 once per step:
 *     Wimg from W itself (forward), WTimg from W with the transposed flag (data gradient).  Two kernel families,
 *     chosen by the hidden size: H = 64 — a wave owns 16 rows and all output columns, v_mfma_f32_16x16x4_f32,
 *     images in that fragment order (layout 0); H = 128 / 256 / 512 — LDS-tiled GEMM, workgroup tile 64 (H = 128) or
 *     128 rows x 256 columns, v_mfma_f32_32x32x2_f32, images in LDS-stage order (forward: layout 1 "paired", the f1 / f0 halves of
 *     a column side by side in a wave; data gradient: layout 2 "plain", or layout 3 "split" for the 128-wide output of
 *     H = 128's trans pair: the transposed [128][256] operand, both stacked halves side by side in one 256-slot tile over
 *     K = 128).  glass_dual_linear_layout(H) tells which family.
 *   fwd : xb == NULL (trans): T = xa @ W^T + bias is written to T (kept for the backward),
 *                             out = mix(act(T1), act(T0)).
 *         xb != NULL (comb) : the input is the virtual concatenation [xa || xb] (no cat copy),
 *                             out = mix(C1, C0); T may be NULL (C is never materialised).
 *   dgrad: out[N, n_out] = dZ @ W (+ addend), dZ[n,o] = coef(mask[n], o<H) * dsrc[n, o mod H] * act'(T[n,o])
 *          synthesised on the fly; WTimg = packed image of W^T ([n_out] x [2H]).
 *   wgrad: dW[2H, K] (+)= dZ^T @ [X || X2], db (+)= colsum(dZ), same synthesis, split-K MFMA as K5w.
 *   Hidden sizes 64, 128, 256, 512 (glass_dual_linear_supported); otherwise GLASS_E_UNSUPPORTED and the
 *   caller composes the library GEMM with glass_mix_*.
 * ---------------------------------------------------------------------------------------- */
/* ONE capability record per hidden width instead of the per-feature queries below (which stay, as thin wrappers of the same
 * answers): everything a caller needs to know to route a width — which kernel family serves the Linear pairs, which operand
 * image layouts its weights must be packed in, which fusions exist at that width.  glass_dense_caps fills *out and returns
 * 0, or GLASS_E_ARG for H <= 0 / out == NULL (out->family = 0 when no hand-written dense kernel serves the width:
 * the caller then uses library GEMMs + the stand-alone mix / GraphNorm kernels). */
typedef struct glass_dense_caps {
    int32_t family;            /* 0 none | 1 narrow (thread per row, hidden <= 32) | 2 staged 16x16x4 MFMA (hidden 64) | 3 LDS-tiled 32x32x2 MFMA (128 / 256 / 512) */
    int32_t weight_layout;     /* glass_dual_linear_layout: 0 wave16 images, 1 tiled images, 2 row-major weights as they are */
    int32_t fwd_layout_trans, fwd_layout_comb;       /* glass_dual_linear_fwd_layout(H, H) / (H, 2H) */
    int32_t dgrad_layout_trans, dgrad_layout_comb;   /* glass_dual_linear_dgrad_layout(H, H) / (H, 2H) */
    int32_t stat_rows;         /* rows per statistics partial of the forward (glass_dual_linear_stat_rows) */
    int32_t fwd_gather;        /* layer 0's trans kernel gathers its operand from the embedding table */
    int32_t gn_exact, gn_exact_fwd;                  /* exact GraphNorm accumulators: all sums / the forward sums alone */
    int32_t comb_eff, comb_eff_fwd;                  /* comb pair through effective per-label weights: fwd + bwd / fwd alone */
    int32_t comb_eff_fwd_layout, comb_eff_dgrad_layout2;
    int32_t pair_head;         /* K9 (pre-training head on node pairs) at this width */
    int32_t act_codes;         /* bit mask of the activation codes the family fuses: 1 << GLASS_ACT_ELU | 1 << GLASS_ACT_RELU */
    int32_t product_form;      /* how an fp32 product is formed on the matrix cores: 0 f32-input MFMA (an fmaf chain), 1 six bf16
                                  partial products of 3-way split operands — the family's DEFAULT; a call opts out with
                                  GLASS_DENSE_F32_PRODUCTS in its `act` word */
    int32_t serve_width;       /* the hidden width whose kernels serve H: H itself when family != 0; otherwise the next family
                                  width (64 / 128 / 256 / 512) — a model laid out zero-padded to it computes the width-H model
                                  exactly (padded columns stay 0 through every layer, padded parameters get zero gradients;
                                  glass_amd/widths.py) —; 0 when H > 512 */
} glass_dense_caps;
int glass_dense_caps_query(int64_t H, glass_dense_caps* out);
/* Product form of the LDS-tiled family (hidden 128 / 256 / 512), process-wide.  gfx950 has no tf32 / xf32 and its f32-input
 * MFMA runs at 1/16 of the bf16 rate, so by default (form 1) each operand is cut into three bf16 pieces — x = hi + mid + lo
 * EXACTLY for every fp32 x with |x| >= 2^-100 (each piece the RNE rounding of what the pieces before it left: 8 + 8 + 8 significant bits
 * plus the signs) — and x*w is the sum of the six partial products whose weight is >= 2^-18 (dropped: mid*lo, lo*mid, lo*lo
 * <= 3 * 2^-26 |x w|, below one fp32 rounding of the product), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Measured
 * against fp64 the result is as close as form 0's or closer (profiles/r04_split_product_accuracy.txt; the parity tests run
 * both).  A non-finite operand gives NaN where form 0 gives Inf or NaN.  Form 0 = v_mfma_f32_32x32x2_f32.  Also settable
 * before the first call with GLASS_DENSE_SPLIT=0|1 in the environment.  _set returns 0 or GLASS_E_ARG. */
/* (the product form is chosen PER CALL: GLASS_DENSE_F32_PRODUCTS in the `act` word; glass_dense_caps.product_form names the
 *  default of a width's family) */
int glass_dual_linear_supported(int64_t H);
/* 0: wave16 operand images (flags layout 0 for both operands); 1: tiled (forward operand layout 1, data-gradient
 * operand layout 2 — except a 128-wide output, hidden 128's trans pair, which keeps layout 0 and the wave16 kernel) */
int glass_dual_linear_layout(int64_t H);
/* layout code of the data-gradient operand image for (H, n_out): 0, 2, 3, 4, 9 or 10, and of the forward operand image for
 * (H, K = input width): 0, 1, 5 or 9 (see glass_dense_pack_batch_f32).  Hidden 128 (round 6, "stage-run" kernels that keep a
 * wave's weight slice in registers): forward of the trans pair 9, data gradient of the trans pair 9, of the comb pair 10 (its
 * 256-wide data gradient takes no activation, addend or dropout: GLASS_E_UNSUPPORTED otherwise). */
int glass_dual_linear_dgrad_layout(int64_t H, int64_t n_out);
int glass_dual_linear_fwd_layout(int64_t H, int64_t K);
/* rows covered by one workgroup of the fused kernels at hidden H = rows per `stats` / `gn_partial` entry */
int64_t glass_dual_linear_stat_rows(int64_t H);
/*   fwd, stats != NULL: the epilogue also writes the column statistics of `out` for the GraphNorm that consumes
 *   it — stats[ceil(n_nodes/R)][2][H] doubles, R = glass_dual_linear_stat_rows(H) (per workgroup: sum, sum of squares) — so that GraphNorm
 *   needs no statistics pass of its own: glass_graphnorm_finalize_f32 + glass_graphnorm_apply_f32.
 *   fwd, gn_saved != NULL: xa is the INPUT of a GraphNorm whose statistics are final (gn_saved[4H] from
 *   glass_graphnorm_finalize_f32 / _stats_f32); the kernel computes dropout(act(xa*scale + shift)) while loading
 *   (gn_act, p_drop, rng_state, call_id as in glass_graphnorm_fwd_f32), multiplies THAT, and writes it to xa_out
 *   [n_nodes, H] for the backward — no separate GraphNorm apply launch.
 *   fwd, xa_index != NULL (trans pair, with gn_saved; glass_dual_linear_fwd_gather_supported(H)): operand row n is row
 *   xa_index[n] of xa [xa_rows, H] — the embedding lookup `input_emb(x)` (impl/models.py:248) happens in the operand load:
 *   xa = the embedding table, gn_saved = emb_gn's statistics through the table (glass_step_prologue_f32), xa_out receives
 *   dropout(emb_gn(input_emb(x))) [n_nodes, H], the layer input — no gather launch. */
int glass_dual_linear_fwd_gather_supported(int64_t H);
/*   stats_exact != 0 / gn_src != NULL (hidden 64): the exact-accumulator forms of `stats` and `gn_saved`, see
 *   glass_gn_src below. */
int glass_dual_linear_fwd_f32(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg,
                              const float* bias, const uint8_t* mask, double z_ratio, int act, float* T, int64_t ldt,
                              float* out, int64_t ldo, int64_t n_nodes, int64_t H, double* stats, int stats_exact,
                              const float* gn_saved, const glass_gn_src* gn_src, int gn_act, float p_drop,
                              const uint64_t* rng_state, uint64_t call_id, float* xa_out, int64_t ldxo,
                              const int64_t* xa_index, int64_t xa_rows, void* stream);
/*   dgrad epilogue: out = (dZ @ W + addend) * dropmask(p_drop, rng_state, call_id) — the mask of the dropout that
 *   produced this layer's input (same mask layout as glass_graphnorm_fwd_f32), so the consumer receives the
 *   gradient w.r.t. the pre-dropout tensor; p_drop = 0 disables it (rng_state may be NULL).
 *   dgrad, gn_partial != NULL: the first H output columns are the gradient dy of a GraphNorm OUTPUT (input gn_x,
 *   forward statistics gn_saved, mean scale gn_alpha, activation / dropout gn_act, gn_p_drop, gn_call_id of that
 *   GraphNorm); the epilogue accumulates its two backward column sums into gn_partial[ceil(n_nodes/R)][2][H]
 *   doubles, for glass_graphnorm_bwd_from_stats_f32 — no backward statistics launch. */
int glass_dual_linear_dgrad_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask,
                                double z_ratio, int act, const float* WTimg, int64_t n_out, const float* addend,
                                int64_t ldadd, float p_drop, const uint64_t* rng_state, uint64_t call_id, float* out,
                                int64_t ldo, int64_t n_nodes, int64_t H, double* gn_partial, const float* gn_x,
                                int64_t gn_ldx, const float* gn_saved, const float* gn_alpha, int gn_act, float gn_p_drop,
                                uint64_t gn_call_id, int gn_exact, void* stream);
/*   bwd: the whole backward of one pair for the step program — exactly glass_dual_linear_dgrad_f32 (same arguments)
 *   followed by glass_dual_linear_wgrad_f32 with dW == NULL (partial sums of dW / db into `ws`, reduced later by
 *   glass_linear_wgrad_reduce_batch_f32; X / X2 = the pair's inputs).  At hidden 64 on graphs of up to 100 000 nodes
 *   the two are independent, latency-bound 12-17 us kernels: there they run as two branches of ONE launch (their
 *   workgroups share the CUs); otherwise as the two launches.
 *   wgrad at hidden >= 256 on n_nodes >= 65 536 (the LDS-tiled kernels): X / X2 16-byte aligned with ld % 4 == 0 and
 *   `mask` 2-byte aligned, else GLASS_E_UNSUPPORTED. */
int glass_dual_linear_bwd_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask,
                              double z_ratio, int act, const float* WTimg, int64_t n_out, const float* addend,
                              int64_t ldadd, float p_drop, const uint64_t* rng_state, uint64_t call_id, float* out,
                              int64_t ldo, int64_t n_nodes, int64_t H, double* gn_partial, const float* gn_x,
                              int64_t gn_ldx, const float* gn_saved, const float* gn_alpha, int gn_act, float gn_p_drop,
                              uint64_t gn_call_id, int gn_exact, const float* X, int64_t ldx, const float* X2, int64_t ldx2,
                              void* ws, void* stream);
int glass_dual_linear_wgrad_f32(const float* dsrc, int64_t ldd, const float* T, int64_t ldt, const uint8_t* mask,
                                double z_ratio, int act, const float* X, int64_t ldx, const float* X2, int64_t ldx2,
                                int64_t N, int64_t H, float* dW, int64_t lddw, float* db, int accumulate, void* ws,
                                void* stream);
/*   The comb pair in EFFECTIVE-WEIGHT form (hidden 64; impl/models.py:169-173).  No activation sits between the comb
 *   pair's Linear layers and the label mix, so  y[r] = [g || x_][r] . (w1(r) W1 + w0(r) W0)^T + (w1 b1 + w0 b0): one product
 *   per row with one of two effective weights.  Every row tile multiplies the unlabeled-row weight (1-z) W1 + z W0 and
 *   skips the store of its labeled rows; the unique labeled rows of the batch (lab_rows / lab_count from
 *   glass_batch_labels; lab_cap = capacity of the list, fixes the grid) are gathered 16 per wave by extra workgroups of
 *   the same launch, which multiply z W1 + (1-z) W0 — every row written once, no atomics; half the matrix work of
 *   glass_dual_linear_fwd_f32 / _bwd_f32 on the same pair.  Wimg_eff / WTimg_eff: operand images of layout 6 / 7
 *   (glass_dense_pack_batch_f32).  stats holds glass_comb_eff_fwd_blocks(n_nodes, H, lab_cap) and gn_partial
 *   glass_comb_eff_blocks(n_nodes, H, lab_cap) entries of [2][H] doubles (row tiles, then extra workgroups).  Other arguments as in glass_dual_linear_fwd_f32 (xb != NULL, act none,
 *   no T) and glass_dual_linear_bwd_f32 (n_out = 2H, no addend, no dropout on the output; X == NULL: data gradient only).
 *   The weight-gradient partials written to `ws` are in S / L form: one [H x 2H] product over all rows plus the same over the
 *   listed rows (half the matrix work of the plain form); reduce them with glass_linear_wgrad_reduce_batch_f32, lab_cap[j]
 *   = this call's lab_cap. */
int glass_comb_eff_supported(int64_t H);
int glass_comb_eff_fwd_supported(int64_t H); /* the forward alone: also hidden 128 */
int glass_comb_eff_fwd_layout(int64_t H); /* pack layout of Wimg_eff for glass_comb_eff_fwd_f32: 6 or 8 */
int glass_comb_eff_dgrad_layout2(int64_t H); /* != 0: WTimg_eff holds a second pair of images in this layout behind the layout-7 pair */
int64_t glass_comb_eff_max_rows(int64_t ld); /* most rows the forward serves at operand row strides <= ld floats */
int64_t glass_comb_eff_blocks(int64_t n_nodes, int64_t H, int64_t lab_cap);     /* entries of the backward's gn_partial */
int64_t glass_comb_eff_fwd_blocks(int64_t n_nodes, int64_t H, int64_t lab_cap); /* entries of the forward's stats (partials form): one per workgroup of the launch's geometry — tall row tiles (one round of the chip) on graphs beyond 80 x 256 rows */
int64_t glass_comb_eff_ws_bytes(int64_t n_nodes, int64_t H, int64_t lab_cap); /* `ws` of glass_comb_eff_bwd_f32 */
int glass_comb_eff_fwd_f32(const float* xa, int64_t lda, const float* xb, int64_t ldb, const float* Wimg_eff,
                           const float* bias, const uint8_t* mask, double z_ratio, float* out, int64_t ldo,
                           int64_t n_nodes, int64_t H, double* stats, int stats_exact, const float* gn_saved,
                           const glass_gn_src* gn_src, int gn_act, float p_drop, const uint64_t* rng_state,
                           uint64_t call_id, float* xa_out, int64_t ldxo, const int32_t* lab_rows,
                           const int32_t* lab_count, int64_t lab_cap, void* stream);
/*   dsrc_gn != NULL (hidden 64, small graphs: glass_comb_eff_bwd_gn_src_supported): dsrc is NOT materialised — it is the
 *   input gradient of the GraphNorm between two layers (gns[l], impl/models.py:257-259 backward) and the kernels derive
 *   it while loading their rows: dc = A * g + Bx * x + K (+ addend), g = dy * dropmask * act'(x * scale + shift), with the
 *   coefficients from that GraphNorm's two backward sums in exact accumulators (what glass_graphnorm_bwd_from_stats_f32
 *   with nblk = -n_rep would apply in a launch of its own); workgroup 0 writes dgamma / dbeta / dalpha. */
struct glass_gn_bwd_src {
    const int64_t* acc;
    int64_t n_rep;
    const float* dy;
    int64_t lddy;
    const float* x;
    int64_t ldx;
    const float* addend; /* may be NULL */
    int64_t ldadd;
    const float *saved, *gamma, *alpha;
    float *dgamma, *dbeta, *dalpha;
    int accumulate, act;
    float p_drop;
    uint64_t call_id;
};
int glass_comb_eff_bwd_gn_src_supported(int64_t n_nodes, int64_t H);
int glass_comb_eff_bwd_f32(const float* dsrc, int64_t ldd, const uint8_t* mask, double z_ratio, const float* WTimg_eff,
                           float* out, int64_t ldo, int64_t n_nodes, int64_t H, double* gn_partial, const float* gn_x,
                           int64_t gn_ldx, const float* gn_saved, const float* gn_alpha, int gn_act, float gn_p_drop,
                           const uint64_t* rng_state, uint64_t gn_call_id, int gn_exact, const float* X, int64_t ldx,
                           const float* X2, int64_t ldx2, void* ws, const int32_t* lab_rows, const int32_t* lab_count,
                           int64_t lab_cap, const glass_gn_bwd_src* dsrc_gn, void* stream);
/*     Deferred reduction: glass_dual_linear_wgrad_f32 with dW == NULL only writes the per-slab partial sums
 *     into `ws` (one scratch buffer per pending gradient); this call then reduces n_jobs of them — job j is
 *     the gradient of a [O[j], I[j]] weight over N[j] rows — into dW[j] / db[j] (db[j] may be NULL) with ONE
 *     launch.  All array arguments are HOST arrays.  lab_cap (may be NULL): lab_cap[j] > 0 marks job j as the partials
 *     of glass_comb_eff_bwd_f32 called with that list capacity (S / L form: dW1 = (1-z) S + (2z-1) L, dW0 = z S - (2z-1) L
 *     with S the all-rows and L the labeled-rows sum of dc^T [g || x_]). */
int glass_linear_wgrad_reduce_batch_f32(int64_t n_jobs, const void* const* ws, const int64_t* N, const int64_t* O,
                                        const int64_t* I, float* const* dW, const int64_t* lddw, float* const* db,
                                        const int32_t* accumulate, const int64_t* lab_cap, void* stream);
/*     The same reduction and a K1 product Y = M @ X as ONE launch when M's plan holds workgroup items only (the one-hot
 *     selection matrix of the embedding backward, glass_embed_label_f32's note) — both only wait for the end of the
 *     backward chain; otherwise the two calls in sequence.  Arguments: those of the reduction, then those of
 *     glass_spmm_csr_f32.  glass_spmm_reduce_rows_f32: the reduce step of a K1 plan on its own (partial rows -> Y). */
int glass_wgrad_reduce_spmm_f32(int64_t n_jobs, const void* const* ws, const int64_t* N, const int64_t* O, const int64_t* I,
                                float* const* dW, const int64_t* lddw, float* const* db, const int32_t* accumulate,
                                const int64_t* lab_cap, const int32_t* rowptr, const int32_t* col, const float* val,
                                const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t n_rows, int64_t H,
                                const int32_t* plan_header_host, const int32_t* plan_dev, void* ws_spmm, void* stream);
int glass_spmm_reduce_rows_f32(const float* partials, float* Y, int64_t ldy, int64_t H, const int32_t* reduce_rows_dev,
                               int64_t n_reduce, void* stream);
/*     Pack up to 16 weight operands B[NT][KT] (NT, KT multiples of 64) into MFMA image order in one launch.
 *     flags[k] bit 0 = transposed: 0: B = src[k] ([NT][KT] row-major); 1: B[n][k] = src[k][k][n] (src is [KT][NT]).
 *     flags[k] >> 1 = layout: 0 wave16, 1 tiled paired (NT = 2H), 2 tiled plain (tiled: NT a multiple of 256), 3 tiled
 *     split (the transposed 128 x 256 operand), 4 tiled plain followed by the effective weight of unlabeled rows
 *     (1 - z_ratio[k]) * B[:, :KT/2] + z_ratio[k] * B[:, KT/2:] in the same tiling over K = KT/2 (dst[k] then holds
 *     1.5 * NT*KT floats; transposed operands only) — glass_dual_linear_dgrad_layout(H, n_out) names the layout the
 *     data-gradient kernels read; 5 tiled paired followed by (1 - z) * B[:NT/2] + z * B[NT/2:] in the plain tiling
 *     (forward operand of a comb pair, again 1.5 * NT*KT floats; glass_dual_linear_fwd_layout); 6 (not transposed, NT = 2H
 *     stacked outputs) / 7 (transposed, KT = 2H): the two wave16 images of the comb pair's effective weights, unlabeled rows
 *     (1-z) * f1 half + z * f0 half, then labeled rows z * f1 half + (1-z) * f0 half, NT*KT floats in all — the operands of
 *     glass_comb_eff_fwd_f32 / _bwd_f32.  Layouts 8 / 10 / 9 are 6 / 7 / 0 with another order of the output columns inside a
 *     tile — tile t holds columns 64 (t >> 2) + 16 (t & 3) .. + 15 — for the "staged" hidden-64 kernels, where a wave owns 16
 *     consecutive output columns and keeps its slice of the operand in registers: 8 = forward image of the comb pair
 *     (glass_comb_eff_fwd_layout), 10 = its data-gradient images, packed BEHIND the layout-7 pair in the same buffer
 *     (glass_comb_eff_dgrad_layout2; alone, [256][256] transposed, as the comb pair's data-gradient operand at hidden 128:
 *     glass_dual_linear_dgrad_layout(128, 256)), 9 = both operands of the trans pair (glass_dual_linear_fwd_layout(H, H),
 *     glass_dual_linear_dgrad_layout(H, H), H = 64 or 128).  z_ratio (may be NULL when no job has layout 4 - 8, 10): per-job label mix of
 *     the pair.  dst[k] holds NT*KT floats otherwise.  The pointer / size arrays are HOST arrays.
 *     The staged kernels address their row operands through buffer resources (32-bit offsets): rows * ld * 4 < 2^31,
 *     checked per call; glass_comb_eff_max_rows(ld) tells the limit. */
/*     rng_state (may be NULL): the same launch also advances the dropout stream, rng_state[1] += 1 (both are
 *     once-per-step prologue work; equivalent to a following glass_rng_advance). */
/* Floats the image buffer dst[j] of a pack job must hold: NT*KT; + half of that for the effective-weight appendix of layouts
 * 4 / 5; and for the tiled layouts (1..5) 3/2 of the sum again behind it — the same image cut into three bf16 pieces per
 * element in the order the LDS-tiled kernels copy it to LDS, always written by the pack kernel and read by the kernels in
 * product form 1 (a call with GLASS_DENSE_F32_PRODUCTS reads the fp32 image: no re-pack between the forms).  Host arithmetic; GLASS_E_ARG for NT, KT <= 0. */
int64_t glass_dense_image_floats(int64_t NT, int64_t KT, int32_t flags);
/* dst_floats[k] = floats dst[k] can hold: a job whose image (glass_dense_image_floats) does not fit is refused with
 * GLASS_E_ARG before anything is launched — the image sizes differ by layout, so the callee never assumes one. */
int glass_dense_pack_batch_f32(const float* const* src, float* const* dst, const int64_t* dst_floats, const int64_t* NT,
                               const int64_t* KT, const int32_t* flags, const float* z_ratio, int64_t n_jobs,
                               uint64_t* rng_state, void* stream);
/*     The once-per-step prologue as ONE launch: glass_dense_pack_batch_f32 (same first nine arguments; n_jobs may be 0)
 *     plus the statistics of emb_gn through the embedding table — the first half of glass_embed_norm_fwd_f32
 *     (impl/models.py:248-249): saved[4H] = mean, rstd, scale, shift of GraphNorm(W[x]) as count-weighted sums over the V
 *     table rows (class_rowptr as there); table[V,H] = W*scale + shift, or NULL when the consumer normalises while
 *     gathering from W (glass_dual_linear_fwd_f32 with xa_index).
 *     W == NULL: no table job.  zero_words / n_zero_words: int64 words the same launch zero-fills — the step's exact
 *     GraphNorm accumulators (below). */
int glass_step_prologue_f32(const float* const* src, float* const* dst, const int64_t* dst_floats, const int64_t* NT,
                            const int64_t* KT, const int32_t* flags, const float* z_ratio, int64_t n_jobs, uint64_t* rng_state,
                            const float* W, int64_t V, const int32_t* class_rowptr, const float* gamma, const float* beta,
                            const float* alpha, float eps, float* saved, float* table, int64_t H, int64_t* zero_words,
                            int64_t n_zero_words, void* stream);
/*     The head of a REPLAYED training step as ONE launch: glass_step_prologue_f32 (same first 21 arguments) and, beside it in
 *     the same grid, the label launch glass_batch_labels_gather for the step's batch (impl/utils.py:32-45,
 *     impl/SubGDataset.py:69-72, 92-96) — the two depend on nothing but the parameters and the batch, so the shorter one
 *     (~5 us at ppi_bp-shape) disappears from the step's chain.  Inside a captured graph no launch argument can follow the
 *     epoch's shuffle, so the batch is named by a DEVICE-RESIDENT cursor: `cur` points to a glass_batch_cursor in device
 *     memory that the caller fills once per epoch — the data set's matrices, the epoch's index batches idx[n_batches][n_idx]
 *     (ZGDataloader's permutation cut into batches; a data-parallel rank stores its own slices) and cursor = 0; every launch
 *     takes batch min(cursor, n_batches - 1) (wrap != 0: cursor % n_batches) and advances the cursor by one.  n_idx, smax, y_row_bytes are launch arguments
 *     (the batch shape is fixed for a captured step).  pos_dst .. ws, n_nodes as in glass_batch_labels_gather (incremental
 *     form: pos_dst holds the previous batch). */
typedef struct glass_batch_cursor {
    const int64_t* pos_all; /* [n_all, smax] padded node matrix of the data set */
    const void* y_all;      /* [n_all, y_row_bytes] targets (NULL: none) */
    const int64_t* idx;     /* [n_batches, n_idx] rows of every batch of the epoch */
    int64_t n_all;
    int64_t n_batches;
    int64_t cursor;         /* next batch; advanced by the launch */
    int64_t wrap;           /* != 0: batch cursor % n_batches (a loop over the same batches) instead of the clamp */
} glass_batch_cursor;
int glass_step_head_f32(const float* const* src, float* const* dst, const int64_t* dst_floats, const int64_t* NT,
                        const int64_t* KT, const int32_t* flags, const float* z_ratio, int64_t n_jobs, uint64_t* rng_state,
                        const float* W, int64_t V, const int32_t* class_rowptr, const float* gamma, const float* beta,
                        const float* alpha, float eps, float* saved, float* table, int64_t H, int64_t* zero_words,
                        int64_t n_zero_words, glass_batch_cursor* cur, int64_t n_idx, int64_t smax, int64_t y_row_bytes,
                        int64_t* pos_dst, void* y_dst, uint8_t* mask, int32_t* lab_rows, int32_t* lab_count, void* ws,
                        int64_t n_nodes, void* stream);
/*     One-shot peer all-reduce of the gradient arena fused with Adam (the small bucket of the data-parallel exchange,
 *     SURVEY.md 8e; replaces `all_reduce(grads) / world` + `Adam.step()` of the replicas: impl/train.py:10-16, GLASSTest.py:213).
 *     xGMI is point to point, so every rank READS every peer's arena directly instead of walking a ring: one launch per
 *     rank and step — publish "my gradients of step s are final" (grp->flags[rank][0] = s, system-scope release), wait for every
 *     peer's flag (bounded spin), mean gradient = (g_0 + g_1 + ... + g_{N-1}) / N summed in RANK order (the same bits on every
 *     rank), Adam on this rank's replica (the arithmetic of glass_adam_step_f32), publish "done reading" (flags[rank][1] = s)
 *     and wait for the peers' done flags before the launch ends (a rank's next backward may then overwrite its arena).
 *     grp: every rank's arena and flag block (2 x uint64, zero before the first call) as THIS process addresses them — the
 *     own ones plain, the peers' through hipIpc mappings (glass_peer_export / _import) or peer access inside one process.
 *     seq_dev uint64[2] = {last finished exchange, ticket} (zero before the first call; every rank advances in lock step);
 *     status_dev int32: set non-zero (sticky) when a flag did not arrive within spin_limit polls — the launch then ends
 *     without touching parameters, moments or step_dev: an error for the host to raise, never a hang.  mean_out (may be
 *     NULL) receives the mean gradient.  No allocation, no host synchronisation: capturable in the step's graph.
 *     glass_peer_alloc / _free / _export / _import / _close are SET-UP helpers (the only entries that allocate): an arena
 *     shared over hipIpc must be a whole runtime allocation, not a slice of a framework's pooled block. */
typedef struct glass_peer_group {
    int32_t world, rank;
    const float* grad[8];
    uint64_t* flags[8];
} glass_peer_group;
int glass_peer_allreduce_adam_f32(const glass_peer_group* grp, int64_t n, float* param, float* exp_avg, float* exp_avg_sq,
                                  const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                                  int64_t* step_dev, uint64_t* seq_dev, int32_t* status_dev, int64_t spin_limit,
                                  float* mean_out, void* stream);
int glass_peer_alloc(int64_t bytes, void** out);
int glass_peer_free(void* p);
int glass_peer_export(void* p, void* handle64);   /* handle64: 64 bytes (hipIpcMemHandle_t) */
int glass_peer_import(const void* handle64, void** out);
int glass_peer_close(void* p);
/*     Exact cross-workgroup GraphNorm sums (hidden 64): the whole-graph GraphNorm (PyG GraphNorm with batch = None,
 *     impl/models.py:165,249,257,266,271) needs column sums over ALL rows between every pair of kernels of the step.
 *     Instead of per-workgroup fp64 partials + a finalize launch, the producers add their per-workgroup sums into
 *     FIXED-POINT accumulators with 64-bit integer atomics: int64 [replicas][2][C][2 limbs] = glass_gn_exact_words(C)
 *     words per GraphNorm (column block), zeroed per step by glass_step_prologue_f32.  Integer addition is
 *     associative, so the sums do not depend on the order the workgroups arrive in — bitwise repeatable (float atomics
 *     are not).  Forward sums: resolution 2^-52 per workgroup partial, range 2^50; backward sums: 2^-64, range 2^38.
 *     The consumers fold the replicas in every workgroup's prologue and derive the coefficients themselves:
 *     stats_exact / gn_exact / n_rep = the number of replicas (2, 4, 8 or 16; 0 = the partials form) a producer spreads
 *     its adds over — workgroup b adds to replica b % n_rep — and its consumer folds: adds to ONE address queue at the
 *     memory-side atomic unit, so producers whose workgroups all finish together want many, while every consumer
 *     workgroup reads n_rep * 2 KB (hidden 64).
 *       backward (gn_exact = n_rep on the data-gradient entries): glass_graphnorm_bwd_from_stats_f32 with nblk = -n_rep
 *         and partial = the accumulators — finalize + apply as ONE launch;
 *       forward (stats_exact = n_rep: `stats` points to accumulators; glass_graphnorm_stats_exact_f32 for a stand-alone
 *         statistics pass): the kernel that applies the GraphNorm receives a glass_gn_src — the accumulators and the
 *         GraphNorm's parameters — instead of final statistics, and its workgroup 0 writes gn_saved[4C] (mean, rstd,
 *         scale, shift) for the backward.  n_src accumulator blocks of C / n_src columns each (the column blocks of a
 *         jumping-knowledge buffer, impl/models.py:268-270); the dense kernels take n_src = 1. */
struct glass_gn_src {
    const int64_t* acc;
    int64_t n_src;
    int64_t n_rep; /* replicas the producers spread their adds over: the value they were given as stats_exact */
    const float *gamma, *beta, *alpha;
    float eps;
};
int glass_gn_exact_supported(int64_t H);
int glass_gn_exact_fwd_supported(int64_t H); /* the forward sums alone (glass_graphnorm_stats_exact_f32 -> glass_comb_eff_fwd_f32 ->
                                                 glass_readout_train_f32): also hidden 128 */
int64_t glass_gn_exact_words(int64_t C);
int glass_graphnorm_stats_exact_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t C, int64_t* acc, int n_rep,
                                    void* stream);

/* K8  prediction head + loss (the bare nn.Linear head of GLASSTest.py:159-160 followed by
 *     CrossEntropyLoss, GLASSTest.py:69, mode 0, target int64[B]; or BCEWithLogitsLoss on the flattened
 *     logits, GLASSTest.py:57-58, mode 1, target float[B,K]; mean reduction) in one launch, and their
 *     whole backward in one launch.  fwd writes logits [B,K], prob (float[B*K + B]: softmax / sigmoid
 *     kept for the backward, then B per-subgraph loss terms) and loss[0] (their mean, summed in order).  bwd: dlogits = grad_loss[0] * (prob - target) / (B or B*K);
 *     dpooled = dlogits @ W; dW (+)= dlogits^T @ pooled; db (+)= colsum(dlogits); fixed summation order. */
int glass_head_loss_fwd_f32(const float* pooled, int64_t ldp, const float* W, const float* bias, const void* target,
                            int mode, int64_t B, int64_t C, int64_t K, float* logits, float* prob, float* loss,
                            void* stream);
int glass_head_loss_bwd_f32(const float* pooled, int64_t ldp, const float* W, const float* prob, const void* target,
                            int mode, const float* grad_loss, int64_t B, int64_t C, int64_t K, float* dpooled,
                            int64_t lddp, float* dW, float* db, int accumulate, void* stream);

/*     The head alone, for evaluation (no target): logits[B,K] = pooled[B,C] @ W[K,C]^T + bias (bias may be NULL). */
int glass_head_linear_f32(const float* pooled, int64_t ldp, const float* W, const float* bias, int64_t B, int64_t C, int64_t K,
                          float* logits, int64_t ldl, void* stream);

/* K8r  training-step readout: final GraphNorm apply -> subgraph pooling -> Linear head -> loss AND the whole
 *      backward down to the gradient of the GraphNorm INPUT, in four launches (impl/models.py:266/271, 346-350;
 *      GLASSTest.py:159-160, 57-58/69).  Only pooled rows carry a gradient into the GraphNorm output, so its two
 *      backward column sums are accumulated per subgraph; the [N,C] normalised embedding and its gradient are
 *      never materialised.  jk = GraphNorm input [N,C] (the jumping-knowledge buffer); gn_saved from
 *      glass_graphnorm_stats_f32 (statistics + finalize without the apply pass); pos [B,Smax] (-1 padding);
 *      pool_mode sum|mean|size; loss_mode 0 = cross-entropy (target int64[B]), 1 = BCE-with-logits (float[B,K]);
 *      grad_loss = device scalar seed.  Outputs: pooled [B,C], logits [B,K], loss [1], djk [N,C] (overwritten);
 *      dWh/dbh and dgamma/dbeta/dalpha are accumulated when acc_* != 0.
 *      Bitwise repeatable: B*Smax <= 16 384 by an ordered, atomic-free scatter of the sparse part staged in LDS; beyond that
 *      by node-bucketed exact sums when scatter_ws (glass_readout_scatter_ws_bytes bytes, uninitialised scratch) is given —
 *      with scatter_ws == NULL such a batch is refused (GLASS_E_WS): no float atomic is reachable from this entry point.  mask / lab_rows / lab_count (all three, or NULL): the label bytes and the unique
 *      labeled rows glass_batch_labels produced for THIS pos — the pooled rows of a step are its labeled rows — then the
 *      dense part skips them and extra workgroups of the same launch write their full value: three launches, bitwise
 *      equal to the four.  C % 4 != 0 (or unaligned rows): scalar kernels, which need the label bytes (mask). */
int glass_graphnorm_stats_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t C, const float* gamma,
                              const float* beta, const float* alpha, float eps, float* saved, void* ws, void* stream);
int glass_readout_supported(int64_t C, int64_t K, int pool_mode);
int64_t glass_readout_ws_bytes(int64_t B, int64_t C, int64_t K);
int64_t glass_readout_scatter_ws_bytes(int64_t n_nodes, int64_t B, int64_t Smax); /* 0 while B*Smax <= 16 384 */
int glass_readout_train_f32(const float* jk, int64_t ldj, const float* gn_saved, const float* gamma, const float* alpha,
                            const int64_t* pos, int64_t B, int64_t Smax, int pool_mode, const float* Wh, const float* bh,
                            const void* target, int loss_mode, int64_t K, const float* grad_loss, float* pooled,
                            float* logits, float* loss, float* djk, int64_t lddj, float* dWh, float* dbh, int acc_head,
                            float* dgamma, float* dbeta, float* dalpha, int acc_gn, void* ws, int64_t n_nodes, int64_t C,
                            const uint8_t* mask, const int32_t* lab_rows, const int32_t* lab_count,
                            const glass_gn_src* gn_src, int64_t* gn_bwd_acc, int gn_bwd_rep, void* scatter_ws,
                            float* loss_sum, void* stream);
/*      loss_sum (nullable): the thread that stores the step's mean loss also adds it to loss_sum[0] — the epoch's running
 *      sum of impl/train.py:15-17 (`total_loss.append(loss.item())` ... `np.average`) kept on the device in step order, so
 *      the epoch loop needs neither a host sync nor an extra launch per step.
 *      gn_bwd_acc != NULL (with the listed pooled rows): two launches instead of three — the subgraph kernel adds its share of
 *      the final GraphNorm's two backward column sums to these exact accumulators (glass_gn_exact_words(C) words, zeroed by
 *      the caller per step, gn_bwd_rep replicas in use), the backfill launch folds them and also carries the head-gradient
 *      rows and the mean loss. */

/*     Two small device-to-device copies in one launch (4-byte granularity): a training step that is replayed from a
 *     captured graph reads its batch (pos, target) from fixed buffers; this fills both per step. */
int glass_copy_pair(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1, int64_t bytes1,
                    void* stream);

/* K9  Adam over a flat parameter arena (torch.optim.Adam as used at GLASSTest.py:213; amsgrad
 *     off): one launch for all parameters.  lr and the step counter live in DEVICE memory so a
 *     captured graph follows ReduceLROnPlateau and advances its own bias correction.
 *     step_dev = int64[2]: [0] = steps completed (the launch uses [0]+1 and stores it back), [1] = scratch
 *     ticket counter, zero between launches. */
int glass_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                        int64_t* step_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLASS_HIP_H */
