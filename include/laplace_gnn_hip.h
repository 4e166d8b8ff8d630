/*
 * laplace_gnn_hip.h -- C ABI of the MI355X (gfx950) curvature-accumulation engine.
 *
 * This is the drop-in boundary for ONE path of anitasyang/Laplace-GNN: what
 * `laplace.curvature.CurvlinopsGGN.kron / .diag / .full` do for a GCN / GraphSAGE model
 * inside `Laplace(...).fit()`.  The reference is pure Python (no native code), so nothing
 * in it binds a C library today; each entry point below names the reference interface it
 * replaces (file:line under the reference tree).  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, one GPU per process) unless marked "host";
 *     tensors are row-major, fp32 / int64 exactly as PyTorch hands them over;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued asynchronously on it.  The calls that DO synchronise `stream` -- a
 *     count has to reach the host before the next launch or allocation can be sized -- are,
 *     completely:
 *       lgnn_create, lgnn_adj_to_edge_index     the deduplicated nnz / the off-diagonal count;
 *       lgnn_check_async_errors                 by design (reports the sticky error flags);
 *       the FIRST KFAC / adjacency-gradient call on a graph, once per graph: the list of rows with
 *         more than 64 stored entries (hubs) and the graph's number of 2-hop paths;
 *       lgnn_glm_variance(_mapped), once per call: the size of the rotated-row table;
 *       lgnn_update_adjacency: the new number of stored entries.
 *     Everything else -- KFAC of GCN / GraphSAGE models of any depth, diagonal and last-layer GGN, forward,
 *     Jacobians -- enqueue only; workspaces grow on first use of a shape and are reused after;
 *   - every function returns 0 on success, non-zero on error; the message is available
 *     from lgnn_last_error() (thread-local).  No C++ exception crosses the boundary;
 *   - borrowed pointers (weights, features, indices, outputs) are owned by the caller
 *     and must stay valid until the stream work that uses them has finished;
 *   - accumulate-style calls ADD into caller-owned fp32 buffers, so one all-reduce of
 *     those buffers is all a data-parallel caller needs.
 */
#ifndef LAPLACE_GNN_HIP_H
#define LAPLACE_GNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGNN_ABI_VERSION 1

#if defined(__GNUC__)
#define LGNN_API __attribute__((visibility("default")))
#else
#define LGNN_API
#endif

/* graph kinds: which propagation matrix the convolutions use */
#define LGNN_KIND_GCN 0  /* D^-1/2 A^T D^-1/2, self loops added  (gnn/models/models.py:23-31, utils.py:106-112) */
#define LGNN_KIND_SAGE 1 /* A / max(rowsum,1), self loops removed (gnn/models/models.py:47, layers.py:18-24)    */

/* activations between layers (gnn/models/base_gnn.py:85, default "relu") */
#define LGNN_ACT_RELU 0
#define LGNN_ACT_TANH 1

/* normalisation between a hidden layer's convolution and its activation (gnn/models/base_gnn.py:86-95, 148) */
#define LGNN_NORM_NONE 0  /* nn.Identity                                        */
#define LGNN_NORM_LAYER 1 /* nn.LayerNorm(hidden): per-row statistics           */
#define LGNN_NORM_BATCH 2 /* nn.BatchNorm1d(hidden) in eval mode: running stats */

/* likelihoods (laplace/curvature/curvature.py:63-72) */
#define LGNN_LIK_CLASSIFICATION 0 /* CrossEntropyLoss(sum), factor 1   */
#define LGNN_LIK_REGRESSION 1     /* MSELoss(sum), factor 0.5          */

/* flags of lgnn_kfac_accumulate */
#define LGNN_FLAG_FORK_EXACT_SEED 1u /* back-propagate d/df sum_i f_i S_ic(f) (curvlinops/kfac.py:637-661 as
                                        modified by the fork) instead of upstream's detached S[:,c]          */
#define LGNN_FLAG_NO_FUSE 2u         /* debugging: run SpMM^T and the Gram contraction as separate kernels    */
#define LGNN_FLAG_NO_PATHS 4u        /* 2-layer GCN: keep the class-plane route (backward GEMM + fused SpMM^T -> Gram) instead
                                        of the two-hop path route (csrc/paths.hip); same results up to fp32 reassociation */
#define LGNN_FLAG_FORCE_PATHS 8u     /* take the path route wherever the model's shape allows it, also on hub-heavy graphs where
                                        the library would choose the planes (its default weighs the batch's expected number of
                                        2-hop paths per node); the first eligible call counts the graph's 2-hop paths once and
                                        synchronises the stream that one time                                            */

typedef struct lgnn_ctx lgnn_ctx; /* opaque: graph + bound model + forward cache + workspace */

/* ---- library ------------------------------------------------------------------------ */
LGNN_API int lgnn_abi_version(void);
LGNN_API const char* lgnn_last_error(void);

/* ---- graph ingest: the integer path (bit exact) ---------------------------------------
 * Replaces gnn/utils.py:325-330 (edge_index -> dense adj, duplicates summed),
 * gnn/marglik_training.py:405 (clamp to 1), gnn/models/models.py:23 / :47 (self loops on /
 * off), gnn/models/base_gnn.py:68-72 (optional symmetrise) and gnn/models/utils.py:106-112 /
 * gnn/models/layers.py:18-24 (normalisation, done once here instead of on every forward).
 * edge_index: int64 [2,E] (row 0 = source/row, row 1 = target/col of the dense adj).
 * Synchronises `stream` once (the deduplicated nnz has to reach the host).                */
LGNN_API int lgnn_create(lgnn_ctx** out, int64_t num_nodes, const int64_t* edge_index, int64_t num_edges,
                int kind, int symmetric, void* stream);
LGNN_API void lgnn_destroy(lgnn_ctx* h);

/* nnz of the stored 0/1 adjacency (incl. self loops for GCN).  host value. */
LGNN_API int64_t lgnn_nnz(const lgnn_ctx* h);
LGNN_API int64_t lgnn_num_nodes(const lgnn_ctx* h);
/* Rows of the backward propagation matrix with more than 64 stored entries (hubs): the 256-wide fused kernel receives
 * them finished from a side kernel instead of gathering them in one wave.  -1 until the first KFAC call built the list. */
LGNN_API int64_t lgnn_num_long_rows(const lgnn_ctx* h);
/* 1 if the last lgnn_kfac_accumulate* call on this context took the two-hop path route (csrc/paths.hip), 0 if it built class
 * planes: which of the two implementations of curvlinops/kfac.py:653-661, 777-817 ran (measurement labels; host value). */
LGNN_API int lgnn_kfac_last_route(const lgnn_ctx* h);
/* 1 if the stored adjacency equals its transpose (then forward and backward share one CSR) */
LGNN_API int lgnn_is_symmetric(const lgnn_ctx* h);

/* Export the stored adjacency in the order `model.adj.nonzero()` yields (row-major):
 * rows,cols int64 [nnz].  (tests: bit-exactness against the reference.)                    */
LGNN_API int lgnn_export_adj(const lgnn_ctx* h, int64_t* rows, int64_t* cols, void* stream);
/* gnn/utils.py:333-336 adj_to_edge_index: diagonal dropped, row-major, int64 [2,E'].
 * Call with edge_index_out = NULL to get E' in *num_out (host); then again with a buffer. */
LGNN_API int lgnn_adj_to_edge_index(const lgnn_ctx* h, int64_t* edge_index_out, int64_t* num_out, void* stream);
/* Apply a structure-learning step to the stored 0/1 adjacency in place: `num_flips` pairs (rows[k], cols[k]) (device, int64)
 * and the state each shall have afterwards (device uint8: 1 = stored, 0 = absent).  What the reference does by writing its
 * dense `adj` parameter (`adj_optimizer.step()`, gnn/marglik_training.py:211-221) and re-thresholding it on the next forward
 * (`BinarizeSTE` + `fill_diagonal_(1)` + `normalize_adj`, gnn/models/models.py:103-116, gnn/models/utils.py:42-112): here the
 * caller (laplace_gnn_amd.models.STEGCN) thresholds its continuous values and hands over the entries that changed side.
 * Diagonal pairs are ignored (a GCN's self loops are overwritten ones, GraphSAGE's zeros); a pair listed twice or an id out of
 * range is an error.  No re-ingest: the flips are sorted and merged into both CSRs, degrees and the values of the propagation
 * matrices are recomputed, everything cached from the graph (forward pass, P X, long-row lists) is dropped; what depends on
 * the features only survives.  Synchronises the stream (the new entry count has to reach the host).                       */
LGNN_API int lgnn_update_adjacency(lgnn_ctx* h, const int64_t* rows, const int64_t* cols, const uint8_t* state,
                                   int64_t num_flips, void* stream);
/* Export the propagation matrix the convs multiply with (A_hat or A_bar) as COO in row-major
 * order: rows,cols int64 [nnz], vals fp32 [nnz].                                            */
LGNN_API int lgnn_export_propagation(const lgnn_ctx* h, int64_t* rows, int64_t* cols, float* vals, void* stream);

/* ---- model binding ----------------------------------------------------------------------
 * Replaces the module state gnn/models/base_gnn.py:62-76 + the parameter filter
 * laplace/curvature/curvature.py:74-79: L layers `convs.{l}.lin` with weight [dims[l+1], in_l]
 * and bias [dims[l+1]]; in_l = dims[l] (GCN) or 2*dims[l] (GraphSAGE, cat[x, mean_agg]).
 * X: [N, dims[0]].  Pointers are borrowed.  Binding (re)sizes the workspace and invalidates the
 * forward cache; call lgnn_invalidate() after changing weights in place.                     */
LGNN_API int lgnn_bind_model(lgnn_ctx* h, int num_layers, const int64_t* dims /* host [L+1] */,
                    const float* const* weights /* host array of L device ptrs */,
                    const float* const* biases /* host array of L device ptrs */, const float* X,
                    int activation, int likelihood);
/* Optional pieces of BaseGNN.forward (gnn/models/base_gnn.py:86-113 construction, :141-149 forward; shipped configs
 * gnn/configs/original/gcn_config.yaml:36-58 use norm: layer + res: True).  Call after lgnn_bind_model (which resets them):
 *   res_weights / res_biases : host arrays of L-1 device pointers, `res.{l}.weight` [dims[l+1], dims[l]] / `res.{l}.bias`
 *                              (NULL: res=False); x = res_l(x) + conv_l(adj, x) for every hidden layer;
 *   norm_kind                : LGNN_NORM_*; norm_weight / norm_bias [dims[l+1]] per hidden layer (`norms.{l}.weight|bias`),
 *                              norm_mean / norm_var = BatchNorm1d running statistics (eval mode; NULL otherwise).
 * The `res.{l}` Linears are Laplace parameters AFTER all `convs.*` (named_parameters order; laplace/curvature/
 * curvature.py:74-79 filters `norms.*` out): every per-parameter output grows accordingly -- A_out / B_out of the KFAC calls
 * take 2 L - 1 pointers (convs.0 .. convs.{L-1}, res.0 .. res.{L-2}; curvlinops/kfac.py:877-916 gives every nn.Linear its
 * block), diag / Jacobians / full GGN append the res parameters.  With extras bound the KFAC accumulate runs its unfused
 * route (GEMM, row-local norm backward, SpMM^T, Gram as separate kernels) and the diagonal GGN / Jacobians the generic plane
 * route; the adjacency gradient and the matrix-free GLM variance refuse such models.                                 */
LGNN_API int lgnn_bind_extras(lgnn_ctx* h, const float* const* res_weights, const float* const* res_biases, int norm_kind,
                     const float* const* norm_weight, const float* const* norm_bias, const float* const* norm_mean,
                     const float* const* norm_var, float norm_eps);
LGNN_API int lgnn_invalidate(lgnn_ctx* h);
/* bytes currently held by the context (graph + caches + workspace).  host value. */
LGNN_API int64_t lgnn_device_bytes(const lgnn_ctx* h);
/* cap for the backward workspace (chunks over classes are sized to fit); default 32 GiB */
LGNN_API int lgnn_set_workspace_limit(lgnn_ctx* h, int64_t bytes);

/* ---- forward: model(x_indices) -> [M, C]  (gnn/models/base_gnn.py:136-161, eval mode) ----- */
LGNN_API int lgnn_forward(lgnn_ctx* h, const int64_t* idx, int64_t M, float* out /* [M, C] */, void* stream);
/* all-node logits [N, C] (what forward() indexes into) */
LGNN_API int lgnn_forward_all(lgnn_ctx* h, float* out /* [N, C] */, void* stream);

/* ---- KFAC factors of one mini-batch --------------------------------------------------------
 * Replaces CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:77-108) =
 * KFACLinearOperator._compute_kfac (curvlinops/kfac.py:540-581, 607-661, 777-875) +
 * _rescale_kron_factors (curvlinops.py:46-53) + the loss (curvlinops.py:106).
 *   A_out[l]  [in_l, in_l]  += in_l^T in_l / n_train      (all N rows the Linear sees)
 *   B_out[l]  [out_l,out_l] += sum_c g_{l,c}^T g_{l,c}     (one backward per class column c)
 *   *loss_out               += factor * loss(model(idx), y)
 * idx int64 [M]; y int64 [M] (classification) -- regression passes fp32 targets [M, C] as y.
 * The batch is the unit: B has cross-sample terms inside a batch, so callers
 * must keep the reference loader's batch boundaries.                                          */
LGNN_API int lgnn_kfac_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                         uint32_t flags, float* const* A_out /* host array of L device ptrs */,
                         float* const* B_out /* host array of L device ptrs */, float* loss_out,
                         void* stream);

/* The same for the class columns [class_begin, class_end) only.  B_l is a sum over the C class columns of
 * the Hessian square root (one reference backward pass each, curvlinops/kfac.py:653-661), so class ranges
 * are exact additive shares of a batch: a data-parallel caller can balance (batch, class-range) units over
 * ranks without ever splitting a batch's samples.  The share that contains class 0 also adds the loss and
 * the A increment (once per batch).                                                                  */
LGNN_API int lgnn_kfac_accumulate_classes(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                                 uint32_t flags, int64_t class_begin, int64_t class_end,
                                 float* const* A_out, float* const* B_out, float* loss_out, void* stream);

/* Parts [part_begin, part_end) of `part_count` equal parts of the batch's work -- the unit a data-parallel caller deals
 * to its ranks (laplace_gnn_amd.KronLaplace: part_count = C, contiguous balanced runs of (batch, part) units).  HOW a
 * batch is cut is the library's choice, the same on every rank for the same (graph, model, batch size, flags):
 *   - path routes (2-layer GCN / GraphSAGE, lgnn_kfac_last_route == 1): B_0 = sum over destination nodes n of Y_n^T Y_n,
 *     so a part is the node range [N part_begin / part_count, N part_end / part_count) with ALL classes -- nothing is
 *     computed twice (a class range would repeat the path products, which do not depend on the class count);
 *     the small top-layer factor goes with part 0;
 *   - everywhere else: the class range [C part_begin / part_count, C part_end / part_count) of
 *     lgnn_kfac_accumulate_classes (possibly empty).
 * The parts of a batch add up to lgnn_kfac_accumulate exactly (sums of disjoint terms); part 0 also adds the loss and
 * the A increment.  Replaces the same reference loop as lgnn_kfac_accumulate (curvlinops/kfac.py:653-661, 777-817).      */
LGNN_API int lgnn_kfac_accumulate_share(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                               uint32_t flags, int64_t part_begin, int64_t part_end, int64_t part_count,
                               float* const* A_out, float* const* B_out, float* loss_out, void* stream);

/* ---- empirical / Monte-Carlo Fisher -----------------------------------------------------------------------------
 * KFAC with FisherType.EMPIRICAL / FisherType.MC (curvlinops/kfac.py:663-674; reached through CurvlinopsEF and
 * CurvlinopsGGN(stochastic=True), laplace/curvature/curvlinops.py:143-179): ONE backward pass per call, seeded with
 * resid_scale * d loss(f_n, y_seed_n) / d f_n  (softmax - onehot, resp. f - y_seed for the regression likelihood);
 * B_out[l] += b_scale * g^T g  (b_scale = 1 / mc_samples).  y_loss != NULL: the call also adds factor-free
 * loss(model(idx), y_loss) and the A increment (once per batch); y_loss == NULL: further MC samples of a batch.
 * MC labels are drawn by the caller (curvlinops/kfac.py:698-745 draw_label) and passed as y_seed.             */
LGNN_API int lgnn_kfac_accumulate_fisher(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss, int64_t M,
                                int64_t n_train, uint32_t flags, float resid_scale, float b_scale, float* const* A_out,
                                float* const* B_out, float* loss_out, void* stream);
/* Per-sample loss gradients G[m, :] = J_m^T r_m (CurvatureInterface.gradients, laplace/curvature/curvature.py:169-210)
 * and what EFInterface (:435-504) and GGNInterface with stochastic=True (:343-364, 401-432) build from them:
 *   grads_out [M, P]  = G (may be NULL)     diag_out [P] += scale * sum_m G[m, p]^2 (may be NULL)
 *   full_out [P, P]  += scale * G^T G (may be NULL)         *loss_out += loss(model(idx), y_loss) when y_loss != NULL
 * r_m = resid_scale * (softmax(f_m) - onehot(y_seed_m)), regression: resid_scale * (f_m - y_seed_m).          */
LGNN_API int lgnn_ef_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss, int64_t M,
                       float resid_scale, float scale, float* diag_out, float* full_out, float* grads_out,
                       float* loss_out, void* stream);

/* Host-only query (no context, no device work): which kernels lgnn_kfac_accumulate would run for a model of this
 * shape -- the per-layer choice between the fused SpMM^T -> Gram kernel and the SpMM + Gram pair through HBM, the
 * compacted backward GEMM, and the workspace layout (curvlinops/kfac.py:653-661 has one autograd backward per class
 * instead; nothing to choose there).  out: int64 [4 + num_layers] = {seeds_on_the_fly, sage_compact, need_pong,
 * classes_per_chunk, step_0 .. step_{L-1}} with step_l = 1 (fused) | 2 (compacted backward GEMM) | 4 (step L-1 only: the
 * first layer's B comes from the batch's two-hop paths, no class planes); step_0 is unused. */
LGNN_API int lgnn_kfac_plan(int kind, int num_layers, const int64_t* dims /* host [L+1] */, int64_t num_nodes, int64_t nnz,
                   int activation, uint32_t flags, int64_t workspace_limit, int64_t* out /* host */);

/* ---- diagonal GGN of one mini-batch ----------------------------------------------------------
 * Replaces GGNInterface.diag (laplace/curvature/curvature.py:412-432 with jacobians :89-130 and
 * _get_functional_hessian :365-372):  diag_out[P] += einsum('bcp,bck,bkp->p', J, Lambda, J),
 * parameter order = named_parameters() (W0 row-major, b0, W1, b1, ...).  1- and 2-layer models: closed form, no
 * Jacobians; deeper models and the regression likelihood (y = fp32 targets [M, C], H_lik = None: sum J^T J): chunks
 * of per-sample Jacobians contracted on the device.                                              */
LGNN_API int lgnn_diag_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags,
                         float* diag_out /* [P] */, float* loss_out, void* stream);

/* ---- full GGN over all weights of one mini-batch ---------------------------------------------------------------
 * Replaces CurvlinopsInterface.full (laplace/curvature/curvlinops.py:110-140: GGNLinearOperator @ I, one GGN-vector
 * product per column, curvlinops/ggn.py:44-75) = GGNInterface.full (laplace/curvature/curvature.py:374-410):
 * H_out[P, P] += sum_n J_n^T Lambda_n J_n (regression: sum_n J_n^T J_n, no factor), parameters in named_parameters
 * order; *loss_out += loss(model(idx), y) without the interface factor.  Chunks of per-sample Jacobians are mixed in
 * place with the Hessian square root and contracted by the fp32 MFMA Gram kernel; for models whose P x P fits.     */
LGNN_API int lgnn_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out, float* loss_out,
                         void* stream);

/* ---- last-layer full GGN of one mini-batch ------------------------------------------------------
 * Replaces GGNInterface.full with last_layer_jacobians (laplace/curvature/curvature.py:132-167,
 * 374-410): J_n = [I_C (x) phi_n^T | s_n I_C], H_out[P_ll, P_ll] += sum_n J_n^T Lambda_n J_n,
 * P_ll = C*(D+1), weight index c*D+d then C bias entries.                                         */
LGNN_API int lgnn_lastlayer_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M,
                                   float* H_out, float* loss_out, void* stream);

/* What CurvatureInterface.last_layer_jacobians (laplace/curvature/curvature.py:132-167) builds its Jacobians from:
 * phi_out [M, D + 1] = [input row of the final nn.Linear seen from the batch node | bias scale s_n] (GraphSAGE:
 * [h_n | (A_bar h)_n | 1]; GCN: [(A_hat h)_n | rowsum(A_hat)_n]), f_out [M, C] = logits (or NULL);
 * J_n = [I_C (x) phi_n^T | s_n I_C].                                                                              */
LGNN_API int lgnn_lastlayer_features(lgnn_ctx* h, const int64_t* idx, int64_t M, float* phi_out, float* f_out, void* stream);

/* The same matrix accumulated in its natural layout: H = sum_n Lambda_n (x) phi~_n phi~_n^T, so block (c, c') of H is the
 * weighted Gram Phi~^T diag(Lambda[:, c, c']) Phi~ -- symmetric in (c, c') and inside the block.  S_pairs
 * [C (C + 1) / 2][D][D] (pairs c <= c' row major; the upper 32 x 32 sub-tiles of each block are valid) and Sb_pairs
 * [C (C + 1) / 2][D + 1] (bias column of each block) are caller-owned and ADDED to: a fit accumulates every batch
 * there (a data-parallel caller all-reduces them: half the bytes of H) and calls lgnn_lastlayer_pairs_place ONCE,
 * which adds the blocks into H_out [P_ll, P_ll] and mirrors the upper triangle.                                  */
LGNN_API int lgnn_lastlayer_pairs_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* S_pairs,
                                    float* Sb_pairs, float* loss_out, void* stream);
LGNN_API int lgnn_lastlayer_pairs_place(lgnn_ctx* h, const float* S_pairs, const float* Sb_pairs, float* H_out, void* stream);

/* Invalid batch contents (node ids outside [0, N), labels outside [0, C)) are detected on the device: such
 * samples contribute nothing and a sticky flag is raised.  This call synchronises `stream`, reports the flag
 * (non-zero return + message) and clears it; call it once per fit, not per batch.                          */
LGNN_API int lgnn_check_async_errors(lgnn_ctx* h, void* stream);
/* The same report WITHOUT synchronising: the flags are sticky words in pinned host memory, so this sees the errors of every kernel
 * that has finished; errors of work still in flight are reported by the next peek or check.  For callers that queue fit after fit
 * (a replayed hipGraph of a whole fit, DiagLaplace.fit_graph) and must not stall the stream once per fit; the reference's own
 * index errors on a CUDA / HIP device are asynchronous device-side asserts as well.                                          */
LGNN_API int lgnn_peek_async_errors(lgnn_ctx* h);

/* ---- timing hook (bench.py roofline) -----------------------------------------------------------
 * While enabled, every launch of the dominant kernel of the path in use -- KFAC: the fused SpMM^T -> Gram kernel of
 * the lowest layer; diagonal GGN: the first-layer kernel; last-layer full GGN: the batched weighted Gram -- is
 * bracketed by HIP events recorded on `stream` itself (torch.cuda.Event only sees torch's current stream).
 * lgnn_kernel_timing_read synchronises on the recorded events and returns the number of launches, their summed
 * duration in ms and the summed number of units they processed (KFAC: class planes; diagonal: samples; last layer:
 * class pairs); enabling resets the counters.                                                     */
LGNN_API int lgnn_enable_kernel_timing(lgnn_ctx* h, int enable);
LGNN_API int lgnn_kernel_timing_read(lgnn_ctx* h, int64_t* launches, double* total_ms, int64_t* planes);
/* The same events, launch by launch (host array of `capacity` doubles, milliseconds, in launch order; *launches = how many
 * were recorded): bench.py separates the full batches' launches from the short last batch's.  Synchronises on the events. */
LGNN_API int lgnn_kernel_timing_launches(lgnn_ctx* h, double* ms_out, int64_t capacity, int64_t* launches);

/* ---- per-sample Jacobians for the GLM predictive ("next" row) ------------------------------------
 * Replaces CurvatureInterface.jacobians (laplace/curvature/curvature.py:89-130, torch.func.jacrev of the dense
 * model): J [M, C, P] fp32 with J[m][c][:] = d f[idx[m], c] / d theta, parameters in named_parameters order
 * (per layer: weight [out, in] row major, then bias); f_out [M, C] = logits[idx] or NULL.  M*C*P floats, as the
 * reference; the samples are processed in chunks that fit the workspace cap.                              */
LGNN_API int lgnn_jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, void* stream);

/* ---- gradient of the negative log marginal likelihood w.r.t. the adjacency ("next" row 8(f)-4) --------------
 * Replaces what autograd does for ``neg_marglik.backward()`` into ``model.adj`` (gnn/marglik_training.py:197-216):
 * back through KronDecomposed.logdet / log_marginal_likelihood (laplace/utils/matrix.py:371-394,
 * laplace/baselaplace.py:938-973), the KFAC accumulation the fork keeps attached to the graph (curvlinops/kfac.py:
 * 637-661 non-detached square root + create_graph, :789-790, :836-837), the forward passes, normalize_adj
 * (gnn/models/utils.py:106-112) and the straight-through binarisation (gnn/models/utils.py:42-86).  2-layer GCN,
 * ReLU, classification.  The caller supplies gamma_B[l] = d(neg marglik)/dB_l, gamma_A[l] = d/dA_l (from the
 * eigenpairs of the fitted factors; fp32 [out_l, out_l] / [in_l, in_l], symmetric).
 *   per batch :  grad_P [nnz] += the terms of this batch on the stored entries of the propagation matrix (order of
 *                lgnn_export_propagation);  out_bar [N, C] += d/d(all-node logits) of loss_scale * CE and of the seeds
 *   once      :  adds the forward-pass terms (a_scale = batches / N_train: A_1 = a_scale * H1^T H1) to grad_P, then
 *                grad_adj [nnz] = gradient w.r.t. the stored entries of the 0/1 adjacency, order of lgnn_export_adj
 *                (diagonal 0: fill_diagonal_(1) overwrites it; symmetric models: average of (i,j) and (j,i)).
 * Candidate edges (num_cand may be 0): the reference's dense adj.grad also has an entry for every NON-edge -- that is how
 * its structure learning proposes edges.  cand_a / cand_b int32 [num_cand] list pairs in the propagation matrix's
 * coordinates (entry (i, j) of the adjacency <-> a = j, b = i); grad_cand [num_cand] accumulates like grad_P and
 * lgnn_adjgrad_finish writes grad_cand_adj [num_cand] = d / d adj[i, j] (no symmetrisation: a symmetric model's caller
 * lists both orientations and averages).
 * Models bound with res / norm (lgnn_bind_extras; gnn/models/base_gnn.py:141-149 -- the STE-GCN configurations of Cornell /
 * Texas / Wisconsin / Circle, gnn/configs/original/stegcn_config.yaml:54-105, 129-145): GCN only.  gamma_B then has one more
 * entry per hidden layer after the conv entries (the res.{l} blocks, named_parameters order; res=True only); the chain gains
 * the norm's row-local backward, the res block's B and the adjoint of the pre-norm rows through the LayerNorm statistics.
 * All N rows, unfused kernels: these configurations are graphs of a few hundred to a few thousand nodes.               */
LGNN_API int lgnn_kfac_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags,
                            const float* const* gamma_B /* host array of L device ptrs */, float loss_scale,
                            float* grad_P, float* out_bar, const int32_t* cand_a, const int32_t* cand_b, int64_t num_cand,
                            float* grad_cand, void* stream);
LGNN_API int lgnn_adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* const* gamma_A /* host array of L device ptrs */,
                        float a_scale, float* grad_P, float* grad_adj, const int32_t* cand_a, const int32_t* cand_b,
                        int64_t num_cand, float* grad_cand, float* grad_cand_adj, void* stream);

/* The same gradient under a DIAGONAL posterior -- what the shipped STE-GCN configuration differentiates
 * (gnn/configs/original/stegcn_config.yaml:7 hessian_structure: diag; gnn/marglik_training.py:197-216 with DiagLaplace,
 * laplace/baselaplace.py:1848-1857; the fork's GGNInterface.jacobians keeps the graph, laplace/curvature/curvature.py:89-130,
 * so diag() (:412-432) is differentiable in the adjacency).  2-layer GCN, ReLU, classification.
 *   gamma [n_params] (parameter order W_0, b_0, W_1, b_1) = d(-marglik)/dH_p = f / (2 (f H_p + delta_p)), f = H_factor;
 *   per batch :  grad_P [nnz] += the terms on the stored entries of the batch nodes' rows;  out_bar [N, C] += d/d(logits) of
 *                loss_scale * CE and of the softmax inside the GGN;  h1_bar [N, H] += d/dH_1 and e_bar [N, F + 1] +=
 *                d/d[P X | rowsum(P)] of the closed-form diagonal;  candidates as above;
 *   once      :  lgnn_diag_adjgrad_finish propagates out_bar, h1_bar and e_bar through the forward pass into grad_P and writes
 *                grad_adj [nnz] / grad_cand_adj exactly like lgnn_adjgrad_finish.
 * Workspace: [chunk][H][F + 1 rounded up to 4] floats for the first-layer tiles of a chunk of samples (0.5 GB for a Cora-shaped
 * batch), under the workspace limit (lgnn_set_workspace_limit); no synchronisation, nothing allocated after the first call.
 * Models bound with res / norm (GCN; the shipped WebKB / Circle configurations are exactly diag + res + LayerNorm): no closed
 * form of the diagonal GGN exists; gamma covers the res.{l} parameters too (after W_1, b_1), e_bar is left untouched, and per
 * (sample, class) the kernel chain runs one tangent forward pass along R = 2 Lambda J diag(gamma) and its reverse pass
 * (d sum_p gamma_p H_p = <K, d Lambda> + <R, d J>, K = J diag(gamma) J^T).  Workspace per sample of a chunk: 2 C P floats
 * (Jacobian rows and directions) + C N (3 H + C) floats of planes -- sized for graphs of a few hundred to a few thousand nodes.
 * Plain 2-layer GraphSAGE models (STEGraphSAGE + DiagLaplace; the driver offers the pair, gnn/utils.py:55-59, 81): the same identity,
 * evaluated locally -- everything of a (sample, class) pair lives on the rows {n} + N(n), one workgroup per pair, no planes; e_bar
 * [N, F + 1] carries the adjoint of P X in its first F columns, h1_bar the direct adjoint of H_1; workspace 2 C P floats per sample. */
LGNN_API int lgnn_diag_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, const float* gamma,
                            float loss_scale, float* grad_P, float* out_bar, float* h1_bar, float* e_bar,
                            const int32_t* cand_a, const int32_t* cand_b, int64_t num_cand, float* grad_cand, void* stream);
LGNN_API int lgnn_diag_adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* h1_bar, const float* e_bar, float* grad_P,
                             float* grad_adj, const int32_t* cand_a, const int32_t* cand_b, int64_t num_cand,
                             float* grad_cand, float* grad_cand_adj, void* stream);

/* ---- matrix-free GLM predictive ("next" row 8(f)-3 at scale) -------------------------------------------------------
 * Replaces the Jacobian route of the default la(x) (laplace/baselaplace.py:1123-1158 + laplace/utils/matrix.py:396-451 resp.
 * baselaplace.py:1901-1903) for 2-layer GCN and GraphSAGE models: f_mu [M, C] = logits and f_var_diag [M, C] = diag(J P^-1 J^T) per
 * evaluation node from the closed-form Jacobian, nothing of size M * C * P is formed.  The probit link (the default,
 * baselaplace.py:610-616) needs exactly this diagonal.
 *   Kronecker posterior: QA0 [F, F], QB0 [H, H], QA1 [H, H] = eigenvectors (columns) of A_0, B_0, A_1;
 *     S0 [H, F + 1]: 1 / (f lB0_i lA0_j + delta_W0), column F: 1 / (f lB0_i + delta_b0);  S1 [C, H]: 1 / (f lB1_i lA1_j + delta_W1);
 *     QB1sq [C, C] = Q_B1[c, i]^2;  kappa [C] = sum_i Q_B1[c, i]^2 / (f lB1_i + delta_b1)      (bias blocks share Q_B)
 *   diagonal posterior: QA0 = QB0 = QA1 = QB1sq = NULL; S0 [H, F + 1] = 1 / precision of (W_0 | b_0), S1 [C, H] of W_1,
 *     kappa [C] of b_1.
 *   GraphSAGE: F and the H of QA1 / S1's second dimension read as the widths of what the two Linear layers multiply
 *     (2 F and 2 H: cat = [h | P h], gnn/models/layers.py:26-29).                                                */
LGNN_API int lgnn_glm_variance(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* QA0, const float* QB0, const float* S0,
                      const float* QA1, const float* S1, const float* QB1sq, const float* kappa, float* f_mu,
                      float* f_var_diag, void* stream);
/* The same for a linear map of the logits: var_mapped [M, Cm] = diag(E J P^-1 J^T E^T) for E [Cm, C], handed over as the
 * mapped head W1m = E W_1 [Cm, in_dim_1] (row major; GraphSAGE: both halves).  With E = [I ; 1^T / C ; I + 1 1^T / C]
 * (Cm = 2 C + 1) polarisation gives the row sums and the total of the C x C predictive covariance next to its diagonal --
 * all that the Laplace bridge with its zero-mean correction reads (link_approx "bridge" / "bridge_norm",
 * laplace/baselaplace.py:637-661: f_var.sum(-1), f_var.sum((1, 2)), diagonal) -- so those links no longer need Jacobians.
 * Operands as above except: Kronecker posterior S1 [C, .] as above, QB1sq [Cm, C] = (E Q_B1)[r, i]^2,
 * kappa [Cm] = sum_i (E Q_B1)[r, i]^2 / (f lB1_i + delta_b1);  diagonal posterior S1 [Cm, .] = (E * E) S1,
 * kappa [Cm] = (E * E) kappa (elementwise squares of E: the posterior is independent across parameters).
 * f_mu [M, C] stays the model's logits (may be NULL).  Cm <= 4096.                                                  */
LGNN_API int lgnn_glm_variance_mapped(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* W1m, int64_t Cm,
                             const float* QA0, const float* QB0, const float* S0, const float* QA1, const float* S1,
                             const float* QB1sq, const float* kappa, float* f_mu, float* var_mapped, void* stream);

/* ---- decomposition of the fitted factors ("next" row: KronLaplace.fit -> Kron.decompose) ----------
 * Replaces the per-factor torch.linalg.eigh calls of laplace/utils/matrix.py:118-145 (symeig,
 * laplace/utils/utils.py:193-226) by one call for all factors: n <= 256 -- a one-workgroup Householder tridiagonalisation
 * per factor (all factors side by side), rocSOLVER's divide and conquer on the tridiagonal matrices (one chain per factor on
 * side streams joined back into `stream`), a back-transformation kernel; larger n -- ONE strided-batched rocSOLVER syevd.
 * A: device fp32 [batch][n][n], symmetric, overwritten with the eigenvectors: ROW j of matrix b is the unit eigenvector of
 * eigenvalue W[b][j]; W: device fp32 [batch][n], ascending; info: device int32 [batch] (0 = converged).  Asynchronous on
 * `stream`; no graph handle involved.    */
LGNN_API int lgnn_symeig_batched(float* A, int64_t n, int64_t batch, float* W, int32_t* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAPLACE_GNN_HIP_H */
