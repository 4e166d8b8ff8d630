// Internal declarations shared by the HIP translation units of liblaplace_gnn_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <cstring>  // rocprim's texture iterator needs host memset declared first
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/laplace_gnn_hip.h"

namespace lgnn {

void set_error(const std::string& msg);

#define LGNN_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      lgnn::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

#define LGNN_REQUIRE(cond, msg)                                                           \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      lgnn::set_error(std::string(msg) + " (" #cond ")");                                 \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)

#define LGNN_CALL(expr)                                                                   \
  do {                                                                                    \
    int _rc = (expr);                                                                     \
    if (_rc != 0) return _rc;                                                             \
  } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grow-only device buffer (no allocation in steady state)
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool host = false;  // reserve_host: pinned host memory the device reads and writes in place (same address on both sides)
  int reserve(size_t want);
  int reserve_host(size_t want);
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// CSR with int32 indices (N, nnz < 2^31), fp32 values aligned with `col`
struct Csr {
  int32_t* rowptr = nullptr;  // [n+1]
  int32_t* col = nullptr;     // [nnz]
  float* val = nullptr;       // [nnz]
};

constexpr int kMaxLayers = 8;

struct ForwardCache {
  bool valid = false;
  bool aux_valid = false;            // rowsum / propagated inputs for diag + last layer
  bool wt_valid = false;             // lgnn_ctx::Wt (transposed weights) match the bound weights' current version
  // What depends on the graph and X only -- not on the weights -- survives lgnn_invalidate and is rebuilt at bind time:
  bool x_valid = false;              // Xpad, gram_raw[0] (GCN: X^T X)
  bool px_valid = false;             // rowsum, prop_in[0] (GCN: P X)
  DevBuf Xpad;                       // GCN, F % 4 != 0: X copied to rows of F rounded up to 4 floats (zero padded) so that
                                     // the first GEMM, SpMM and Gram run their 16-byte paths (Cora: F = 1 433)
  int64_t prop_ld[kMaxLayers] = {};  // row stride of prop_in[l]
  bool gram_valid[kMaxLayers] = {};  // raw input Grams (upper triangle) per layer
  DevBuf lin_in[kMaxLayers];         // GraphSAGE: cat_l = [h_l | P h_l]  [N, 2 d_l]
  DevBuf act_out[kMaxLayers];        // GCN: h_{l+1} = act(P Z_l), l < L-1 : [N, out_l]
  const float* lin_in_p[kMaxLayers] = {};  // what nn.Linear l sees: [N, in_l] with row stride lin_in_ld
  int64_t lin_in_ld[kMaxLayers] = {};
  const float* hact_p[kMaxLayers] = {};    // activation output h_{l+1} (input of layer l+1), row stride hact_ld
  int64_t hact_ld[kMaxLayers] = {};
  DevBuf out;                        // logits for all nodes [N, C]
  DevBuf tmp;                        // [N, max width] scratch of the forward
  DevBuf gram_raw[kMaxLayers];       // [in_l, in_l] raw in^T in (upper sub-tiles valid)
  DevBuf prop_in[kMaxLayers];        // P @ lin_in[l]  [N, in_l]   (diag / last layer; GCN)
  DevBuf rowsum;                     // rowsum(P) [N]
  DevBuf dact0;                      // act'(h_1) [N, dims[1]] (closed-form diagonal GGN of 2-layer models)
  DevBuf mask_bits[kMaxLayers];      // ReLU: bit j of word w of node n = (h_{l+1}[n][32w+j] > 0)
  // models with res / norm (gnn/models/base_gnn.py:141-149): the norm's backward needs the normalised rows and 1/sigma
  DevBuf xhat[kMaxLayers];           // [N, dims[l+1]] normalised pre-activation of hidden layer l
  DevBuf rstd[kMaxLayers];           // LayerNorm: [N] 1/sqrt(var + eps) per row; BatchNorm (eval): [dims[l+1]] per channel
  DevBuf res_out;                    // [N, max width] res_l(h_l) of the layer being computed (GCN)
  DevBuf pre_norm;                   // [N, max width] s_l = res_l(h_l) + conv_l(h_l) before the norm
};

struct Workspace {
  DevBuf pos;       // int32 [N]   batch position of a node, INT_MAX if not in batch
  DevBuf seeds;     // fp32 [M, C, C] (c-major rows inside a sample: [m][c][k])
  DevBuf probs;     // fp32 [M, C] softmax
  DevBuf mult;      // int32 [M]: occurrences of the node whose first batch position this is (0 elsewhere)
  DevBuf planes_a;  // backward planes, ping
  // GraphSAGE compact top level: planes_a is kept all zero outside the rows of the current batch (which are cleared
  // again after use); these remember for which buffer / extent that holds
  const void* planes_a_zero_ptr = nullptr;
  size_t planes_a_zero_bytes = 0;
  DevBuf planes_b;  // backward planes, pong
  DevBuf gram_scratch[kMaxLayers];  // [out_l, out_l] per-call partial B (upper sub-tiles)
  DevBuf misc;
  DevBuf jac;      // fp32 [chunk, C, P]: Jacobians of a chunk of samples (generic-depth diagonal GGN)
  DevBuf top;    // top-layer gradient planes [C][N][C]
  DevBuf active;   // uint8 [N]: node has a non-zero top-layer gradient row
  DevBuf val_act;  // fp32 [nnz]: P^T values with inactive source columns zeroed
  DevBuf act_list; // int32 [N]: sorted ids of the active nodes; act_count: their number (device int32)
  DevBuf act_count;
  DevBuf select_tmp;
  DevBuf flags;  // 64 B of asynchronous error flags
  DevBuf out_flags, out_list, out_count;  // GraphSAGE KFAC: rows of the first backward plane set that can be non-zero
  DevBuf val_act2;  // P^T's values with the columns outside that row set zeroed (second backward level of deeper models)
  // two-hop path route of the first layer's B (paths.hip): per-sample coefficient rows [M][3][64], the rows (u | p) [2 M][C]
  // and their products with W_1 [2 M][H]; R = P^T[:, batch] as CSR over v (counts / cursors, row pointers, sample, weight)
  DevBuf path_coef, path_up, path_bg, path_cnt, path_rptr, path_rm, path_rw, path_zeros;
  DevBuf path_alpha;  // GraphSAGE: the samples' alpha coefficients, class major [M][64] (weights of the one-hot paths)
  DevBuf path_pcnt, path_pptr, path_pm, path_pv, path_pw;  // the batch's paths per destination node (CSR over n)
  DevBuf path_flags, path_nodes, path_nnodes;              // nodes of the launch's range that have a path: flags, list, count
  // (the list's capacity is max(4 nnz, 4 M entries); a batch whose list is longer takes the enumerating route on the device)
  bool path_zeros_set = false;
  DevBuf gram_scratch_res[kMaxLayers];  // res / norm models: per-call partial B of res.{l} (GCN; GraphSAGE shares the conv's)
  DevBuf planes_c;  // GCN with res, >= 3 layers: u_l Wr_l of the level being computed; adjacency gradient (diag, res / norm): tangent planes
  DevBuf adj_z0;    // adjacency gradient of res / norm models: Z0 = X W0^T + b0 [N, H]
  DevBuf adj_dir;   // the same, diagonal posterior: parameter directions R [chunk, C, P]
};

}  // namespace lgnn

struct lgnn_ctx {
  int64_t N = 0;
  int64_t nnz = 0;
  int kind = 0;
  bool sym = false;
  // stored 0/1 adjacency A (rows) and its transpose; AT aliases A when symmetric
  lgnn::Csr A, AT;
  lgnn::DevBuf A_rowptr, A_col, AT_rowptr, AT_col, val_fwd, val_bwd, deg_scale;
  // propagation matrix P (forward) and P^T (backward) views into the buffers above
  lgnn::Csr P, PT;
  // model
  int L = 0;
  int64_t dims[lgnn::kMaxLayers + 1] = {};
  int64_t in_dim[lgnn::kMaxLayers] = {};  // columns of weight l (2*dims[l] for GraphSAGE)
  const float* W[lgnn::kMaxLayers] = {};
  const float* b[lgnn::kMaxLayers] = {};
  const float* X = nullptr;
  int act = 0, lik = 0;
  int64_t n_params = 0;
  lgnn::DevBuf Wt[lgnn::kMaxLayers];  // W_l^T [in_l, out_l] (forward GEMM operand)
  // optional pieces of BaseGNN.forward (gnn/models/base_gnn.py:86-113, 141-149), bound by lgnn_bind_extras:
  // x = res_l(x) + conv_l(adj, x); x = norms[l](x); x = act(x) for every hidden layer l < L-1
  bool has_res = false;
  int norm = 0;            // LGNN_NORM_*
  float norm_eps = 1e-5f;
  const float* Wr[lgnn::kMaxLayers] = {};  // res.{l}.weight [dims[l+1], dims[l]]
  const float* br[lgnn::kMaxLayers] = {};
  const float* norm_w[lgnn::kMaxLayers] = {};
  const float* norm_b[lgnn::kMaxLayers] = {};
  const float* norm_mean[lgnn::kMaxLayers] = {};  // BatchNorm1d running statistics (eval mode)
  const float* norm_var[lgnn::kMaxLayers] = {};
  lgnn::DevBuf Wrt[lgnn::kMaxLayers];    // GCN: Wr_l^T [dims[l], dims[l+1]] (forward GEMM operand)
  // GraphSAGE with res: convs.{l}.lin and res.{l} read the same rows, so the forward and the backward use ONE combined
  // weight  Wcomb_l = W_l + [Wr_l | 0]  [dims[l+1], 2 dims[l]] (row major; its transpose goes to Wt[l]) and bias b_l + br_l
  lgnn::DevBuf Wcomb[lgnn::kMaxLayers], bcomb[lgnn::kMaxLayers];
  bool extras() const { return has_res || norm != 0; }
  int n_blocks() const { return has_res ? 2 * L - 1 : L; }  // KFAC blocks: convs.{0..L-1}.lin, then res.{0..L-2}
  const float* Wback(int l) const {  // what the backward through layer l multiplies with
    return (has_res && kind == LGNN_KIND_SAGE && l < L - 1) ? Wcomb[l].as<float>() : W[l];
  }
  lgnn::ForwardCache fc;
  lgnn::Workspace ws;
  int64_t ws_limit = int64_t(32) << 30;  // backward planes (ping + pong) per class chunk: 288 GB of HBM, keep chunks large
  // rows of P^T with more than kLongRow stored entries (hubs), built once on first use (longrows.hip)
  bool last_route_paths = false;   // the last KFAC accumulate took the two-hop path route
  double two_hop_max = -1.0;       // largest number of 2-hop paths starting at one node (same count pass)
  double two_hop = -1.0;           // number of 2-hop paths n <- v <- m of the graph (-1: not counted yet; paths.hip)
  int64_t n_long = -1;             // -1: not looked at yet
  int64_t n_long_tasks = 0;
  lgnn::DevBuf long_rows, long_slot, long_tasks, hub;
  // top-layer kernel (kfac.hip): hubs with more than kTopSlice entries are cut into slices; their node ids, count, and the
  // number of slices over all of them (host copies); per-batch task list and partial tiles
  lgnn::DevBuf top_multi, top_tasks, top_task_count, top_cnt, top_offs, top_hub_tiles;
  int64_t n_top_multi = 0, n_top_slices = 0;
  // the same list for the forward matrix P (the forward SpMMs)
  int64_t n_long_fwd = -1;
  lgnn::DevBuf long_rows_fwd;
  // timing of the dominant kernel
  bool timing = false;
  std::vector<hipEvent_t> ev;   // pairs (start, stop), grown on demand
  size_t ev_used = 0;           // events recorded since the last reset
  int64_t ev_planes = 0;
};

namespace lgnn {

struct SpmmArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  int64_t nrows;
  const float* in;
  int64_t in_ld;
  int64_t in_plane_stride;
  float* out;
  int64_t out_ld;
  int64_t out_plane_stride;
  int64_t width;
  const float* self;  // optional [plane][r][self_ld]
  int64_t self_ld;
  int64_t self_plane_stride;
  const float* hact;  // optional: multiply by act'(hact[r]) (shared by all planes)
  int64_t hact_ld;
  int act;
  int out_act;  // -1 none, else apply activation to the result
  // rows with more than kLongRow stored entries (hubs), optional: the row kernel skips them and a side kernel with a
  // whole workgroup per row writes them (vector path only)
  const int32_t* long_rows; int64_t n_long;
  // entries whose value is exactly zero are not gathered (per-batch copies of P^T's values with the columns of all-zero
  // source rows zeroed: at the products shape 99.6 % of the first backward SpMM's entries)
  bool skip_zero;
};

struct FusedArgs {
  const int32_t* rowptr; const int32_t* col; const float* val;
  int64_t nrows;     // rows per plane (N)
  int64_t nplanes;
  const float* in; int64_t in_ld; int64_t in_plane_stride;
  float* store; int64_t store_ld; int64_t store_plane_stride;  // optional
  const float* self; int64_t self_ld; int64_t self_plane_stride;  // optional
  const float* hact; int64_t hact_ld; int act;                    // optional
  // GraphSAGE fast path of the 256-wide kernel (both set): rows whose flag is 0 have an all-zero self row (not read);
  // ReLU derivative from bit masks [nrows][mask_words] instead of the float activations
  const uint8_t* self_rows; const uint32_t* mask_bits; int mask_words;
  int64_t width;     // D <= DT
  float* scratch;    // [D, D]
  int debug;         // dev experiments: 1 = skip MFMAs, 2 = skip gathers
  // 256-wide kernel: rows with more than 64 stored entries arrive finished from long_rows_spmm (longrows.hip):
  // long_slot[row] = slot in hub (-1: ordinary row), hub [plane][n_long][width]
  const int32_t* long_slot; const float* hub; int64_t hub_plane_stride; int64_t n_long;
  // 256-wide kernel, GraphSAGE fast path: only the rows listed here can be non-zero (the batch nodes and their neighbours:
  // 58 % of the rows at the arxiv shape) -- a block is 32 LISTED rows, the others are never visited.  count on the device.
  const int32_t* row_list; const int32_t* row_count;
};

struct BackGemmArgs {
  const float* G;   // planes [P][N][K]
  const float* W;   // [K][ldw] row major, Nout columns used
  int64_t ldw;
  float* U;         // planes [P][N + 1][Nout]: row N of a plane is a spare row (takes the stores of padding rows)
  int64_t u_plane_stride;  // floats between planes of U, >= (N + 1) * Nout
  int64_t N, K, Nout;
  int64_t planes;
  const int32_t* rows;      // optional sorted list of active nodes (null: all N)
  const int32_t* na_dev;    // device count of `rows`
  const uint32_t* mask_bits; int64_t mask_words;  // ReLU mask bits of h (or null)
  const float* hact; int64_t hact_ld; int act;    // float activations otherwise
};
bool backgemm_supported(int64_t K, int64_t Nout, bool with_mask);
int launch_backgemm(const BackGemmArgs& g, hipStream_t s);
int launch_relu_mask_bits(const float* h, int64_t ld, int64_t N, int64_t H, uint32_t* bits, hipStream_t s);
// out[0:count) = sorted indices i with flags[i] != 0; *count_dev = count   (graph.hip, rocPRIM select)
int compact_flags(const uint8_t* flags, int64_t n, int32_t* out, int32_t* count_dev, DevBuf& tmp, hipStream_t s);
int kfac_top_planes(lgnn_ctx* h, const int64_t* idx, int64_t M, bool fork_exact, float* g, hipStream_t s);  // kfac.hip
int exclusive_scan_i32(int32_t* in, int32_t* out, int64_t n, DevBuf& tmp, hipStream_t s);  // graph.hip, rocPRIM

// ---- graph.hip -----------------------------------------------------------------------
int graph_build(lgnn_ctx* h, const int64_t* edge_index, int64_t E, hipStream_t s);
int graph_values(lgnn_ctx* h, bool same, hipStream_t s);
int graph_update(lgnn_ctx* h, const int64_t* fi, const int64_t* fj, const uint8_t* st, int64_t K, hipStream_t s);

// ---- kernels (launchers) -------------------------------------------------------------
// out[r, 0:width) = sum_j val[j] * in[col[j], 0:width)   for r in [0, nrows)
// epilogue: 0 none, 1 relu, 2 tanh
int launch_spmm(const Csr& m, int64_t nrows, const float* in, int64_t in_ld, float* out, int64_t out_ld,
                int64_t width, int epilogue, hipStream_t s, const int32_t* long_rows = nullptr, int64_t n_long = 0);
// rowsum of the CSR values
int launch_csr_rowsum(const Csr& m, int64_t nrows, float* out, hipStream_t s);

// C[R, Nout] = A[R, K] @ B[K, Nout] (+ bias[Nout]) (* dact(Hact[r / rows_per_node]))
struct GemmEpilogue {
  const float* bias = nullptr;
  const float* hact = nullptr;  // activation output of the layer below, [N, Nout] (ld = hact_ld)
  int64_t hact_ld = 0;
  int act = 0;                  // LGNN_ACT_*
  int64_t hact_row_mod = 0;     // hact row = r % hact_row_mod (planes are [c][n][w]); 0 -> r
  int out_act = -1;             // apply activation to the result itself (-1 none)
  const uint8_t* row_active = nullptr;  // optional [hact rows]: rows flagged 0 are neither computed on nor written
};
int launch_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t R,
                int64_t K, int64_t Nout, const GemmEpilogue& ep, hipStream_t s);
// C[0 : R) = alpha A[0 : R) B with R = *r_dev * r_mul (<= rows_bound) read on the device
int launch_gemm_devrows(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t rows_bound,
                        const int32_t* r_dev, int64_t r_mul, int64_t K, int64_t Nout, float alpha, hipStream_t s);
// the same on the listed nodes of every plane ([plane][plane_rows][.] in A and C), list length read on the device
int launch_gemm_listed(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t planes,
                       int64_t plane_rows, const int32_t* list, const int32_t* na_dev, int64_t K, int64_t Nout,
                       const GemmEpilogue& ep, hipStream_t s);

// scratch[D, D] (upper sub-tiles) += X[0:R, col0:col0+D)^T X[...]; X row-major with ld
int launch_gram(const float* X, int64_t ld, int64_t R, int64_t D, float* scratch, hipStream_t s);
// out[i,j] += scale * (X^T X)[i,j] on the upper sub-tiles, without atomics when the rows are not split
int launch_gram_scaled(const float* X, int64_t ld, int64_t R, int64_t D, float* out, float scale, hipStream_t s);
// nz weighted Grams over the same rows: out[z] += scale * zscale[z] * X^T diag(row_scale[z]^2) X (upper sub-tiles)
int launch_gram_batched(const float* X, int64_t ld, int64_t R, int64_t D, float* out, int64_t out_zstride, int64_t nz,
                        const float* row_scale, const float* zscale, float scale, hipStream_t s);
// row major Sb[Q, D1] = Wq[Q, M] * Phi[M, D1] (rocBLAS; jacobian.hip)
int ll_bias_gemm(const float* Wq, const float* Phi, float* Sb, int64_t Q, int64_t M, int64_t D1, int64_t ldp, hipStream_t s,
                 float beta = 0.f);  // Sb = Wq Phi + beta Sb
// lower triangle <- upper triangle
int launch_symmetrize_upper(float* H, int64_t D, hipStream_t s);
// out[i,j] += scale * scratch[min(i,j), max(i,j)]
int launch_sym_accumulate(const float* scratch, int64_t D, float scale, float* out, hipStream_t s, int64_t scratch_ld = 0);
// fused: rows r=(plane, n): y = sum_j val*in_plane[col[j]]; scratch += y^T y; optional store of y
int launch_spmm_ex(const SpmmArgs& a, int64_t nplanes, hipStream_t s);
int launch_spmm_gram_ex(const FusedArgs& a, hipStream_t s);
int launch_spmm_gram256(const FusedArgs& a, hipStream_t s);  // fused256.hip
// `rows`: rows per plane incl. a spare row where one exists; planes wider than 128 columns go through 32-bit buffer
// offsets (fused256.hip), so rows * in_ld * 4 must stay below 4 GiB there -- otherwise the unfused path takes over
bool fused_supported(int64_t width, int64_t in_ld, int64_t in_plane_stride, const void* in, int64_t rows);
int launch_spmm_gram(const Csr& m, int64_t nrows, int64_t nplanes, const float* in, float* store_or_null,
                     int64_t width, float* scratch, hipStream_t s);

int launch_transpose(const float* in, int64_t rows, int64_t cols, float* out, hipStream_t s);
int launch_act_deriv(const float* h, int64_t ld, int64_t N, int64_t H, int act, float* out, hipStream_t s);
int launch_fill_i32(int32_t* p, int64_t n, int32_t v, hipStream_t s);
int launch_gather_rows(const float* in, int64_t ld, int64_t nrows_in, const int64_t* idx, int64_t M, int64_t width,
                       float* out, int* bad_flag, hipStream_t s);

// ---- resnorm.hip: row-local pieces of models with res / norm (gnn/models/base_gnn.py:141-149) ------------------
// h[n] = act(gamma * xhat[n] + beta), xhat = (s[n] - mean) * rstd; LayerNorm: per-row statistics (saved in rstd [N]);
// eval-mode BatchNorm1d: running statistics (rstd [width], written here too)
int launch_norm_forward(lgnn_ctx* h, int layer, const float* spre, int64_t s_ld, float* out, int64_t out_ld, hipStream_t s);
// U [rows, width] (rows = planes * N, node = row % N), in place:
//   t = U (+ add);  if mask: t *= act'(h_{layer+1}[node]);  U = norm_backward_layer(t)
int launch_resnorm_backward(lgnn_ctx* h, int layer, float* U, int64_t ld, int64_t rows, const float* add, bool mask,
                            hipStream_t s);
int build_sage_res_weights(lgnn_ctx* h, int layer, hipStream_t s);  // Wcomb / bcomb of a GraphSAGE layer with res
// ---- longrows.hip -----------------------------------------------------------------------
constexpr int kLongRow = 64;
constexpr int kTopSlice = 128;  // stored entries of a hub row per top-layer task  // rows of P^T with more stored entries leave the fused kernel's per-wave gather
int long_rows_ensure(lgnn_ctx* h, hipStream_t s);  // builds h->long_* once (synchronises the stream that one time)
int long_rows_fwd_ensure(lgnn_ctx* h, hipStream_t s);  // the list of long rows of P (forward SpMMs)
// hub[plane][slot][0:width) = sum_j val[j] * in[plane][col[j]][0:width) for the long rows of P^T
int launch_long_rows_spmm(lgnn_ctx* h, const float* val, const float* in, int64_t in_ld, int64_t in_plane_stride,
                          int64_t nplanes, int64_t width, hipStream_t s);
// ---- kfac.hip ---------------------------------------------------------------------------
struct KfacPlan {
  bool seeds_on_the_fly;       // GCN top layer rebuilds the seed blocks from probabilities + logits
  bool sage_compact;           // GraphSAGE top level over the batch nodes only (compacted backward GEMM + fused MODE 1)
  bool need_pong;              // the second plane buffer is written by some step
  bool paths;                  // 2-layer GCN, ReLU, 128 < H <= 256, C <= 64: B_0 from the batch's 2-hop paths (paths.hip),
                               // no class planes, no backward GEMM, no gather of planes
  bool fuse[kMaxLayers];       // step l (l = L-1 .. 1): fused SpMM^T -> Gram kernel (else SpMM + Gram through HBM)
  bool backgemm[kMaxLayers];   // step l: compacted producer / consumer backward GEMM (else the generic GEMM)
  int64_t maxw, cc_max;        // widest GEMM output of the lower layers; class planes per chunk under the workspace cap
};
KfacPlan plan_kfac(int kind, int L, int64_t N, int64_t nnz, const int64_t* dims, int act, bool no_fuse, int64_t ws_limit,
                   bool no_paths = false);
// empirical / Monte-Carlo Fisher variant of the KFAC accumulate (curvlinops/kfac.py:663-674)
struct KfacFisherOpts {
  const void* y_seed;   // labels int64 [M] (classification) / fp32 targets [M, C] (regression) the gradient seed uses
  float resid_scale;    // seed = resid_scale * d loss_n / d f_n  (p - onehot resp. f - y)
  float b_scale;        // B_out += b_scale * g^T g               (1 / mc_samples)
  bool add_loss_and_A;  // this call also adds the loss of the true labels `y` and the A increment
};
struct KfacShare { int64_t begin, end, count; };  // parts [begin, end) of `count` equal parts of a batch (lgnn_kfac_accumulate_share)
int kfac_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train, uint32_t flags,
                    int64_t class_begin, int64_t class_end, float* const* A_out, float* const* B_out, float* loss_out,
                    hipStream_t s, const KfacFisherOpts* fisher = nullptr, const KfacShare* share = nullptr);
int batch_prologue(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, bool want_seeds, bool fork_exact,
                   float* loss_out, hipStream_t s, const void* y_seed = nullptr, float resid_scale = 1.0f);
int batch_epilogue(lgnn_ctx* h, const int64_t* idx, int64_t M, hipStream_t s);
// timing hook (lgnn_enable_kernel_timing): one HIP event on the launch stream; callers bracket the dominant kernel
int record_event(lgnn_ctx* h, hipStream_t s);
// ---- paths.hip ----------------------------------------------------------------------------
bool paths_supported(int kind, int L, const int64_t* dims, int act, int64_t nnz);
int two_hop_ensure(lgnn_ctx* h, hipStream_t s);   // h->two_hop = 2-hop paths of the graph, counted once (one synchronisation)
bool paths_pay(const lgnn_ctx* h, int64_t M);     // expected paths per destination node of a batch of M small enough
// scratch [H, H] += B_0 of this batch's class columns [cb, ce) (seed_mode: 0 upstream, 1 fork exact, 2 regression)
// (nb, ne: the destination nodes whose Y_n^T Y_n this call adds -- B_0 is a sum over nodes: the multi-GPU cut of these routes)
int kfac_paths_first_layer(lgnn_ctx* h, const int64_t* idx, int64_t M, int seed_mode, int64_t cb, int64_t ce, float* scratch,
                           hipStream_t s, int64_t nb = 0, int64_t ne = -1);
int kfac_paths_first_layer_sage(lgnn_ctx* h, const int64_t* idx, int64_t M, int seed_mode, int64_t cb, int64_t ce,
                                float* scratch, hipStream_t s, int64_t nb = 0, int64_t ne = -1);  // GraphSAGE: one-hop paths through the same fused kernel
// scratch [width, width] (upper 32 x 32 sub-tiles) += Y^T Y for rows of `width` floats (row stride ld), 128 < width <= 256:
// all eight waves of a persistent workgroup per CU on the matrix pipes, row blocks by LDS-DMA (paths.hip)
int launch_gram256_stream(const float* Y, int64_t ld, int64_t rows, int64_t width, float* scratch, hipStream_t s,
                          const int32_t* gate = nullptr, int64_t gate_cap = 0);
// ---- forward.hip ------------------------------------------------------------------------
int forward_ensure(lgnn_ctx* h, hipStream_t s);
int forward_ensure_grams(lgnn_ctx* h, hipStream_t s);
int forward_ensure_aux(lgnn_ctx* h, hipStream_t s);
int forward_input_view(lgnn_ctx* h, hipStream_t s);  // GCN: lin_in_p[0] / lin_in_ld[0] (X or its padded copy)
int build_px(lgnn_ctx* h, hipStream_t s);            // rowsum(P) and (GCN) [P X | rowsum(P) | 0]: graph and X only
int ensure_wt(lgnn_ctx* h, hipStream_t s);           // transposed weights (forward GEMM, adjacency gradient)
// gcn2_forward.hip: small plain 2-layer GCN, forward + auxiliary products through the cached P X
bool gcn2_small_forward_supported(const lgnn_ctx* h);
int gcn2_forward_through_px(lgnn_ctx* h, hipStream_t s);
// ---- diag.hip ---------------------------------------------------------------------------
int diag_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags, float* diag_out,
                    float* loss_out, hipStream_t s);
int lastlayer_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out,
                              float* loss_out, hipStream_t s);
int lastlayer_pairs_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* S, float* Sb,
                               float* loss_out, hipStream_t s);
int lastlayer_features(lgnn_ctx* h, const int64_t* idx, int64_t M, float* out, float* f_out, hipStream_t s);
int lastlayer_pairs_place(lgnn_ctx* h, const float* S, const float* Sb, float* H_out, hipStream_t s);
int full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out, float* loss_out, hipStream_t s);
int ef_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss, int64_t M, float resid_scale,
                  float scale, float* diag_out, float* full_out, float* grads_out, float* loss_out, hipStream_t s);
// ---- jacobian.hip -----------------------------------------------------------------------
int jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, hipStream_t s);

}  // namespace lgnn
