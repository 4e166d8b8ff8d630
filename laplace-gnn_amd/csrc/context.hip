// Context management, model binding and the cached full-graph forward pass.
//
// The reference recomputes the dense-adjacency forward (incl. normalize_adj) three times per
// mini-batch (curvlinops/kfac.py:576, laplace/curvature/curvlinops.py:106, laplace/baselaplace.py:832)
// although it does not depend on the batch: model(x_indices) computes all N node outputs and indexes
// them (gnn/models/base_gnn.py:136-161).  Here it runs once per weight version and is cached,
// together with the raw input Grams in_l^T in_l of the KFAC A factors (kfac.py:819-875), which are
// batch independent for the same reason.
#include "lgnn_internal.h"

namespace lgnn {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

int DevBuf::reserve(size_t want) {
  if (want <= bytes && p) return 0;
  if (p) {
    hipError_t e = hipFree(p);
    p = nullptr;
    bytes = 0;
    if (e != hipSuccess) { set_error(std::string("hipFree: ") + hipGetErrorString(e)); return 1; }
  }
  size_t sz = std::max<size_t>(want, 256);
  hipError_t e = hipMalloc(&p, sz);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc(" + std::to_string(sz) + " B): " + hipGetErrorString(e));
    return 1;
  }
  bytes = sz;
  return 0;
}
// Pinned, device-visible host memory (the sticky error flags: kernels store to it only when something is wrong, the
// host reads it after a stream synchronisation -- no device-to-host copy, no clearing launch)
int DevBuf::reserve_host(size_t want) {
  if (want <= bytes && p && host) return 0;
  release();
  size_t sz = std::max<size_t>(want, 256);
  hipError_t e = hipHostMalloc(&p, sz, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipHostMalloc(" + std::to_string(sz) + " B): " + hipGetErrorString(e));
    return 1;
  }
  host = true;
  bytes = sz;
  return 0;
}
void DevBuf::release() {
  if (p) (void)(host ? hipHostFree(p) : hipFree(p));
  p = nullptr;
  bytes = 0;
  host = false;
}

// GCN: what the first Linear reads.  X itself, or -- unaligned feature rows -- a zero padded copy made once per binding
int forward_input_view(lgnn_ctx* h, hipStream_t s) {
  ForwardCache& fc = h->fc;
  const int64_t N = h->N;
  fc.lin_in_p[0] = h->X;
  fc.lin_in_ld[0] = h->dims[0];
  if (h->dims[0] % 4 != 0) {
    const int64_t ldx = cdiv(h->dims[0], 4) * 4;
    if (!fc.x_valid) {
      LGNN_CALL(fc.Xpad.reserve(size_t(N) * ldx * 4));
      LGNN_HIP_CHECK(hipMemsetAsync(fc.Xpad.p, 0, size_t(N) * ldx * 4, s));
      LGNN_HIP_CHECK(hipMemcpy2DAsync(fc.Xpad.p, size_t(ldx) * 4, h->X, size_t(h->dims[0]) * 4, size_t(h->dims[0]) * 4,
                                      size_t(N), hipMemcpyDeviceToDevice, s));
    }
    fc.lin_in_p[0] = fc.Xpad.as<float>();
    fc.lin_in_ld[0] = ldx;
  }
  return 0;
}

__global__ void set_column_kernel(float* __restrict__ A, int64_t ld, int64_t col, const float* __restrict__ x, int64_t n) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n) A[r * ld + col] = x[r];
}

// What depends on the graph and the features only (ForwardCache::px_valid): rowsum(P) and, for a GCN, the first Linear's
// input seen from an output node WITH its bias entry, E = [P X | rowsum(P) | 0 ..] as one matrix of row stride
// round_up(F + 1, 4) -- the closed-form diagonal's matrix-core kernel copies a row block with one 16-byte-per-lane DMA
// (diag.hip); everybody else reads the first F columns.
int build_px(lgnn_ctx* h, hipStream_t s) {
  ForwardCache& fc = h->fc;
  const int64_t N = h->N;
  LGNN_CALL(fc.rowsum.reserve(size_t(N) * 4));
  LGNN_CALL(launch_csr_rowsum(h->P, N, fc.rowsum.as<float>(), s));
  if (h->kind != LGNN_KIND_GCN) return 0;
  const int64_t F = h->dims[0], d_in = fc.lin_in_ld[0], ldE = cdiv(F + 1, 4) * 4;
  fc.prop_ld[0] = ldE;
  LGNN_CALL(fc.prop_in[0].reserve(size_t(N) * ldE * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(fc.prop_in[0].p, 0, size_t(N) * ldE * 4, s));
  // (the padded width of an unaligned X is propagated as a whole: its padding columns are zero)
  LGNN_CALL(launch_spmm(h->P, N, fc.lin_in_p[0], d_in, fc.prop_in[0].as<float>(), ldE, d_in, 0, s));
  hipLaunchKernelGGL(set_column_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, fc.prop_in[0].as<float>(), ldE, F,
                     fc.rowsum.as<float>(), N);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// W_l^T [in_l, out_l]: the forward GEMM's operand (and the adjacency gradient's); built once per weight version
int ensure_wt(lgnn_ctx* h, hipStream_t s) {
  if (h->fc.wt_valid) return 0;
  for (int l = 0; l < h->L; ++l) {
    LGNN_CALL(h->Wt[l].reserve(size_t(h->in_dim[l]) * h->dims[l + 1] * 4));
    const float* W = h->W[l];
    if (h->has_res && l < h->L - 1) {
      if (h->kind == LGNN_KIND_SAGE) {
        LGNN_CALL(build_sage_res_weights(h, l, s));
        W = h->Wcomb[l].as<float>();
      } else {
        LGNN_CALL(h->Wrt[l].reserve(size_t(h->dims[l]) * h->dims[l + 1] * 4));
        LGNN_CALL(launch_transpose(h->Wr[l], h->dims[l + 1], h->dims[l], h->Wrt[l].as<float>(), s));
      }
    }
    LGNN_CALL(launch_transpose(W, h->dims[l + 1], h->in_dim[l], h->Wt[l].as<float>(), s));
  }
  h->fc.wt_valid = true;
  return 0;
}

static int gcn_or_sage_forward(lgnn_ctx* h, hipStream_t s) {
  ForwardCache& fc = h->fc;
  const int64_t N = h->N;
  const int L = h->L;
  // hub rows of the forward matrix go to a whole workgroup each (list built once per graph)
  LGNN_CALL(long_rows_fwd_ensure(h, s));
  const int32_t* lr = h->n_long_fwd > 0 ? (h->P.rowptr == h->PT.rowptr ? h->long_rows.as<int32_t>() : h->long_rows_fwd.as<int32_t>())
                                         : nullptr;
  const int64_t nlr = h->n_long_fwd > 0 ? h->n_long_fwd : 0;
  int64_t maxw = 0;
  for (int l = 0; l < L; ++l) maxw = std::max(maxw, h->dims[l + 1]);
  LGNN_CALL(fc.out.reserve(size_t(N) * h->dims[L] * 4));

  if (h->kind == LGNN_KIND_GCN) {
    LGNN_CALL(fc.tmp.reserve(size_t(N) * maxw * 4));
    LGNN_CALL(forward_input_view(h, s));
    for (int l = 0; l < L; ++l) {
      const int64_t din = h->dims[l], dout = h->dims[l + 1];
      GemmEpilogue ep;
      ep.bias = h->b[l];
      // Z_l = h_l W_l^T + b_l   (nn.Linear inside GCNConv, gnn/models/layers.py:45-46)
      LGNN_CALL(launch_gemm(fc.lin_in_p[l], fc.lin_in_ld[l], h->Wt[l].as<float>(), dout, fc.tmp.as<float>(), dout, N,
                            din, dout, ep, s));
      if (l < L - 1) {
        LGNN_CALL(fc.act_out[l].reserve(size_t(N) * dout * 4));
        if (!h->extras()) {
          // h_{l+1} = act(A_hat Z_l)  (norm = Identity, dropout = identity in eval; base_gnn.py:141-156)
          LGNN_CALL(launch_spmm(h->P, N, fc.tmp.as<float>(), dout, fc.act_out[l].as<float>(), dout, dout,
                                h->act == LGNN_ACT_RELU ? 1 : 2, s, lr, nlr));
        } else {
          // s = res_l(h_l) + A_hat Z_l (base_gnn.py:141-144); n = norms[l](s) (:148); h_{l+1} = act(n) (:151)
          const bool nrm = h->norm != LGNN_NORM_NONE;
          SpmmArgs a{};
          a.rowptr = h->P.rowptr; a.col = h->P.col; a.val = h->P.val; a.nrows = N;
          a.in = fc.tmp.as<float>(); a.in_ld = dout; a.width = dout;
          a.long_rows = lr; a.n_long = nlr;
          if (h->has_res) {
            LGNN_CALL(fc.res_out.reserve(size_t(N) * maxw * 4));
            GemmEpilogue er;
            er.bias = h->br[l];
            LGNN_CALL(launch_gemm(fc.lin_in_p[l], fc.lin_in_ld[l], h->Wrt[l].as<float>(), dout, fc.res_out.as<float>(), dout,
                                  N, din, dout, er, s));
            a.self = fc.res_out.as<float>(); a.self_ld = dout;
          }
          if (nrm) {
            LGNN_CALL(fc.pre_norm.reserve(size_t(N) * maxw * 4));
            a.out = fc.pre_norm.as<float>(); a.out_ld = dout; a.out_act = -1;
            LGNN_CALL(launch_spmm_ex(a, 1, s));
            LGNN_CALL(launch_norm_forward(h, l, fc.pre_norm.as<float>(), dout, fc.act_out[l].as<float>(), dout, s));
          } else {
            a.out = fc.act_out[l].as<float>(); a.out_ld = dout; a.out_act = h->act;
            LGNN_CALL(launch_spmm_ex(a, 1, s));
          }
        }
        fc.hact_p[l] = fc.act_out[l].as<float>();
        fc.hact_ld[l] = dout;
        if (h->act == LGNN_ACT_RELU) {
          LGNN_CALL(fc.mask_bits[l].reserve(size_t(N) * cdiv(dout, 32) * 4 + 32));  // + 32: backgemm reads 8 words per node
          LGNN_CALL(launch_relu_mask_bits(fc.hact_p[l], dout, N, dout, fc.mask_bits[l].as<uint32_t>(), s));
        }
        fc.lin_in_p[l + 1] = fc.hact_p[l];
        fc.lin_in_ld[l + 1] = dout;
      } else {
        LGNN_CALL(launch_spmm(h->P, N, fc.tmp.as<float>(), dout, fc.out.as<float>(), dout, dout, 0, s, lr, nlr));
      }
    }
  } else {
    // GraphSAGE: cat_l = [h_l | A_bar h_l]  (gnn/models/layers.py:26-29)
    for (int l = 0; l < L; ++l) {
      const int64_t d = h->dims[l];
      LGNN_CALL(fc.lin_in[l].reserve(size_t(N) * 2 * d * 4));
      fc.lin_in_p[l] = fc.lin_in[l].as<float>();
      fc.lin_in_ld[l] = 2 * d;
    }
    LGNN_HIP_CHECK(hipMemcpy2DAsync(fc.lin_in[0].p, size_t(2 * h->dims[0]) * 4, h->X, size_t(h->dims[0]) * 4,
                                    size_t(h->dims[0]) * 4, size_t(N), hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < L; ++l) {
      const int64_t d = h->dims[l], dout = h->dims[l + 1];
      float* cat = fc.lin_in[l].as<float>();
      SpmmArgs a{};
      a.rowptr = h->P.rowptr; a.col = h->P.col; a.val = h->P.val; a.nrows = N;
      a.in = cat; a.in_ld = 2 * d; a.out = cat + d; a.out_ld = 2 * d; a.width = d; a.out_act = -1;
      a.long_rows = lr; a.n_long = nlr;
      LGNN_CALL(launch_spmm_ex(a, 1, s));
      GemmEpilogue ep;
      ep.bias = h->b[l];
      if (l < L - 1) {
        float* nxt = fc.lin_in[l + 1].as<float>();
        // with res: Wt[l] holds (W_l + [Wr_l | 0])^T and the bias is b_l + br_l (res_l and the conv's Linear read the same
        // rows, base_gnn.py:141-144); with a norm the activation moves behind it (:148-151)
        if (h->has_res) ep.bias = h->bcomb[l].as<float>();
        if (h->norm != LGNN_NORM_NONE) {
          LGNN_CALL(fc.pre_norm.reserve(size_t(N) * maxw * 4));
          LGNN_CALL(launch_gemm(cat, 2 * d, h->Wt[l].as<float>(), dout, fc.pre_norm.as<float>(), dout, N, 2 * d, dout, ep, s));
          LGNN_CALL(launch_norm_forward(h, l, fc.pre_norm.as<float>(), dout, nxt, 2 * dout, s));
        } else {
          ep.out_act = h->act;
          LGNN_CALL(launch_gemm(cat, 2 * d, h->Wt[l].as<float>(), dout, nxt, 2 * dout, N, 2 * d, dout, ep, s));
        }
        fc.hact_p[l] = nxt;
        fc.hact_ld[l] = 2 * dout;
        if (h->act == LGNN_ACT_RELU) {  // bit masks of the ReLU derivative for the fused backward kernel
          LGNN_CALL(fc.mask_bits[l].reserve(size_t(N) * cdiv(dout, 32) * 4 + 32));
          LGNN_CALL(launch_relu_mask_bits(fc.hact_p[l], 2 * dout, N, dout, fc.mask_bits[l].as<uint32_t>(), s));
        }
      } else {
        LGNN_CALL(launch_gemm(cat, 2 * d, h->Wt[l].as<float>(), dout, fc.out.as<float>(), dout, N, 2 * d, dout, ep, s));
      }
    }
  }
  return 0;
}

static void forward_mark_valid(lgnn_ctx* h) {
  h->fc.valid = true;
  h->fc.aux_valid = false;
  // layer 0 of a GCN sees X itself: its input Gram and P X do not depend on the weights
  const bool keep0 = h->fc.x_valid && h->kind == LGNN_KIND_GCN;
  for (int l = 0; l < kMaxLayers; ++l) h->fc.gram_valid[l] = (l == 0 && keep0) ? h->fc.gram_valid[0] : false;
  h->fc.x_valid = true;
}

int forward_ensure(lgnn_ctx* h, hipStream_t s) {
  LGNN_REQUIRE(h->L > 0 && h->X, "no model bound (call lgnn_bind_model first)");
  if (h->fc.valid) return 0;
  if (gcn2_small_forward_supported(h)) {
    // small plain 2-layer GCN: the pass through the cached P X (gcn2_forward.hip) leaves the auxiliary products of
    // forward_ensure_aux behind as well, in four launches
    LGNN_CALL(gcn2_forward_through_px(h, s));
    forward_mark_valid(h);
    h->fc.px_valid = true;
    h->fc.aux_valid = true;
    return 0;
  }
  LGNN_CALL(ensure_wt(h, s));
  LGNN_CALL(gcn_or_sage_forward(h, s));
  forward_mark_valid(h);
  return 0;
}

int forward_ensure_grams(lgnn_ctx* h, hipStream_t s) {
  LGNN_CALL(forward_ensure(h, s));
  for (int l = 0; l < h->L; ++l) {
    if (h->fc.gram_valid[l]) continue;
    const int64_t D = h->in_dim[l];
    LGNN_CALL(h->fc.gram_raw[l].reserve(size_t(D) * D * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(h->fc.gram_raw[l].p, 0, size_t(D) * D * 4, s));
    LGNN_CALL(launch_gram(h->fc.lin_in_p[l], h->fc.lin_in_ld[l], h->N, D, h->fc.gram_raw[l].as<float>(), s));
    h->fc.gram_valid[l] = true;
  }
  return 0;
}

// rowsum(P) and P @ lin_in[l] for GCN (closed-form diagonal GGN and last-layer features)
int forward_ensure_aux(lgnn_ctx* h, hipStream_t s) {
  const bool had_x = h->fc.x_valid && h->fc.px_valid;
  LGNN_CALL(forward_ensure(h, s));
  if (h->fc.aux_valid) return 0;
  if (!had_x) LGNN_CALL(build_px(h, s));  // rowsum(P), P X: the graph and X only, kept across weight updates
  if (h->kind == LGNN_KIND_GCN) {
    for (int l = 1; l < h->L; ++l) {
      const int64_t d = h->dims[l];
      h->fc.prop_ld[l] = d;
      LGNN_CALL(h->fc.prop_in[l].reserve(size_t(h->N) * d * 4));
      LGNN_CALL(launch_spmm(h->P, h->N, h->fc.lin_in_p[l], h->fc.lin_in_ld[l], h->fc.prop_in[l].as<float>(), d, d, 0, s));
    }
  }
  h->fc.px_valid = true;
  if (h->L == 2) {  // act'(h_1), contiguous [N, H]: the first-layer diagonal kernel multiplies with it (one FMA per element)
    const int64_t H = h->dims[1];
    LGNN_CALL(h->fc.dact0.reserve(size_t(h->N) * H * 4));
    LGNN_CALL(launch_act_deriv(h->fc.hact_p[0], h->fc.hact_ld[0], h->N, H, h->act, h->fc.dact0.as<float>(), s));
  }
  h->fc.aux_valid = true;
  return 0;
}

}  // namespace lgnn

using namespace lgnn;

extern "C" int lgnn_abi_version(void) { return LGNN_ABI_VERSION; }
extern "C" const char* lgnn_last_error(void) { return lgnn::g_error.c_str(); }

extern "C" int lgnn_create(lgnn_ctx** out, int64_t num_nodes, const int64_t* edge_index, int64_t num_edges, int kind,
                           int symmetric, void* stream) {
  if (!out) { set_error("null output handle"); return 2; }
  *out = nullptr;
  LGNN_REQUIRE(kind == LGNN_KIND_GCN || kind == LGNN_KIND_SAGE, "unknown graph kind");
  LGNN_REQUIRE(num_edges == 0 || edge_index != nullptr, "edge_index is null");
  lgnn_ctx* h = new (std::nothrow) lgnn_ctx();
  if (!h) { set_error("out of host memory"); return 1; }
  h->N = num_nodes;
  h->kind = kind;
  h->sym = symmetric != 0;
  int rc = graph_build(h, edge_index, num_edges, static_cast<hipStream_t>(stream));
  if (rc == 0) {
    rc = h->ws.pos.reserve(size_t(num_nodes) * 4);
    if (rc == 0) rc = launch_fill_i32(h->ws.pos.as<int32_t>(), num_nodes, INT32_MAX, static_cast<hipStream_t>(stream));
    if (rc == 0) rc = h->ws.flags.reserve_host(64);  // sticky asynchronous error flags, zero = clean
    if (rc == 0) memset(h->ws.flags.p, 0, 64);
  }
  if (rc != 0) { lgnn_destroy(h); return rc; }
  *out = h;
  return 0;
}

#ifdef LGNN_DEV
namespace lgnn { void paths_phase_report(); }
#endif
extern "C" void lgnn_destroy(lgnn_ctx* h) {
  if (!h) return;
  (void)hipDeviceSynchronize();
#ifdef LGNN_DEV
  if (getenv("LGNN_PHASE_REPORT")) lgnn::paths_phase_report();
#endif
  DevBuf* bufs[] = {&h->A_rowptr, &h->A_col, &h->AT_rowptr, &h->AT_col, &h->val_fwd, &h->val_bwd, &h->deg_scale,
                    &h->fc.out, &h->fc.tmp, &h->fc.res_out, &h->fc.pre_norm, &h->ws.planes_c, &h->ws.path_coef, &h->ws.path_up, &h->ws.path_bg, &h->ws.path_alpha, &h->ws.adj_z0, &h->ws.adj_dir, &h->ws.path_cnt, &h->ws.path_rptr, &h->ws.path_rm, &h->ws.path_rw, &h->ws.path_zeros, &h->ws.path_pcnt, &h->ws.path_pptr, &h->ws.path_pm, &h->ws.path_pv, &h->ws.path_pw, &h->ws.path_flags, &h->ws.path_nodes, &h->ws.path_nnodes, &h->fc.rowsum, &h->fc.dact0, &h->fc.Xpad, &h->ws.pos, &h->ws.seeds, &h->ws.probs, &h->ws.mult, &h->ws.planes_a,
                    &h->ws.planes_b, &h->ws.misc, &h->ws.jac, &h->long_rows, &h->long_slot, &h->long_tasks, &h->hub, &h->long_rows_fwd, &h->top_multi, &h->top_tasks, &h->top_task_count, &h->top_cnt, &h->top_offs, &h->top_hub_tiles, &h->ws.top, &h->ws.flags, &h->ws.out_flags, &h->ws.out_list, &h->ws.out_count, &h->ws.val_act2, &h->ws.active, &h->ws.val_act, &h->ws.act_list, &h->ws.act_count, &h->ws.select_tmp};
  for (DevBuf* b : bufs) b->release();
  for (int l = 0; l < kMaxLayers; ++l) {
    h->Wt[l].release(); h->fc.lin_in[l].release(); h->fc.act_out[l].release(); h->fc.gram_raw[l].release();
    h->fc.prop_in[l].release(); h->ws.gram_scratch[l].release(); h->fc.mask_bits[l].release();
    h->Wrt[l].release(); h->Wcomb[l].release(); h->bcomb[l].release(); h->fc.xhat[l].release(); h->fc.rstd[l].release();
    h->ws.gram_scratch_res[l].release();
  }
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  delete h;
}

extern "C" int64_t lgnn_nnz(const lgnn_ctx* h) { return h ? h->nnz : -1; }
extern "C" int64_t lgnn_num_nodes(const lgnn_ctx* h) { return h ? h->N : -1; }
extern "C" int lgnn_is_symmetric(const lgnn_ctx* h) { return h && h->sym ? 1 : 0; }
extern "C" int64_t lgnn_num_long_rows(const lgnn_ctx* h) { return h ? h->n_long : -1; }
extern "C" int lgnn_kfac_last_route(const lgnn_ctx* h) { return h && h->last_route_paths ? 1 : 0; }

extern "C" int lgnn_bind_model(lgnn_ctx* h, int num_layers, const int64_t* dims, const float* const* weights,
                               const float* const* biases, const float* X, int activation, int likelihood) {
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(num_layers >= 1 && num_layers <= kMaxLayers, "num_layers out of range");
  LGNN_REQUIRE(dims && weights && biases && X, "null model pointer");
  LGNN_REQUIRE(activation == LGNN_ACT_RELU || activation == LGNN_ACT_TANH, "unsupported activation");
  LGNN_REQUIRE(likelihood == LGNN_LIK_CLASSIFICATION || likelihood == LGNN_LIK_REGRESSION, "unsupported likelihood");
  h->L = num_layers;
  h->n_params = 0;
  for (int l = 0; l <= num_layers; ++l) {
    LGNN_REQUIRE(dims[l] > 0, "layer width must be positive");
    h->dims[l] = dims[l];
  }
  for (int l = 0; l < num_layers; ++l) {
    LGNN_REQUIRE(weights[l] && biases[l], "null weight / bias pointer (bias=False is not supported)");
    h->W[l] = weights[l];
    h->b[l] = biases[l];
    h->in_dim[l] = h->kind == LGNN_KIND_SAGE ? 2 * dims[l] : dims[l];
    h->n_params += h->in_dim[l] * dims[l + 1] + dims[l + 1];
  }
  h->X = X;
  h->act = activation;
  h->lik = likelihood;
  h->has_res = false;  // lgnn_bind_extras brings res / norm back
  h->norm = LGNN_NORM_NONE;
  // the compact GraphSAGE top level keeps planes_a all zero outside the batch rows for ONE plane layout; a new
  // binding (other widths) must not inherit that claim: the old layout's spare rows hold stale backward-GEMM stores
  h->ws.planes_a_zero_ptr = nullptr;
  h->ws.planes_a_zero_bytes = 0;
  h->fc.x_valid = false;  // a new X / new widths: nothing survives
  h->fc.px_valid = false;
  h->fc.gram_valid[0] = false;
  return lgnn_invalidate(h);
}

// Weights changed: the forward pass, the activations' Grams and everything derived from them are stale.  What depends on
// the graph and X only (padded X, rowsum(P), P X, X^T X of a GCN's first layer) is kept; lgnn_bind_model drops that too.
extern "C" int lgnn_invalidate(lgnn_ctx* h) {
  if (!h) { set_error("null context"); return 2; }
  h->fc.valid = false;
  h->fc.aux_valid = false;
  h->fc.wt_valid = false;
  const bool keep0 = h->fc.x_valid && h->kind == LGNN_KIND_GCN;
  for (int l = 0; l < kMaxLayers; ++l) h->fc.gram_valid[l] = (l == 0 && keep0) ? h->fc.gram_valid[0] : false;
  return 0;
}

extern "C" int lgnn_set_workspace_limit(lgnn_ctx* h, int64_t bytes) {
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(bytes >= (int64_t(1) << 20), "workspace limit too small");
  h->ws_limit = bytes;
  return 0;
}

extern "C" int64_t lgnn_device_bytes(const lgnn_ctx* h) {
  if (!h) return -1;
  size_t t = 0;
  const DevBuf* bufs[] = {&h->A_rowptr, &h->A_col, &h->AT_rowptr, &h->AT_col, &h->val_fwd, &h->val_bwd, &h->deg_scale,
                          &h->fc.out, &h->fc.tmp, &h->fc.res_out, &h->fc.pre_norm, &h->ws.planes_c, &h->ws.path_coef, &h->ws.path_up, &h->ws.path_bg, &h->ws.path_alpha, &h->ws.adj_z0, &h->ws.adj_dir, &h->ws.path_cnt, &h->ws.path_rptr, &h->ws.path_rm, &h->ws.path_rw, &h->ws.path_zeros, &h->ws.path_pcnt, &h->ws.path_pptr, &h->ws.path_pm, &h->ws.path_pv, &h->ws.path_pw, &h->ws.path_flags, &h->ws.path_nodes, &h->ws.path_nnodes, &h->fc.rowsum, &h->fc.dact0, &h->fc.Xpad, &h->ws.pos, &h->ws.seeds, &h->ws.probs, &h->ws.mult,
                          &h->ws.planes_a, &h->ws.planes_b, &h->ws.misc, &h->ws.jac, &h->long_rows, &h->long_slot, &h->long_tasks, &h->hub, &h->long_rows_fwd, &h->top_multi, &h->top_tasks, &h->top_task_count, &h->top_cnt, &h->top_offs, &h->top_hub_tiles, &h->ws.top, &h->ws.flags, &h->ws.out_flags, &h->ws.out_list, &h->ws.out_count, &h->ws.val_act2, &h->ws.active, &h->ws.val_act, &h->ws.act_list, &h->ws.act_count, &h->ws.select_tmp};
  for (const DevBuf* b : bufs) t += b->bytes;
  for (int l = 0; l < kMaxLayers; ++l)
    t += h->Wt[l].bytes + h->fc.lin_in[l].bytes + h->fc.act_out[l].bytes + h->fc.gram_raw[l].bytes +
         h->fc.prop_in[l].bytes + h->ws.gram_scratch[l].bytes + h->fc.mask_bits[l].bytes + h->Wrt[l].bytes +
         h->Wcomb[l].bytes + h->bcomb[l].bytes + h->fc.xhat[l].bytes + h->fc.rstd[l].bytes + h->ws.gram_scratch_res[l].bytes;
  return int64_t(t);
}

extern "C" int lgnn_forward_all(lgnn_ctx* h, float* out, void* stream) {
  if (!h || !out) { set_error("null argument"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  LGNN_CALL(forward_ensure(h, s));
  LGNN_HIP_CHECK(hipMemcpyAsync(out, h->fc.out.p, size_t(h->N) * h->dims[h->L] * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

extern "C" int lgnn_forward(lgnn_ctx* h, const int64_t* idx, int64_t M, float* out, void* stream) {
  if (!h || (M > 0 && (!idx || !out))) { set_error("null argument"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  LGNN_CALL(forward_ensure(h, s));
  // out-of-range node ids read as zero rows and raise the asynchronous flag (lgnn_check_async_errors)
  return launch_gather_rows(h->fc.out.as<float>(), h->dims[h->L], h->N, idx, M, h->dims[h->L], out,
                            h->ws.flags.as<int>() + 2, s);
}

extern "C" int lgnn_check_async_errors(lgnn_ctx* h, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!h->ws.flags.p) return 0;
  // the flags live in pinned host memory the kernels store to directly: wait for the stream, read, clear
  LGNN_HIP_CHECK(hipStreamSynchronize(s));
  int flags[4];
  volatile int* dev = h->ws.flags.as<int>();
  for (int i = 0; i < 4; ++i) { flags[i] = dev[i]; dev[i] = 0; }
  LGNN_REQUIRE(flags[1] == 0 && flags[0] != 1 && flags[2] == 0, "a batch contained a node index outside [0, num_nodes)");
  LGNN_REQUIRE(flags[0] != 2, "a batch contained a label outside [0, num_classes)");
  return 0;
}

extern "C" int lgnn_peek_async_errors(lgnn_ctx* h) {
  if (!h) { set_error("null context"); return 2; }
  if (!h->ws.flags.p) return 0;
  // no synchronisation: whatever the kernels that have FINISHED stored into the pinned flags (sticky until a raising read)
  volatile int* dev = h->ws.flags.as<int>();
  int flags[4];
  for (int i = 0; i < 4; ++i) flags[i] = dev[i];
  if (flags[0] == 0 && flags[1] == 0 && flags[2] == 0) return 0;
  for (int i = 0; i < 4; ++i) dev[i] = 0;
  LGNN_REQUIRE(flags[1] == 0 && flags[0] != 1 && flags[2] == 0, "a batch contained a node index outside [0, num_nodes)");
  LGNN_REQUIRE(flags[0] != 2, "a batch contained a label outside [0, num_classes)");
  return 0;
}

extern "C" int lgnn_enable_kernel_timing(lgnn_ctx* h, int enable) {
  if (!h) { set_error("null context"); return 2; }
  h->timing = enable != 0;
  h->ev_used = 0;
  h->ev_planes = 0;
  return 0;
}

extern "C" int lgnn_kernel_timing_read(lgnn_ctx* h, int64_t* launches, double* total_ms, int64_t* planes) {
  if (!h || !launches || !total_ms || !planes) { set_error("null argument"); return 2; }
  double tot = 0.0;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    LGNN_HIP_CHECK(hipEventSynchronize(h->ev[i + 1]));
    float ms = 0.f;
    LGNN_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    tot += ms;
  }
  *launches = int64_t(h->ev_used / 2);
  *total_ms = tot;
  *planes = h->ev_planes;
  return 0;
}

extern "C" int lgnn_kernel_timing_launches(lgnn_ctx* h, double* ms_out, int64_t capacity, int64_t* launches) {
  if (!h || !launches || (capacity > 0 && !ms_out)) { set_error("null argument"); return 2; }
  const int64_t n = int64_t(h->ev_used / 2);
  *launches = n;
  for (int64_t i = 0; i < n && i < capacity; ++i) {
    LGNN_HIP_CHECK(hipEventSynchronize(h->ev[2 * i + 1]));
    float ms = 0.f;
    LGNN_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
    ms_out[i] = ms;
  }
  return 0;
}

extern "C" int lgnn_kfac_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                                    uint32_t flags, float* const* A_out, float* const* B_out, float* loss_out,
                                    void* stream) {
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(h->L > 0, "no model bound");
  return kfac_accumulate(h, idx, y, M, n_train, flags, 0, h->dims[h->L], A_out, B_out, loss_out,
                         static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_kfac_accumulate_classes(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                                            uint32_t flags, int64_t class_begin, int64_t class_end,
                                            float* const* A_out, float* const* B_out, float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return kfac_accumulate(h, idx, y, M, n_train, flags, class_begin, class_end, A_out, B_out, loss_out,
                         static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_kfac_accumulate_share(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train,
                                          uint32_t flags, int64_t part_begin, int64_t part_end, int64_t part_count,
                                          float* const* A_out, float* const* B_out, float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  const KfacShare share{part_begin, part_end, part_count};
  return kfac_accumulate(h, idx, y, M, n_train, flags, 0, h->L > 0 ? h->dims[h->L] : 1, A_out, B_out, loss_out,
                         static_cast<hipStream_t>(stream), nullptr, &share);
}

extern "C" int lgnn_diag_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags,
                                    float* diag_out, float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return diag_accumulate(h, idx, y, M, flags, diag_out, loss_out, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_lastlayer_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out,
                                              float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return lastlayer_full_accumulate(h, idx, y, M, H_out, loss_out, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_ef_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss, int64_t M,
                                  float resid_scale, float scale, float* diag_out, float* full_out, float* grads_out,
                                  float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return ef_accumulate(h, idx, y_seed, y_loss, M, resid_scale, scale, diag_out, full_out, grads_out, loss_out,
                       static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out, float* loss_out,
                                    void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return full_accumulate(h, idx, y, M, H_out, loss_out, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_lastlayer_pairs_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* S_pairs,
                                               float* Sb_pairs, float* loss_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return lastlayer_pairs_accumulate(h, idx, y, M, S_pairs, Sb_pairs, loss_out, static_cast<hipStream_t>(stream));
}
extern "C" int lgnn_lastlayer_pairs_place(lgnn_ctx* h, const float* S_pairs, const float* Sb_pairs, float* H_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return lastlayer_pairs_place(h, S_pairs, Sb_pairs, H_out, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_lastlayer_features(lgnn_ctx* h, const int64_t* idx, int64_t M, float* phi_out, float* f_out, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  return lastlayer_features(h, idx, M, phi_out, f_out, static_cast<hipStream_t>(stream));
}
