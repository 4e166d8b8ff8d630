// Per-sample Jacobians  J[m][c][:] = d f[idx[m], c] / d theta  for the GLM predictive ("next" row 8(f)-3).
//
// Reference: CurvatureInterface.jacobians (laplace/curvature/curvature.py:89-130): torch.func.jacrev of the whole
// dense-adjacency model, M*C backward passes through autograd, result [M, C, P] with the parameters in
// named_parameters order, each flattened row major (weight [out, in], then bias).
//
// Here the M*C backward passes travel together as planes, exactly like the C class columns of the KFAC path
// (kfac.hip), with the per-layer contraction g_l^T in_l (one strided-batched rocBLAS GEMM per layer, written straight
// into J) in place of the Gram g_l^T g_l:
//     top    g_{L-1}[(m,c)] = P^T-scatter of e_c at node idx[m]          (GraphSAGE: e_c at idx[m] itself)
//     layer  dW_l[(m,c)] = g_l^T in_l   [out_l x in_l]      db_l[(m,c)] = column sums of g_l
//     down   g_{l-1} = P^T (act'(h_l) * (g_l W_l))          (GraphSAGE: act'(h_l) * (dcat[:, :d] + P^T dcat[:, d:]))
// The result is as large as the reference's (M*C*P floats); samples are processed in chunks that fit the workspace cap.
#include <rocblas/rocblas.h>

#include "lgnn_internal.h"

namespace lgnn {

void* blas_handle(hipStream_t s);  // eigh.hip: the process-wide rocBLAS handle, bound to `s`

namespace {

// GCN: planes[(m, c)][v][c] = P[idx[m], v] for the entries v of row idx[m] of P; one wave per sample.
__global__ __launch_bounds__(256) void jac_top_gcn_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int64_t C,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, const float* __restrict__ val,
                                                          float* __restrict__ g, int* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) { if (lane == 0) bad[1] = 1; return; }
  const int32_t e = rowptr[n + 1];
  for (int32_t p = rowptr[n] + lane; p < e; p += 64) {
    const int64_t v = col[p];
    const float x = val[p];
    for (int64_t c = 0; c < C; ++c) g[((m * C + c) * N + v) * C + c] = x;
  }
}
// GraphSAGE: the last Linear's output is not propagated
__global__ void jac_top_sage_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int64_t C,
                                    float* __restrict__ g, int* __restrict__ bad) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= M * C) return;
  const int64_t m = q / C, c = q - m * C;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) { bad[1] = 1; return; }
  g[(q * N + n) * C + c] = 1.f;
}

// out[p * out_stride + o] = sum_v g[p][v][o]; one workgroup per plane
__global__ __launch_bounds__(256) void plane_colsum_kernel(const float* __restrict__ g, int64_t N, int64_t D,
                                                           float* __restrict__ out, int64_t out_stride) {
  const int64_t p = blockIdx.x;
  const float* __restrict__ gp = g + p * N * D;
  for (int64_t o = threadIdx.x; o < D; o += blockDim.x) {
    float acc = 0.f;
    for (int64_t v = 0; v < N; ++v) acc += gp[v * D + o];
    out[p * out_stride + o] = acc;
  }
}

}  // namespace

// row major C[Q, D1] = A[Q, M] * B[M, D1] (ld ldp): the bias column of the last-layer pair Grams (diag.hip)
int ll_bias_gemm(const float* Wq, const float* Phi, float* Sb, int64_t Q, int64_t M, int64_t D1, int64_t ldp,
                 hipStream_t s, float beta) {
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  const float one = 1.f, zero = beta;
  // column-major view: C^T[D1, Q] = B^T[D1, M] * A^T[M, Q]
  const rocblas_status st = rocblas_sgemm(blas, rocblas_operation_none, rocblas_operation_none, rocblas_int(D1),
                                          rocblas_int(Q), rocblas_int(M), &one, Phi, rocblas_int(ldp), Wq, rocblas_int(M),
                                          &zero, Sb, rocblas_int(D1));
  if (st != rocblas_status_success) { set_error("rocblas_sgemm failed"); return 3; }
  return 0;
}

// out[P, P] += scale * G^T G for row-major G [rows, P] (the full empirical / MC Fisher of a chunk of per-sample gradients)
int gram_rows_sgemm(const float* G, int64_t rows, int64_t P, float scale, float* out, hipStream_t s) {
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  const float one = 1.f;
  // column-major view of G: [P, rows] with ld = P;  out = G_cm G_cm^T (symmetric: row / column major coincide)
  const rocblas_status st = rocblas_sgemm(blas, rocblas_operation_none, rocblas_operation_transpose, rocblas_int(P),
                                          rocblas_int(P), rocblas_int(rows), &scale, G, rocblas_int(P), G, rocblas_int(P), &one,
                                          out, rocblas_int(P));
  if (st != rocblas_status_success) { set_error("rocblas_sgemm failed"); return 3; }
  return 0;
}

int jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, hipStream_t s) {
  LGNN_REQUIRE(h->L > 0, "no model bound");
  LGNN_CALL(forward_ensure(h, s));
  const int64_t N = h->N;
  const int L = h->L;
  const int64_t C = h->dims[L];
  const bool gcn = h->kind == LGNN_KIND_GCN;
  int64_t off_w[kMaxLayers], off_b[kMaxLayers], P = 0;
  for (int l = 0; l < L; ++l) {
    off_w[l] = P; P += h->dims[l + 1] * h->in_dim[l];
    off_b[l] = P; P += h->dims[l + 1];
  }
  int* bad = h->ws.flags.as<int>();
  if (f_out)
    LGNN_CALL(launch_gather_rows(h->fc.out.as<float>(), C, N, idx, M, C, f_out, bad + 2, s));

  int64_t maxw = C;
  for (int l = 1; l < L; ++l) maxw = std::max(maxw, h->in_dim[l]);  // d (GCN) or 2 d (GraphSAGE) of the hidden layers
  const int64_t per_sample = C * N * maxw * 4 * 2;                   // two plane buffers
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(M, h->ws_limit / std::max<int64_t>(per_sample, 1)));
  LGNN_REQUIRE(chunk * C < (int64_t(1) << 31) && N < (int64_t(1) << 31), "jacobians: chunk too large");
  LGNN_CALL(h->ws.planes_a.reserve(size_t(chunk) * C * N * maxw * 4));
  LGNN_CALL(h->ws.planes_b.reserve(size_t(chunk) * C * N * maxw * 4));
  h->ws.planes_a_zero_ptr = nullptr;  // (the GraphSAGE KFAC path keeps an invariant on this buffer)
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  const float one = 1.f, zero = 0.f;

  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t mc = std::min(chunk, M - m0);
    const int64_t planes = mc * C;
    float* g = h->ws.planes_a.as<float>();      // g_l, width dims[l+1]
    float* other = h->ws.planes_b.as<float>();  // GEMM output
    float* Jc = J + m0 * C * P;
    LGNN_HIP_CHECK(hipMemsetAsync(g, 0, size_t(planes) * N * C * 4, s));
    if (gcn)
      hipLaunchKernelGGL(jac_top_gcn_kernel, dim3(unsigned(cdiv(mc, 4))), dim3(256), 0, s, idx + m0, mc, N, C, h->P.rowptr,
                         h->P.col, h->P.val, g, bad);
    else
      hipLaunchKernelGGL(jac_top_sage_kernel, dim3(unsigned(cdiv(planes, 256))), dim3(256), 0, s, idx + m0, mc, N, C, g, bad);
    LGNN_HIP_CHECK(hipGetLastError());

    for (int l = L - 1; l >= 0; --l) {
      const int64_t dout = h->dims[l + 1], din = h->in_dim[l];
      // dW (row major [dout, din]) = g^T in_l.  Column major view: C[din, dout] = in_l^T[din, N] * g[N, dout]
      const rocblas_status st = rocblas_sgemm_strided_batched(
          blas, rocblas_operation_none, rocblas_operation_transpose, rocblas_int(din), rocblas_int(dout), rocblas_int(N),
          &one, h->fc.lin_in_p[l], rocblas_int(h->fc.lin_in_ld[l]), 0, g, rocblas_int(dout), rocblas_stride(N * dout), &zero,
          Jc + off_w[l], rocblas_int(din), rocblas_stride(P), rocblas_int(planes));
      if (st != rocblas_status_success) { set_error("rocblas_sgemm_strided_batched failed"); return 3; }
      hipLaunchKernelGGL(plane_colsum_kernel, dim3(unsigned(planes)), dim3(256), 0, s, g, N, dout, Jc + off_b[l], P);
      LGNN_HIP_CHECK(hipGetLastError());
      if (l == 0) break;
      const int64_t d = h->dims[l];
      if (gcn) {
        // up = act'(h_l) * (g W_l);  g_{l-1} = P^T up      (gnn/models/layers.py:45-46 backward)
        GemmEpilogue ep;
        ep.hact = h->fc.hact_p[l - 1]; ep.hact_ld = h->fc.hact_ld[l - 1]; ep.act = h->act; ep.hact_row_mod = N;
        LGNN_CALL(launch_gemm(g, dout, h->W[l], d, other, d, planes * N, dout, d, ep, s));
        SpmmArgs sa{};
        sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
        sa.in = other; sa.in_ld = d; sa.in_plane_stride = N * d;
        sa.out = g; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
        LGNN_CALL(launch_spmm_ex(sa, planes, s));
      } else {
        // dcat = g W_l [., 2d];  g_{l-1} = act'(h_l) * (dcat[:, :d] + P^T dcat[:, d:])   (layers.py:26-29 backward)
        GemmEpilogue ep;
        LGNN_CALL(launch_gemm(g, dout, h->W[l], 2 * d, other, 2 * d, planes * N, dout, 2 * d, ep, s));
        SpmmArgs sa{};
        sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
        sa.in = other + d; sa.in_ld = 2 * d; sa.in_plane_stride = N * 2 * d;
        sa.self = other; sa.self_ld = 2 * d; sa.self_plane_stride = N * 2 * d;
        sa.hact = h->fc.hact_p[l - 1]; sa.hact_ld = h->fc.hact_ld[l - 1]; sa.act = h->act;
        sa.out = g; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
        LGNN_CALL(launch_spmm_ex(sa, planes, s));
      }
    }
  }
  return 0;
}

}  // namespace lgnn

extern "C" int lgnn_jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, void* stream) {
  if (!h || (M > 0 && (!idx || !J))) { lgnn::set_error("null argument"); return 2; }
  if (M <= 0) return 0;
  return lgnn::jacobians(h, idx, M, J, f_out, static_cast<hipStream_t>(stream));
}
