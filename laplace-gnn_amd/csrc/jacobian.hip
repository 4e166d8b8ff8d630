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

// Closed-form Jacobians of 1- and 2-layer models: no planes over all N nodes, no GEMM with K = N.  With a = idx[m],
// E[v] = [what the first Linear multiplies at node v | 1-column] and d_v = act'(h_1[v]):
//   last layer   d f_c / d W_last = e_c (x) phi_a,   d f_c / d b_last = s_a e_c            (phi, s: feat_views of diag.hip)
//   first layer  GCN      : d f_c / d W_0[h, :] = w_c[h] sum_u P[a,u] d_u[h] (P X)[u, :],  bias: ... rowsum(P)[u]
//                GraphSAGE: d f_c / d W_0[h, :] = ws_c[h] d_a[h] cat_0[a, :] + wn_c[h] sum_u P[a,u] d_u[h] cat_0[u, :],  bias: ... 1
// One workgroup per sample: T_self / T_neigh [H x (in_0 + 1)] tiles are built once from the ~15 neighbours and every
// class row of J is a row scaling of them -- the kernel is bound by writing J (M * C * P floats, like the reference's).
__global__ __launch_bounds__(256) void jac_closed_form_kernel(
    const int64_t* __restrict__ idx, int64_t M, int64_t N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int L, int sage, int64_t in0, int64_t H, int64_t C,
    const float* __restrict__ E0, int64_t e0_ld, const float* __restrict__ e0_bias /* null: 1 */,
    const float* __restrict__ dact, const float* __restrict__ W1, int64_t w1_ld,
    const float* __restrict__ PhiL, int64_t phil_ld, int64_t phil_w, const float* __restrict__ phil_bias /* null: 1 */,
    int64_t P, float* __restrict__ J, int* __restrict__ bad) {
  const int64_t m = blockIdx.x;
  const int64_t a = idx[m];
  float* __restrict__ Jm = J + m * C * P;
  const int64_t t0 = int64_t(blockIdx.y) * blockDim.x + threadIdx.x, tstep = int64_t(gridDim.y) * blockDim.x;
  if (a < 0 || a >= N) {
    if (t0 == 0) bad[1] = 1;
    for (int64_t t = t0; t < C * P; t += tstep) Jm[t] = 0.f;
    return;
  }
  const int64_t in_last = phil_w;  // columns of the last weight
  const int64_t off_last = L == 2 ? H * in0 + H : 0;
  // last layer: row c of J holds phi_a in its own block, zeros in the other classes' blocks, s_a at its bias entry
  for (int64_t t = t0; t < C * (C * in_last + C); t += tstep) {
    const int64_t c = t / (C * in_last + C), q = t - c * (C * in_last + C);
    float v = 0.f;
    if (q < C * in_last) { if (q / in_last == c) v = PhiL[a * phil_ld + (q - c * in_last)]; }
    else if (q - C * in_last == c) v = phil_bias ? phil_bias[a] : 1.f;
    Jm[c * P + off_last + q] = v;
  }
  if (L == 1) return;
  // first layer: element (h, i) with i <= in0 (i == in0: the bias entry of unit h)
  const int32_t ps = rowptr[a], pe = rowptr[a + 1];
  const int64_t in1 = in0 + 1;
  for (int64_t t = t0; t < H * in1; t += tstep) {
    const int64_t hh = t / in1, i = t - hh * in1;
    float tn = 0.f;
    for (int32_t p = ps; p < pe; ++p) {
      const int64_t u = col[p];
      const float e = i < in0 ? E0[u * e0_ld + i] : (e0_bias ? e0_bias[u] : 1.f);
      tn = fmaf(val[p] * dact[u * H + hh], e, tn);
    }
    float ts = 0.f;
    if (sage) ts = dact[a * H + hh] * (i < in0 ? E0[a * e0_ld + i] : 1.f);
    const int64_t dst = i < in0 ? hh * in0 + i : H * in0 + hh;
    for (int64_t c = 0; c < C; ++c) {
      const float v = sage ? W1[c * w1_ld + hh] * ts + W1[c * w1_ld + H + hh] * tn : W1[c * w1_ld + hh] * tn;
      Jm[c * P + dst] = v;
    }
  }
}

static int jacobians_closed_form(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, hipStream_t s) {
  LGNN_CALL(forward_ensure_aux(h, s));
  const int L = h->L;
  const int64_t N = h->N, C = h->dims[L], H = L == 2 ? h->dims[1] : 0, in0 = h->in_dim[0], P = h->n_params;
  const bool sage = h->kind == LGNN_KIND_SAGE;
  int* bad = h->ws.flags.as<int>();
  if (f_out) LGNN_CALL(launch_gather_rows(h->fc.out.as<float>(), C, N, idx, M, C, f_out, bad + 2, s));
  // first-layer feature rows: GCN (P X)[u] with the bias entry rowsum(P)[u]; GraphSAGE cat_0[u] with 1
  const float* E0 = sage ? h->fc.lin_in_p[0] : h->fc.prop_in[0].as<float>();
  const int64_t e0_ld = sage ? h->fc.lin_in_ld[0] : h->fc.prop_ld[0];
  const float* e0_bias = sage ? nullptr : h->fc.rowsum.as<float>();
  // last-layer feature row seen from the batch node
  const float* PhiL = sage ? h->fc.lin_in_p[L - 1] : h->fc.prop_in[L - 1].as<float>();
  const int64_t phil_ld = sage ? h->fc.lin_in_ld[L - 1] : h->fc.prop_ld[L - 1];
  const float* phil_bias = sage ? nullptr : h->fc.rowsum.as<float>();
  LGNN_REQUIRE(M < (int64_t(1) << 31), "too many samples");
  // enough workgroups per sample to fill the chip at small M
  const unsigned per = unsigned(std::max<int64_t>(1, std::min<int64_t>(cdiv(std::max<int64_t>(H * (in0 + 1), C * (C * h->in_dim[L - 1] + C)), 256), cdiv(2048, M))));
  hipLaunchKernelGGL(jac_closed_form_kernel, dim3(unsigned(M), per), dim3(256), 0, s, idx, M, N, h->P.rowptr, h->P.col, h->P.val, L,
                     sage ? 1 : 0, in0, H, C, E0, e0_ld, e0_bias, L == 2 ? h->fc.dact0.as<float>() : nullptr,
                     L == 2 ? h->W[1] : nullptr, L == 2 ? h->in_dim[1] : 0, PhiL, phil_ld, h->in_dim[L - 1], phil_bias, P, J, bad);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// Per-sample loss gradients G[m, :] = sum_c r[m, c] J[m, c, :] of 1- and 2-layer models straight from the closed form (the
// empirical / MC Fisher's rows, laplace/curvature/curvature.py:169-210): with rho_s[h] = sum_c r_c ws_c[h], rho_n[h] =
// sum_c r_c wn_c[h] the first layer's gradient is rho_s[h] T_self[h, :] + rho_n[h] T_neigh[h, :], the last layer's is
// r (x) phi_a -- M * P floats instead of the M * C * P of the Jacobians.
__global__ __launch_bounds__(256) void ef_grads_closed_form_kernel(
    const int64_t* __restrict__ idx, int64_t M, int64_t N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int L, int sage, int64_t in0, int64_t H, int64_t C,
    const float* __restrict__ E0, int64_t e0_ld, const float* __restrict__ e0_bias /* null: 1 */,
    const float* __restrict__ dact, const float* __restrict__ W1, int64_t w1_ld,
    const float* __restrict__ PhiL, int64_t phil_ld, int64_t phil_w, const float* __restrict__ phil_bias /* null: 1 */,
    const float* __restrict__ r, int64_t P, float* __restrict__ G) {
  const int64_t m = blockIdx.x;
  const int64_t a = idx[m];
  float* __restrict__ Gm = G + m * P;
  const float* __restrict__ rm = r + m * C;
  const int64_t t0 = int64_t(blockIdx.y) * blockDim.x + threadIdx.x, tstep = int64_t(gridDim.y) * blockDim.x;
  if (a < 0 || a >= N) {  // flagged by the batch prologue
    for (int64_t t = t0; t < P; t += tstep) Gm[t] = 0.f;
    return;
  }
  const int64_t in_last = phil_w;
  const int64_t off_last = L == 2 ? H * in0 + H : 0;
  for (int64_t t = t0; t < C * in_last + C; t += tstep) {
    float v;
    if (t < C * in_last) { const int64_t k = t / in_last; v = rm[k] * PhiL[a * phil_ld + (t - k * in_last)]; }
    else v = rm[t - C * in_last] * (phil_bias ? phil_bias[a] : 1.f);
    Gm[off_last + t] = v;
  }
  if (L == 1) return;
  const int32_t ps = rowptr[a], pe = rowptr[a + 1];
  const int64_t in1 = in0 + 1;
  for (int64_t t = t0; t < H * in1; t += tstep) {
    const int64_t hh = t / in1, i = t - hh * in1;
    float rs = 0.f, rn = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float rc = rm[c];
      if (sage) { rs = fmaf(rc, W1[c * w1_ld + hh], rs); rn = fmaf(rc, W1[c * w1_ld + H + hh], rn); }
      else rn = fmaf(rc, W1[c * w1_ld + hh], rn);
    }
    float tn = 0.f;
    for (int32_t p = ps; p < pe; ++p) {
      const int64_t u = col[p];
      const float e = i < in0 ? E0[u * e0_ld + i] : (e0_bias ? e0_bias[u] : 1.f);
      tn = fmaf(val[p] * dact[u * H + hh], e, tn);
    }
    float v = rn * tn;
    if (sage) v = fmaf(rs, dact[a * H + hh] * (i < in0 ? E0[a * e0_ld + i] : 1.f), v);
    Gm[i < in0 ? hh * in0 + i : H * in0 + hh] = v;
  }
}

int ef_grads_closed_form(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* r, float* G, hipStream_t s) {
  LGNN_REQUIRE(h->L >= 1 && h->L <= 2, "closed-form gradients: 1- and 2-layer models");
  LGNN_CALL(forward_ensure_aux(h, s));
  const int L = h->L;
  const int64_t N = h->N, C = h->dims[L], H = L == 2 ? h->dims[1] : 0, in0 = h->in_dim[0], P = h->n_params;
  const bool sage = h->kind == LGNN_KIND_SAGE;
  const float* E0 = sage ? h->fc.lin_in_p[0] : h->fc.prop_in[0].as<float>();
  const int64_t e0_ld = sage ? h->fc.lin_in_ld[0] : h->fc.prop_ld[0];
  const float* bias_col = sage ? nullptr : h->fc.rowsum.as<float>();
  const float* PhiL = sage ? h->fc.lin_in_p[L - 1] : h->fc.prop_in[L - 1].as<float>();
  const int64_t phil_ld = sage ? h->fc.lin_in_ld[L - 1] : h->fc.prop_ld[L - 1];
  LGNN_REQUIRE(M < (int64_t(1) << 31), "too many samples");
  const unsigned per = unsigned(std::max<int64_t>(1, std::min<int64_t>(cdiv(std::max<int64_t>(H * (in0 + 1), C * h->in_dim[L - 1] + C), 256), cdiv(4096, M))));
  hipLaunchKernelGGL(ef_grads_closed_form_kernel, dim3(unsigned(M), per), dim3(256), 0, s, idx, M, N, h->P.rowptr, h->P.col,
                     h->P.val, L, sage ? 1 : 0, in0, H, C, E0, e0_ld, bias_col, L == 2 ? h->fc.dact0.as<float>() : nullptr,
                     L == 2 ? h->W[1] : nullptr, L == 2 ? h->in_dim[1] : 0, PhiL, phil_ld, h->in_dim[L - 1], bias_col, r, P, G);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* J, float* f_out, hipStream_t s) {
  LGNN_REQUIRE(h->L > 0, "no model bound");
  // 1- and 2-layer models: closed form (LGNN_JAC_PLANES=1 forces the generic plane route, which deeper models use)
  const bool force_planes = getenv("LGNN_JAC_PLANES") != nullptr;
  // (models with res / norm have no closed form: the norm's row-local backward sits between the layers)
  if (h->L <= 2 && !force_planes && !h->extras()) return jacobians_closed_form(h, idx, M, J, f_out, s);
  LGNN_CALL(forward_ensure(h, s));
  const int64_t N = h->N;
  const int L = h->L;
  const int64_t C = h->dims[L];
  const bool gcn = h->kind == LGNN_KIND_GCN;
  int64_t off_w[kMaxLayers], off_b[kMaxLayers], off_rw[kMaxLayers], off_rb[kMaxLayers], P = 0;
  for (int l = 0; l < L; ++l) {
    off_w[l] = P; P += h->dims[l + 1] * h->in_dim[l];
    off_b[l] = P; P += h->dims[l + 1];
  }
  for (int l = 0; h->has_res && l < L - 1; ++l) {  // res.{l}.weight|bias follow all convs.* (named_parameters order)
    off_rw[l] = P; P += h->dims[l + 1] * h->dims[l];
    off_rb[l] = P; P += h->dims[l + 1];
  }
  LGNN_REQUIRE(P == h->n_params, "internal: parameter count");
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
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  const float one = 1.f, zero = 0.f;
  const bool gcn_res = gcn && h->has_res;
  if (gcn_res && L > 2) LGNN_CALL(h->ws.planes_c.reserve(size_t(chunk) * C * N * maxw * 4));
  // d f / d res.{l} = u_l^T h_l, column sums of u_l (u_l = gradient at s_l; GraphSAGE: u_l = g_l)
  auto res_block = [&](int l, const float* u, float* Jc, int64_t planes) -> int {
    const int64_t dout = h->dims[l + 1], din = h->dims[l];
    const rocblas_status st = rocblas_sgemm_strided_batched(
        blas, rocblas_operation_none, rocblas_operation_transpose, rocblas_int(din), rocblas_int(dout), rocblas_int(N), &one,
        h->fc.lin_in_p[l], rocblas_int(h->fc.lin_in_ld[l]), 0, u, rocblas_int(dout), rocblas_stride(N * dout), &zero,
        Jc + off_rw[l], rocblas_int(din), rocblas_stride(P), rocblas_int(planes));
    if (st != rocblas_status_success) { set_error("rocblas_sgemm_strided_batched failed"); return 3; }
    hipLaunchKernelGGL(plane_colsum_kernel, dim3(unsigned(planes)), dim3(256), 0, s, u, N, dout, Jc + off_rb[l], P);
    LGNN_HIP_CHECK(hipGetLastError());
    return 0;
  };
  h->ws.planes_a_zero_ptr = nullptr;  // (the GraphSAGE KFAC path keeps an invariant on this buffer)

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
      if (!gcn && h->has_res && l < L - 1) LGNN_CALL(res_block(l, g, Jc, planes));  // GraphSAGE: res.{l} sees g_l too
      if (l == 0) break;
      const int64_t d = h->dims[l];
      if (gcn) {
        // up = act'(h_l) * (g W_l);  g_{l-1} = P^T up      (gnn/models/layers.py:45-46 backward)
        GemmEpilogue ep;
        ep.hact = h->fc.hact_p[l - 1]; ep.hact_ld = h->fc.hact_ld[l - 1]; ep.act = h->act; ep.hact_row_mod = N;
        // res / norm (base_gnn.py:141-149 backwards): below the top level dh_l = g_l W_l + u_l Wr_l with u_l still in
        // `other`; mask and the norm's row-local backward give u_{l-1}, which res.{l-1} sees and P^T propagates
        const bool res_term = gcn_res && l < L - 1;
        if (res_term) {
          GemmEpilogue none;
          LGNN_CALL(launch_gemm(other, dout, h->Wr[l], d, h->ws.planes_c.as<float>(), d, planes * N, dout, d, none, s));
          ep = none;
        }
        LGNN_CALL(launch_gemm(g, dout, h->W[l], d, other, d, planes * N, dout, d, ep, s));
        if (res_term || h->norm != LGNN_NORM_NONE)
          LGNN_CALL(launch_resnorm_backward(h, l - 1, other, d, planes * N, res_term ? h->ws.planes_c.as<float>() : nullptr,
                                            res_term, s));
        if (gcn_res) LGNN_CALL(res_block(l - 1, other, Jc, planes));
        SpmmArgs sa{};
        sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
        sa.in = other; sa.in_ld = d; sa.in_plane_stride = N * d;
        sa.out = g; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
        LGNN_CALL(launch_spmm_ex(sa, planes, s));
      } else {
        // dcat = g W_l [., 2d];  g_{l-1} = act'(h_l) * (dcat[:, :d] + P^T dcat[:, d:])   (layers.py:26-29 backward)
        // (with res: W_l + [Wr_l | 0], the gradient at the Linear's output is res.{l}'s too)
        GemmEpilogue ep;
        LGNN_CALL(launch_gemm(g, dout, h->Wback(l), 2 * d, other, 2 * d, planes * N, dout, 2 * d, ep, s));
        SpmmArgs sa{};
        sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
        sa.in = other + d; sa.in_ld = 2 * d; sa.in_plane_stride = N * 2 * d;
        sa.self = other; sa.self_ld = 2 * d; sa.self_plane_stride = N * 2 * d;
        sa.hact = h->fc.hact_p[l - 1]; sa.hact_ld = h->fc.hact_ld[l - 1]; sa.act = h->act;
        sa.out = g; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
        LGNN_CALL(launch_spmm_ex(sa, planes, s));
        if (h->norm != LGNN_NORM_NONE) LGNN_CALL(launch_resnorm_backward(h, l - 1, g, d, planes * N, nullptr, false, s));
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
