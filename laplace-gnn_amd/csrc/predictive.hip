// Matrix-free GLM predictive variance for 2-layer GCN and GraphSAGE models: diag(J P^-1 J^T) per evaluation node without ever
// materialising the Jacobians.
//
// Reference: la(x) -> _glm_predictive_distribution (laplace/baselaplace.py:1123-1158): Js [M, C, P] from
// torch.func.jacrev of the dense model (laplace/curvature/curvature.py:89-130), f_var = Js P^-1 Js^T through
// KronDecomposed.inv_square_form (laplace/utils/matrix.py:396-451) resp. the diagonal einsum (baselaplace.py:1901-1903); the
// default link (probit, baselaplace.py:610-616) reads the diagonal of f_var only.  M * C * P floats cap that route at a few
// hundred evaluation nodes of an arxiv-sized model (P = 43 k: 7 MB per node).
//
// For the model family of the path the per-node Jacobian is closed form, with a = idx[n]:
//   last layer :  d f_c / d W_1 = e_c (x) phi_a,   phi_a = (P H_1)[a],    d f_c / d b_1 = s_a e_c,  s_a = rowsum(P)[a]
//   first layer:  d f_c / d W_0 = diag(w_c) T_a,   T_a = sum_u P[a,u] d_u (x) xhat_u,  d_u = act'(h_1[u]), xhat = P X
//                 d f_c / d b_0 = w_c * tb_a,      tb_a = sum_u P[a,u] rowsum(P)[u] d_u                 (w_c = W_1[c, :])
// Kronecker posterior (block l: f B_l (x) A_l + delta, eigenpairs (Q_B, lB), (Q_A, lA); the bias block shares Q_B):
//   W_0 + b_0:  M_c = sum_u P[a,u] r_{u,c} (x) ztilde_u  with  r_{u,c} = Q_B0^T (w_c * d_u)   (precomputed [nodes][C][H])
//               ztilde_u = [Q_A0^T xhat_u | rowsum(P)[u]]  (precomputed [N][F+1]);   var_c += sum_ij M_c[i,j]^2 S0[i,j],
//               S0[i,j] = 1 / (f lB0_i lA0_j + delta_w0), bias column j = F: 1 / (f lB0_i + delta_b0)
//   W_1 + b_1:  var_c += sum_i Q_B1[c,i]^2 sum_j phitilde_j^2 S1[i,j] + s_a^2 kappa_c
// Diagonal posterior: var_c = sum_h w_c[h]^2 sum_j T_a[h,j]^2 / prec_0[h,j] + sum_j phi_a[j]^2 / prec_1[c,j] + s_a^2 / prec_b1[c].
// GraphSAGE (gnn/models/layers.py:18-29): the first Linear sees cat_0[v] = [x_v | (P x)_v] (bias column 1) and the node itself
// joins its neighbours as one more entry:  d f_c / d W_0 = diag(ws_c * d_a) (x) cat_0[a] + sum_u P[a,u] diag(wn_c * d_u) (x) cat_0[u]
// (ws / wn = self / neighbour half of W_1's row c), the last layer's feature row is phi_a = cat_1[a], s_a = 1.  Same kernel:
// the self entry is staged behind the row's entries with weight 1 and its own rotated rows R_self[m, c, :].
// One workgroup per evaluation node: thread (row i, column group) keeps its slice of the H x (F+1) tile in registers.
#include <rocblas/rocblas.h>

#include "lgnn_internal.h"

namespace lgnn {

void* blas_handle(hipStream_t s);  // eigh.hip

namespace {

constexpr int PJT = 64;   // feature columns of the tile per thread (the bias column is one more register)
constexpr int PUC = 48;   // neighbours staged per pass

__global__ void pred_mark_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ col, uint8_t* __restrict__ need, int* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) { if (lane == 0) bad[1] = 1; return; }
  for (int32_t p = rowptr[n] + lane; p < rowptr[n + 1]; p += 64) need[col[p]] = 1;
}
__global__ void pred_slots_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ count, int32_t* __restrict__ slot) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < *count) slot[list[k]] = int32_t(k);
}
// A[(k, c), h] = d[list[k0 + k], h] * W1[c, h] for a chunk of needed nodes
__global__ void pred_dw_kernel(const int32_t* __restrict__ list, int64_t k0, int64_t kn, const float* __restrict__ dact,
                               int64_t H, const float* __restrict__ W1, int64_t ldw, int64_t C, float* __restrict__ A) {
  const int64_t total = kn * C * H;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t h = t % H, kc = t / H, c = kc % C, k = kc / C;
    A[t] = dact[int64_t(list[k0 + k]) * H + h] * W1[c * ldw + h];
  }
}
// the same for the evaluation nodes themselves (GraphSAGE self path): A[(m, c), h] = d[idx[m0 + m], h] * Ws[c, h]
__global__ void pred_dw_self_kernel(const int64_t* __restrict__ idx, int64_t m0, int64_t mn, int64_t N,
                                    const float* __restrict__ dact, int64_t H, const float* __restrict__ W1, int64_t ldw,
                                    int64_t C, float* __restrict__ A) {
  const int64_t total = mn * C * H;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t h = t % H, kc = t / H, c = kc % C, m = kc / C;
    const int64_t a = idx[m0 + m];
    A[t] = (a >= 0 && a < N) ? dact[a * H + h] * W1[c * ldw + h] : 0.f;
  }
}

__device__ __forceinline__ float wave_sum_p(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// KRON = 1: R (rotated, per class) and the S tables; KRON = 0: diagonal posterior (class independent tile T_a).
//   Zt  [N, ldz]   : ztilde (KRON) / [xhat | rowsum] (diag), F1 = F + 1 columns used
//   R   [slots, C, H] (KRON) ; dact [N, H] (diag)
//   S0  [H, F1]    : weights of the squared tile entries (KRON: rotated basis; diag: 1 / prec of W_0 | b_0)
//   last layer: Pt [M, H] = phitilde (KRON) / phi (diag) of the batch rows; S1 [Ce, H] (KRON) / [C, H]; QB1sq [C, Ce] (KRON);
//   kappa [C].  C counts the OUTPUT rows -- the model's classes, or the rows of a linear map of the logits whose head E W_1
//   the caller passes as W1 (lgnn_glm_variance_mapped) -- and Ce = the model's classes (the eigen-directions of B_1)
//   SAGE = 1: one more staged entry per node (the node itself, weight 1; rows R_self[m] resp. the self half of W_1),
//   bias column 1 (rowsum == nullptr), last-layer width D1 = 2 H; W1 has row stride ldw, its neighbour half starts at wn_off
template <int KRON, int SAGE>
__global__ __launch_bounds__(512) void glm_var_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      const float* __restrict__ val, const int32_t* __restrict__ slot,
                                                      const float* __restrict__ Zt, int64_t ldz, int64_t F1,
                                                      const float* __restrict__ R, const float* __restrict__ Rs,
                                                      const float* __restrict__ dact, const float* __restrict__ W1,
                                                      int64_t ldw, int64_t wn_off, int64_t H, int64_t C, int64_t Ce, int Hp,
                                                      const float* __restrict__ S0, const float* __restrict__ Pt, int64_t D1,
                                                      const float* __restrict__ S1, const float* __restrict__ QB1sq,
                                                      const float* __restrict__ kappa, const float* __restrict__ rowsum,
                                                      float* __restrict__ var_out) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t m = blockIdx.x;
  const int64_t a = idx[m];
  const int NG = 512 / Hp;          // column groups
  const int i = tid % Hp, g = tid / Hp;
  const bool row_ok = i < H;
  const int CW = NG * PJT;          // columns per chunk
  const int64_t F = F1 - 1;
  float* __restrict__ sy = sm;                       // [PUC][CW] staged p_au * ztilde_u (chunk columns)
  float* __restrict__ syb = sm + size_t(PUC) * CW;   // [PUC] staged p_au * rowsum(P)[u]: the bias column
  float* __restrict__ svar = syb + PUC;              // [C] per-class variance
  float* __restrict__ sg = svar + C;                 // [H] (diag: g[h]) / [Ce] scratch (kron: t_i)
  const int nsg = max(int(H), max(int(C), int(Ce)));
  int32_t* __restrict__ su = reinterpret_cast<int32_t*>(sg + nsg);  // [PUC] slot / node of the staged neighbours
  if (a < 0 || a >= N) {  // flagged by the marking kernel
    for (int c = tid; c < C; c += 512) var_out[m * C + c] = 0.f;
    return;
  }
  for (int c = tid; c < nsg; c += 512) { if (c < C) svar[c] = 0.f; sg[c] = 0.f; }
  const int32_t ps = rowptr[a], pe = rowptr[a + 1];
  const int deg = pe - ps + SAGE;  // GraphSAGE: the node itself is one more entry (the last)
  const int nub = (deg + PUC - 1) / PUC;
  __syncthreads();

  auto stage = [&](int ub, int64_t jc0) {  // neighbours [ub * PUC, ...) x columns [jc0, jc0 + CW)
    const int un = min(PUC, deg - ub * PUC);
    for (int t = tid; t < un * CW; t += 512) {
      const int u = t / CW, j = t - u * CW;
      const int32_t p = ps + ub * PUC + u;
      const bool self = SAGE && p >= pe;
      const int64_t node = self ? a : int64_t(col[p]);
      const float pv = self ? 1.f : val[p];
      const int64_t jj = jc0 + j;
      sy[t] = jj < F ? pv * Zt[node * ldz + jj] : 0.f;
      if (j == 0) {
        // staged id: kron -> slot of the neighbour's rotated rows, -1 for the self entry; otherwise the node id, the self
        // entry as -1 - node
        su[u] = KRON ? (self ? -1 : slot[node]) : (self ? int32_t(-1 - node) : int32_t(node));
        syb[u] = pv * (rowsum ? rowsum[node] : 1.f);
      }
    }
    return un;
  };

  for (int64_t jc0 = 0; jc0 < F; jc0 += CW) {
    float Sreg[PJT];
#pragma unroll
    for (int j = 0; j < PJT; ++j) {
      const int64_t jj = jc0 + g * PJT + j;
      Sreg[j] = (row_ok && jj < F) ? S0[int64_t(i) * F1 + jj] : 0.f;
    }
    // the bias column rides along with the first chunk's column group 0
    const bool with_bias = jc0 == 0 && g == 0;
    const float Sb = (with_bias && row_ok) ? S0[int64_t(i) * F1 + F] : 0.f;
    int un = 0;
    if (nub == 1) { un = stage(0, jc0); __syncthreads(); }
    const int ncls = (KRON || SAGE) ? int(C) : 1;
    for (int c = 0; c < ncls; ++c) {
      float Mt[PJT], Mb = 0.f;
#pragma unroll
      for (int j = 0; j < PJT; ++j) Mt[j] = 0.f;
      for (int ub = 0; ub < nub; ++ub) {
        if (nub > 1) { __syncthreads(); un = stage(ub, jc0); __syncthreads(); }
        for (int u0 = 0; u0 < un; u0 += 4) {
          float r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            r[q] = 0.f;
            if (u0 + q < un && row_ok) {
              const int32_t id = su[u0 + q];
              if (KRON) r[q] = id >= 0 ? R[(int64_t(id) * C + c) * H + i] : Rs[(m * C + c) * H + i];
              else if (SAGE) r[q] = id >= 0 ? dact[int64_t(id) * H + i] * W1[c * ldw + wn_off + i]
                                            : dact[int64_t(-1 - id) * H + i] * W1[c * ldw + i];
              else r[q] = dact[int64_t(id) * H + i];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (u0 + q < un) {
              const float4* __restrict__ yr = reinterpret_cast<const float4*>(sy + (u0 + q) * CW + g * PJT);
#pragma unroll
              for (int j4 = 0; j4 < PJT / 4; ++j4) {
                const float4 y4 = yr[j4];
                Mt[4 * j4] = fmaf(r[q], y4.x, Mt[4 * j4]); Mt[4 * j4 + 1] = fmaf(r[q], y4.y, Mt[4 * j4 + 1]);
                Mt[4 * j4 + 2] = fmaf(r[q], y4.z, Mt[4 * j4 + 2]); Mt[4 * j4 + 3] = fmaf(r[q], y4.w, Mt[4 * j4 + 3]);
              }
              Mb = fmaf(r[q], syb[u0 + q], Mb);
            }
          }
        }
      }
      float part = Mb * Mb * Sb;
#pragma unroll
      for (int j = 0; j < PJT; ++j) part = fmaf(Mt[j] * Mt[j], Sreg[j], part);
      if (KRON || SAGE) {
        part = wave_sum_p(part);
        if (lane == 0) atomicAdd(&svar[c], part);
      } else if (row_ok) {
        atomicAdd(&sg[i], part);  // g[h] = sum_j T[h, j]^2 / prec[h, j], summed over column groups and chunks
      }
    }
    __syncthreads();
  }
  // ---- the class mixing of the diagonal posterior and the last layer
  const float sa = rowsum ? rowsum[a] : 1.f;
  const float* __restrict__ ph = Pt + m * D1;
  if (!KRON) {
    for (int c = tid; c < C; c += 512) {
      float v = SAGE ? svar[c] : 0.f;
      if (!SAGE)
        for (int64_t h = 0; h < H; ++h) { const float w = W1[c * ldw + h]; v = fmaf(w * w, sg[h], v); }
      for (int64_t j = 0; j < D1; ++j) v = fmaf(ph[j] * ph[j], S1[c * D1 + j], v);
      var_out[m * C + c] = v + sa * sa * kappa[c];
    }
  } else {
    // t_i = sum_j phitilde_j^2 S1[i, j]  (i < Ce), then var_c += sum_i QB1sq[c, i] t_i + s_a^2 kappa_c
    __syncthreads();
    for (int c = tid; c < Ce; c += 512) {
      float t = 0.f;
      for (int64_t j = 0; j < D1; ++j) t = fmaf(ph[j] * ph[j], S1[c * D1 + j], t);
      sg[c] = t;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 512) {
      float v = svar[c];
      for (int64_t k = 0; k < Ce; ++k) v = fmaf(QB1sq[c * Ce + k], sg[k], v);
      var_out[m * C + c] = v + sa * sa * kappa[c];
    }
  }
}

int sgemm_rm_p(hipStream_t s, int64_t R, int64_t Nout, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
               float* Cm, int64_t ldc) {
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  LGNN_REQUIRE(R < (int64_t(1) << 31), "sgemm: too many rows");
  const float one = 1.f, zero = 0.f;
  const rocblas_status st = rocblas_sgemm(blas, rocblas_operation_none, rocblas_operation_none, rocblas_int(Nout),
                                          rocblas_int(R), rocblas_int(K), &one, B, rocblas_int(ldb), A, rocblas_int(lda),
                                          &zero, Cm, rocblas_int(ldc));
  if (st != rocblas_status_success) { set_error("rocblas_sgemm failed"); return 3; }
  return 0;
}

}  // namespace

// kron: QA0 [F, F], QB0 [H, H], QA1 [H, H] rotate; diag posterior: the three are null.  S0 [H, F + 1], S1 [C, H],
// QB1sq [C, C] (kron only), kappa [C].
// W1m [Cm, in_dim_1] (row major, or null): the head E W_1 of a linear map E [Cm, C] of the logits; f_var is then [M, Cm] =
// diag(E J P^-1 J^T E^T) and S1 / QB1sq / kappa are the caller's mapped operands (see the header).
int glm_variance(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* W1m, int64_t Cm, const float* QA0, const float* QB0,
                 const float* S0, const float* QA1, const float* S1, const float* QB1sq, const float* kappa, float* f_mu,
                 float* f_var, hipStream_t s) {
  LGNN_REQUIRE(h->L == 2, "matrix-free GLM predictive: 2-layer models");
  LGNN_REQUIRE(!h->extras(), "matrix-free GLM predictive: models without res / norm (use the Jacobian route)");
  LGNN_REQUIRE(M > 0 && idx && S0 && S1 && kappa && f_var, "empty batch or null pointers");
  const bool kron = QA0 != nullptr;
  const bool sage = h->kind == LGNN_KIND_SAGE;
  LGNN_REQUIRE(!kron || (QB0 && QA1 && QB1sq), "kron posterior needs all eigenvector matrices");
  LGNN_CALL(forward_ensure_aux(h, s));
  // F: width of what the first Linear multiplies (GraphSAGE: the concatenation), D1: the same for the last Linear
  const int64_t N = h->N, F = h->in_dim[0], H = h->dims[1], Ce = h->dims[2], F1 = F + 1, D1 = h->in_dim[1];
  LGNN_REQUIRE(H <= 256, "matrix-free GLM predictive: hidden width <= 256");
  LGNN_REQUIRE(W1m == nullptr || (Cm > 0 && Cm <= 4096), "mapped GLM predictive: 1 <= rows of the map <= 4096");
  const int64_t C = W1m ? Cm : Ce;           // output rows
  const float* W1 = W1m ? W1m : h->W[1];     // their head [C, in_dim_1]
  const int Hp = H <= 64 ? 64 : (H <= 128 ? 128 : 256);
  int* bad = h->ws.flags.as<int>();
  if (f_mu) LGNN_CALL(launch_gather_rows(h->fc.out.as<float>(), Ce, N, idx, M, Ce, f_mu, bad + 2, s));

  // ztilde [N, ldz]: E Q_A0 (kron) or E itself (diag), E = P X (GCN) / cat_0 (GraphSAGE); the bias column (rowsum(P) / 1)
  // is read where it is used
  const int64_t ldz = cdiv(F, 4) * 4;
  const float* xhat = sage ? h->fc.lin_in_p[0] : h->fc.prop_in[0].as<float>();
  const int64_t ldx = sage ? h->fc.lin_in_ld[0] : h->fc.prop_ld[0];
  LGNN_CALL(h->ws.planes_a.reserve(size_t(N) * ldz * 4 + size_t(M) * D1 * 4 * 2));
  h->ws.planes_a_zero_ptr = nullptr;
  float* Zt = h->ws.planes_a.as<float>();
  float* PhiB = Zt + N * ldz;    // [M, D1] phi of the batch rows
  float* PhiT = PhiB + M * D1;   // [M, D1] rotated
  if (kron) LGNN_CALL(sgemm_rm_p(s, N, F, F, xhat, ldx, QA0, F, Zt, ldz));
  else LGNN_HIP_CHECK(hipMemcpy2DAsync(Zt, size_t(ldz) * 4, xhat, size_t(ldx) * 4, size_t(F) * 4, size_t(N),
                                       hipMemcpyDeviceToDevice, s));
  // phi_a = (P H_1)[a] resp. cat_1[a] for the batch rows (+ rotation)
  if (sage) LGNN_CALL(launch_gather_rows(h->fc.lin_in_p[1], h->fc.lin_in_ld[1], N, idx, M, D1, PhiB, bad + 2, s));
  else LGNN_CALL(launch_gather_rows(h->fc.prop_in[1].as<float>(), h->fc.prop_ld[1], N, idx, M, D1, PhiB, bad + 2, s));
  const float* Pt = PhiB;
  if (kron) { LGNN_CALL(sgemm_rm_p(s, M, D1, D1, PhiB, D1, QA1, D1, PhiT, D1)); Pt = PhiT; }

  // nodes whose rows are needed: the columns of the batch rows of P
  const int64_t ldw = h->in_dim[1], wn_off = sage ? H : 0;  // W_1 [C, ldw]; its neighbour half
  const int32_t* slot = nullptr;
  const float* R = nullptr;
  const float* Rs = nullptr;
  if (kron) {
    LGNN_CALL(h->ws.active.reserve(size_t(N)));
    LGNN_HIP_CHECK(hipMemsetAsync(h->ws.active.p, 0, size_t(N), s));
    hipLaunchKernelGGL(pred_mark_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr, h->P.col,
                       h->ws.active.as<uint8_t>(), bad);
    LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
    LGNN_CALL(h->ws.act_count.reserve(64));
    LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                            h->ws.select_tmp, s));
    int32_t nneed = 0;
    LGNN_HIP_CHECK(hipMemcpyAsync(&nneed, h->ws.act_count.p, 4, hipMemcpyDeviceToHost, s));
    LGNN_HIP_CHECK(hipStreamSynchronize(s));  // the size of R has to reach the host
    LGNN_CALL(h->ws.misc.reserve(size_t(N) * 4));
    int32_t* slotw = h->ws.misc.as<int32_t>();
    hipLaunchKernelGGL(pred_slots_kernel, dim3(unsigned(cdiv(std::max<int64_t>(nneed, 1), 256))), dim3(256), 0, s,
                       h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(), slotw);
    slot = slotw;
    // R[slot, c, :] = (d_u * wn_c) Q_B0, in chunks of nodes: the elementwise operand goes through planes_b;
    // GraphSAGE also R_self[m, c, :] = (d_a * ws_c) Q_B0 for the evaluation nodes themselves (behind R)
    const int64_t rows_r = std::max<int64_t>(nneed, 1), rows_s = sage ? M : 0;
    LGNN_CALL(h->ws.top.reserve(size_t(rows_r + rows_s) * C * H * 4));
    float* Rw = h->ws.top.as<float>();
    float* Rsw = Rw + rows_r * C * H;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(nneed, rows_s), (int64_t(1) << 28) / (C * H)));  // <= 1 GiB operand
    LGNN_CALL(h->ws.planes_b.reserve(size_t(chunk) * C * H * 4));
    for (int64_t k0 = 0; k0 < nneed; k0 += chunk) {
      const int64_t kn = std::min<int64_t>(chunk, nneed - k0);
      hipLaunchKernelGGL(pred_dw_kernel, dim3(unsigned(std::min<int64_t>(cdiv(kn * C * H, 256), 8192))), dim3(256), 0, s,
                         h->ws.act_list.as<int32_t>(), k0, kn, h->fc.dact0.as<float>(), H, W1 + wn_off, ldw, C,
                         h->ws.planes_b.as<float>());
      LGNN_CALL(sgemm_rm_p(s, kn * C, H, H, h->ws.planes_b.as<float>(), H, QB0, H, Rw + k0 * C * H, H));
    }
    for (int64_t m0 = 0; m0 < rows_s; m0 += chunk) {
      const int64_t mn = std::min<int64_t>(chunk, rows_s - m0);
      hipLaunchKernelGGL(pred_dw_self_kernel, dim3(unsigned(std::min<int64_t>(cdiv(mn * C * H, 256), 8192))), dim3(256), 0, s,
                         idx, m0, mn, N, h->fc.dact0.as<float>(), H, W1, ldw, C, h->ws.planes_b.as<float>());
      LGNN_CALL(sgemm_rm_p(s, mn * C, H, H, h->ws.planes_b.as<float>(), H, QB0, H, Rsw + m0 * C * H, H));
    }
    R = Rw;
    Rs = Rsw;
  }
  const int NG = 512 / Hp, CW = NG * PJT;
  const size_t smem = (size_t(PUC) * CW + PUC + size_t(C) + size_t(std::max<int64_t>(H, std::max(C, Ce))) + PUC) * 4;
  LGNN_REQUIRE(smem <= 150 * 1024, "matrix-free GLM predictive: tile does not fit LDS");
  const float* rowsum = sage ? nullptr : h->fc.rowsum.as<float>();
#define LGNN_GLM_VAR(K, S)                                                                                                   \
  do {                                                                                                                       \
    static bool attr = false;                                                                                                \
    if (!attr) {                                                                                                             \
      LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&glm_var_kernel<K, S>),                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));                          \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((glm_var_kernel<K, S>), dim3(unsigned(M)), dim3(512), smem, s, idx, M, N, h->P.rowptr, h->P.col,      \
                       h->P.val, slot, Zt, ldz, F1, R, Rs, h->fc.dact0.as<float>(), W1, ldw, wn_off, H, C, Ce, Hp, S0, Pt,   \
                       D1, S1, QB1sq, kappa, rowsum, f_var);                                                                 \
  } while (0)
  if (kron && sage) LGNN_GLM_VAR(1, 1);
  else if (kron) LGNN_GLM_VAR(1, 0);
  else if (sage) LGNN_GLM_VAR(0, 1);
  else LGNN_GLM_VAR(0, 0);
#undef LGNN_GLM_VAR
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn

extern "C" int lgnn_glm_variance(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* QA0, const float* QB0, const float* S0,
                                 const float* QA1, const float* S1, const float* QB1sq, const float* kappa, float* f_mu,
                                 float* f_var_diag, void* stream) {
  if (!h) { lgnn::set_error("null context"); return 2; }
  return lgnn::glm_variance(h, idx, M, nullptr, 0, QA0, QB0, S0, QA1, S1, QB1sq, kappa, f_mu, f_var_diag,
                            static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_glm_variance_mapped(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* W1m, int64_t Cm, const float* QA0,
                                        const float* QB0, const float* S0, const float* QA1, const float* S1, const float* QB1sq,
                                        const float* kappa, float* f_mu, float* var_mapped, void* stream) {
  if (!h) { lgnn::set_error("null context"); return 2; }
  if (!W1m) { lgnn::set_error("lgnn_glm_variance_mapped: null head of the map"); return 2; }
  return lgnn::glm_variance(h, idx, M, W1m, Cm, QA0, QB0, S0, QA1, S1, QB1sq, kappa, f_mu, var_mapped,
                            static_cast<hipStream_t>(stream));
}
