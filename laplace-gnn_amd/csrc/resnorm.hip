// Row-local pieces of models built with ``res=True`` and / or ``norm="layer"|"batch"``.
//
// Reference: BaseGNN.forward (gnn/models/base_gnn.py:136-161) computes, for every hidden layer l < L-1,
//     x = res[l](x) + convs[l](adj, x)      (:141-144; res[l] = nn.Linear(in_l, hidden), :100-113)
//     x = norms[l](x)                       (:148; nn.LayerNorm(hidden) / nn.BatchNorm1d(hidden) / Identity, :86-95)
//     x = act(x); x = dropout(x)            (:151-155; Laplace.fit runs in eval mode: dropout = identity, BatchNorm uses
//                                            its running statistics)
// and the reference's autograd walks the same steps backwards for every class column of the KFAC backward passes
// (curvlinops/kfac.py:653-661) and every row of the Jacobians (laplace/curvature/curvature.py:89-130).  The norm
// parameters are not Laplace parameters (``'norms' not in k``, laplace/curvature/curvature.py:74-79), but the norm's
// Jacobian sits in every backward pass; res[l] is an nn.Linear and gets its own Kronecker block
// (curvlinops/kfac.py:877-916).
//
// Everything here is row local (one wavefront per row, lane-strided columns, wave reductions):
//   forward   xhat = (s - mean) * rstd,  n = gamma * xhat + beta,  h = act(n)
//   backward  LayerNorm:  ds = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)),  dxh = gamma * dn
//             BatchNorm (eval):  ds = gamma * rstd_channel * dn
// HBM bound: one read + one write of the plane rows.
#include "device_utils.h"
#include "lgnn_internal.h"

namespace lgnn {

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per node row
__global__ __launch_bounds__(256) void norm_forward_kernel(const float* __restrict__ spre, int64_t s_ld, int64_t N, int64_t W,
                                                           int norm, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ rmean,
                                                           const float* __restrict__ rvar, int act, float* __restrict__ out,
                                                           int64_t out_ld, float* __restrict__ xhat,
                                                           float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t n = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* __restrict__ x = spre + n * s_ld;
  if (norm == LGNN_NORM_LAYER) {
    float sum = 0.f;
    for (int64_t j = lane; j < W; j += 64) sum += x[j];
    const float mean = wave_sum_f(sum) / float(W);
    float sq = 0.f;
    for (int64_t j = lane; j < W; j += 64) { const float d = x[j] - mean; sq += d * d; }
    const float r = 1.0f / sqrtf(wave_sum_f(sq) / float(W) + eps);  // biased variance, eps inside the root (nn.LayerNorm)
    if (lane == 0) rstd[n] = r;
    for (int64_t j = lane; j < W; j += 64) {
      const float xh = (x[j] - mean) * r;
      xhat[n * W + j] = xh;
      out[n * out_ld + j] = act_apply(gamma[j] * xh + beta[j], act);
    }
  } else {  // eval-mode BatchNorm1d: the affine map of the running statistics
    for (int64_t j = lane; j < W; j += 64) {
      const float r = 1.0f / sqrtf(rvar[j] + eps);
      if (n == 0) rstd[j] = r;
      const float xh = (x[j] - rmean[j]) * r;
      xhat[n * W + j] = xh;
      out[n * out_ld + j] = act_apply(gamma[j] * xh + beta[j], act);
    }
  }
}

// one wave per plane row; node = row % N
__global__ __launch_bounds__(256) void resnorm_backward_kernel(float* __restrict__ U, int64_t ld, int64_t rows, int64_t N,
                                                               int64_t W, const float* __restrict__ add,
                                                               const float* __restrict__ hact, int64_t hact_ld, int act,
                                                               int norm, const float* __restrict__ gamma,
                                                               const float* __restrict__ xhat,
                                                               const float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = int64_t(gridDim.x) * 4;
  for (int64_t r = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); r < rows; r += stride) {
    const int64_t n = r % N;
    float* __restrict__ u = U + r * ld;
    const float* __restrict__ ad = add ? add + r * ld : nullptr;
    const float* __restrict__ hn = hact ? hact + n * hact_ld : nullptr;
    const float* __restrict__ xh = xhat ? xhat + n * W : nullptr;
    // pass 1: t = mask * (u + add), dxh = gamma * t; the two row means of the LayerNorm backward
    float s1 = 0.f, s2 = 0.f;
    for (int64_t j = lane; j < W; j += 64) {
      float t = u[j];
      if (ad) t += ad[j];
      if (hn) t *= act_deriv_from_out(hn[j], act);
      if (norm != LGNN_NORM_NONE) t *= gamma[j];
      u[j] = t;
      if (norm == LGNN_NORM_LAYER) { s1 += t; s2 += t * xh[j]; }
    }
    if (norm == LGNN_NORM_LAYER) {
      const float m1 = wave_sum_f(s1) / float(W), m2 = wave_sum_f(s2) / float(W);
      const float rs = rstd[n];
      // (each lane re-reads only the elements it wrote itself)
      for (int64_t j = lane; j < W; j += 64) u[j] = rs * (u[j] - m1 - xh[j] * m2);
    } else if (norm == LGNN_NORM_BATCH) {
      for (int64_t j = lane; j < W; j += 64) u[j] *= rstd[j];
    }
  }
}

// GraphSAGE with res: Wcomb[o][i] = W[o][i] + (i < d ? Wr[o][i] : 0), bcomb = b + br
__global__ void sage_res_weights_kernel(const float* __restrict__ W, const float* __restrict__ Wr, const float* __restrict__ b,
                                        const float* __restrict__ br, int64_t dout, int64_t d, float* __restrict__ Wc,
                                        float* __restrict__ bc) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q < dout * 2 * d) {
    const int64_t o = q / (2 * d), i = q - o * 2 * d;
    Wc[q] = W[q] + (i < d ? Wr[o * d + i] : 0.f);
  }
  if (q < dout) bc[q] = b[q] + br[q];
}

}  // namespace

int launch_norm_forward(lgnn_ctx* h, int layer, const float* spre, int64_t s_ld, float* out, int64_t out_ld, hipStream_t s) {
  LGNN_REQUIRE(h->norm != LGNN_NORM_NONE && layer >= 0 && layer < h->L - 1, "internal: norm forward without a norm");
  const int64_t N = h->N, W = h->dims[layer + 1];
  LGNN_CALL(h->fc.xhat[layer].reserve(size_t(N) * W * 4));
  LGNN_CALL(h->fc.rstd[layer].reserve(size_t(h->norm == LGNN_NORM_LAYER ? N : W) * 4));
  hipLaunchKernelGGL(norm_forward_kernel, dim3(unsigned(cdiv(N, 4))), dim3(256), 0, s, spre, s_ld, N, W, h->norm, h->norm_eps,
                     h->norm_w[layer], h->norm_b[layer], h->norm_mean[layer], h->norm_var[layer], h->act, out, out_ld,
                     h->fc.xhat[layer].as<float>(), h->fc.rstd[layer].as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_resnorm_backward(lgnn_ctx* h, int layer, float* U, int64_t ld, int64_t rows, const float* add, bool mask,
                            hipStream_t s) {
  LGNN_REQUIRE(layer >= 0 && layer < h->L - 1, "internal: hidden layer index");
  if (rows <= 0) return 0;
  const int64_t W = h->dims[layer + 1];
  const bool nrm = h->norm != LGNN_NORM_NONE;
  hipLaunchKernelGGL(resnorm_backward_kernel, dim3(unsigned(std::min<int64_t>(cdiv(rows, 4), 65536))), dim3(256), 0, s, U, ld,
                     rows, h->N, W, add, mask ? h->fc.hact_p[layer] : static_cast<const float*>(nullptr), h->fc.hact_ld[layer],
                     h->act, h->norm, nrm ? h->norm_w[layer] : static_cast<const float*>(nullptr),
                     nrm ? h->fc.xhat[layer].as<float>() : static_cast<const float*>(nullptr),
                     nrm ? h->fc.rstd[layer].as<float>() : static_cast<const float*>(nullptr));
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int build_sage_res_weights(lgnn_ctx* h, int l, hipStream_t s) {
  const int64_t d = h->dims[l], dout = h->dims[l + 1];
  LGNN_CALL(h->Wcomb[l].reserve(size_t(dout) * 2 * d * 4));
  LGNN_CALL(h->bcomb[l].reserve(size_t(dout) * 4));
  hipLaunchKernelGGL(sage_res_weights_kernel, dim3(unsigned(cdiv(dout * 2 * d, 256))), dim3(256), 0, s, h->W[l], h->Wr[l],
                     h->b[l], h->br[l], dout, d, h->Wcomb[l].as<float>(), h->bcomb[l].as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn

// Optional pieces of the model (see the header): called after lgnn_bind_model, which resets them.
extern "C" int lgnn_bind_extras(lgnn_ctx* h, const float* const* res_weights, const float* const* res_biases, int norm_kind,
                                const float* const* norm_weight, const float* const* norm_bias,
                                const float* const* norm_mean, const float* const* norm_var, float norm_eps) {
  using namespace lgnn;
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(h->L > 0, "bind the model first (lgnn_bind_model)");
  LGNN_REQUIRE(norm_kind == LGNN_NORM_NONE || norm_kind == LGNN_NORM_LAYER || norm_kind == LGNN_NORM_BATCH,
               "unknown norm kind");
  LGNN_REQUIRE((res_weights == nullptr) == (res_biases == nullptr), "res weights and biases come together");
  const int nh = h->L - 1;  // hidden layers
  h->has_res = res_weights != nullptr && nh > 0;
  h->norm = nh > 0 ? norm_kind : LGNN_NORM_NONE;
  h->norm_eps = norm_eps;
  h->n_params = 0;
  for (int l = 0; l < h->L; ++l) h->n_params += h->in_dim[l] * h->dims[l + 1] + h->dims[l + 1];
  for (int l = 0; l < nh; ++l) {
    if (h->has_res) {
      LGNN_REQUIRE(res_weights[l] && res_biases[l], "null res weight / bias pointer");
      h->Wr[l] = res_weights[l];
      h->br[l] = res_biases[l];
      h->n_params += h->dims[l] * h->dims[l + 1] + h->dims[l + 1];
    }
    if (h->norm != LGNN_NORM_NONE) {
      LGNN_REQUIRE(norm_weight && norm_bias && norm_weight[l] && norm_bias[l], "null norm weight / bias pointer");
      h->norm_w[l] = norm_weight[l];
      h->norm_b[l] = norm_bias[l];
      if (h->norm == LGNN_NORM_BATCH) {
        LGNN_REQUIRE(norm_mean && norm_var && norm_mean[l] && norm_var[l], "BatchNorm needs its running statistics");
        h->norm_mean[l] = norm_mean[l];
        h->norm_var[l] = norm_var[l];
      }
    }
  }
  return lgnn_invalidate(h);
}
