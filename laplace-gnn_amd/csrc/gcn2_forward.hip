// Forward pass of a plain 2-layer GCN through the cached propagated input, for the closed-form diagonal GGN of small graphs.
//
// Reference: GCN.forward = adj @ Linear(x) per layer, ReLU between (gnn/models/layers.py:45-46, gnn/models/base_gnn.py:141-156).
// The closed-form diagonal (diag.hip) needs P X, rowsum(P), P h_1 and act'(h_1) next to the logits, and of those P X and
// rowsum(P) depend on the graph and the features only, so they are kept across weight updates (ForwardCache::px_valid).
// With them the pass is four launches and no weight transpose:
//     h_1  = act((P X) W_0^T + rowsum(P) b_0^T)         gemm_nt_partial_kernel + hidden_epilogue_kernel
//     P h_1                                              spmm_kernel
//     out  = (P h_1) W_1^T + rowsum(P) b_1^T             rowdot_kernel
// (P (X W^T + 1 b^T) = (P X) W^T + rowsum(P) b^T: the same numbers up to fp32 summation order.)
// At the Cora shape (BASELINE.json configs[1]: N = 2 708, F = 1 433, H = 64) the standard route was 9 launches -- two weight
// transposes, a split-K memset, a GEMM that waits for a barrier every 32 k, two SpMMs, mask bits, act' -- for 0.12 ms of
// device time; here the long-K product reads both operands along K straight into the MFMA operand layout (a lane's
// 16 bytes are four k steps of its row), every wave has its whole K share in flight, and K is split over waves and
// workgroups with the partial sums combined by the epilogue kernel that is needed anyway (no atomics, no memset).
#include "device_utils.h"
#include "lgnn_internal.h"

namespace lgnn {

namespace {

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };  // 16 bytes at any 4-byte boundary (rows of 1 433 floats)

__device__ __forceinline__ void load4(const float* __restrict__ p, int64_t k, int64_t ke, float (&v)[4]) {
  if (k + 4 <= ke) {
    const f4u t = *reinterpret_cast<const f4u*>(p + k);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = k + q < ke ? p[k + q] : 0.f;
  }
}

constexpr int kChunk = 8;    // k per step of a wave: lane (r, hi) holds k = 4 hi .. 4 hi + 3 of row r
constexpr int kInFlight = 5; // steps whose loads are issued before the first MFMA

// part[z][R][ldp] = A[:, slice z] W[:, slice z]^T for 32-row tiles; grid (row tiles, K slices), 4 waves interleave the
// slice's chunks.  A [R, lda] and W [Nout, ldw] are both row major (K contiguous): no transposed weight copy.
template <int NT>
__global__ __launch_bounds__(256) void gemm_nt_partial_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ W, int64_t ldw, int64_t R, int64_t K,
                                                              int64_t Nout, int64_t k_slice, float* __restrict__ part,
                                                              int64_t ldp) {
  __shared__ float red[4][32][NT * 32 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t row0 = int64_t(blockIdx.x) * 32;
  const int64_t kb = int64_t(blockIdx.y) * k_slice, ke = min(K, kb + k_slice);
  const float* __restrict__ arow = A + min(row0 + l31, R - 1) * lda;
  const float* __restrict__ wrow[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) wrow[n] = W + min(int64_t(n) * 32 + l31, Nout - 1) * ldw;  // columns >= Nout: never stored
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  for (int64_t k = kb + wave * kChunk; k < ke; k += int64_t(4) * kChunk * kInFlight) {
    float a[kInFlight][4], b[kInFlight][NT][4];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      const int64_t kk = k + int64_t(u) * 4 * kChunk + lhi * 4;  // >= ke: all zeros (no load)
      load4(arow, kk, ke, a[u]);
#pragma unroll
      for (int n = 0; n < NT; ++n) load4(wrow[n], kk, ke, b[u][n]);
    }
#pragma unroll
    for (int u = 0; u < kInFlight; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][n][t], acc[n], 0, 0, 0);
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lhi][n * 32 + l31] = acc[n][r];
  __syncthreads();
  float* __restrict__ dst = part + int64_t(blockIdx.y) * R * ldp;
  for (int f = tid; f < 32 * NT * 32; f += 256) {
    const int r = f / (NT * 32), c = f - r * (NT * 32);
    if (row0 + r < R && c < Nout) dst[(row0 + r) * ldp + c] = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
  }
}

// h_1[n][j] = act(sum_z part[z][n][j] + rowsum[n] b[j]); act'(h_1) as floats (diag.hip) and, for ReLU, as bit masks
// (bit j % 32 of word n * words + j / 32: what relu_mask_bits_kernel writes, backgemm.hip).  One thread per (n, padded j).
__global__ void hidden_epilogue_kernel(const float* __restrict__ part, int64_t nparts, int64_t N, int64_t H,
                                       const float* __restrict__ rowsum, const float* __restrict__ bias, int act,
                                       float* __restrict__ hout, float* __restrict__ dact, uint32_t* __restrict__ bits) {
  const int64_t words = (H + 31) / 32, Hp = words * 32;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n = t / Hp, j = t - n * Hp;
  const bool ok = n < N && j < H;
  float hv = 0.f;
  if (ok) {
    float z = rowsum[n] * bias[j];
    for (int64_t p = 0; p < nparts; ++p) z += part[(p * N + n) * H + j];
    hv = act_apply(z, act);
    hout[n * H + j] = hv;
    dact[n * H + j] = act_deriv_from_out(hv, act);
  }
  if (bits) {
    const uint64_t m = __ballot(ok && hv > 0.f);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0 && n < N) bits[t >> 5] = uint32_t(m >> (lane & 32));
  }
}

// out[n][c] = sum_j in[n][j] W[c][j] + rowsum[n] b[c]   (K = hidden width, C classes: 2 708 x 7 x 64 at the Cora shape)
__global__ void rowdot_kernel(const float* __restrict__ in, int64_t ld, int64_t N, int64_t K, int64_t C,
                              const float* __restrict__ W, const float* __restrict__ rowsum, const float* __restrict__ bias,
                              float* __restrict__ out) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  const int64_t n = t / C, c = t - n * C;
  const float* __restrict__ x = in + n * ld;
  const float* __restrict__ w = W + c * K;
  float s0 = 0.f, s1 = 0.f;
  int64_t j = 0;
  for (; j + 2 <= K; j += 2) { s0 = fmaf(x[j], w[j], s0); s1 = fmaf(x[j + 1], w[j + 1], s1); }
  if (j < K) s0 = fmaf(x[j], w[j], s0);
  out[t] = (s0 + s1) + rowsum[n] * bias[c];
}

}  // namespace

// Which models take this route (forward_ensure_aux): plain 2-layer GCN, hidden width one or two MFMA column tiles, and
// a problem small enough that launches and latency -- not bytes -- are its cost.
bool gcn2_small_forward_supported(const lgnn_ctx* h) {
  if (h->kind != LGNN_KIND_GCN || h->L != 2 || h->extras()) return false;
  const int64_t F = h->dims[0], H = h->dims[1], C = h->dims[2];
  return H <= 64 && C <= 64 && h->N * std::max(F, H) <= (int64_t(1) << 24) && h->N >= 32 && getenv("LGNN_NO_SMALL_FORWARD") == nullptr;
}

static int64_t gcn2_target_wgs() {  // (dev: LGNN_GCN2_WGS) workgroups of the long-K product: more K slices, shorter load chains
  static const int64_t n = getenv("LGNN_GCN2_WGS") ? std::max<int64_t>(64, atoll(getenv("LGNN_GCN2_WGS"))) : 512;
  return n;
}

int gcn2_forward_through_px(lgnn_ctx* h, hipStream_t s) {
  ForwardCache& fc = h->fc;
  const int64_t N = h->N, F = h->dims[0], H = h->dims[1], C = h->dims[2];
  LGNN_CALL(long_rows_fwd_ensure(h, s));
  const int32_t* lr = h->n_long_fwd > 0 ? (h->P.rowptr == h->PT.rowptr ? h->long_rows.as<int32_t>() : h->long_rows_fwd.as<int32_t>())
                                         : nullptr;
  const int64_t nlr = h->n_long_fwd > 0 ? h->n_long_fwd : 0;
  LGNN_CALL(forward_input_view(h, s));
  if (!(fc.x_valid && fc.px_valid)) LGNN_CALL(build_px(h, s));  // graph + features only: once per binding
  // enough (row tile, K slice) workgroups for the chip, slices a multiple of the 4 waves' chunk stride
  const int64_t row_tiles = cdiv(N, 32);
  const int64_t stride = int64_t(4) * kChunk;
  int64_t ks = std::max<int64_t>(1, std::min<int64_t>(cdiv(gcn2_target_wgs(), row_tiles), cdiv(F, stride * 2)));
  const int64_t k_slice = cdiv(cdiv(F, ks), stride) * stride;
  ks = cdiv(F, k_slice);
  LGNN_CALL(fc.tmp.reserve(size_t(ks) * N * H * 4));
  const dim3 grid{unsigned(row_tiles), unsigned(ks)};
  if (H <= 32)
    hipLaunchKernelGGL(gemm_nt_partial_kernel<1>, grid, dim3(256), 0, s, fc.prop_in[0].as<float>(), fc.prop_ld[0], h->W[0], F, N, F,
                       H, k_slice, fc.tmp.as<float>(), H);
  else
    hipLaunchKernelGGL(gemm_nt_partial_kernel<2>, grid, dim3(256), 0, s, fc.prop_in[0].as<float>(), fc.prop_ld[0], h->W[0], F, N, F,
                       H, k_slice, fc.tmp.as<float>(), H);
  LGNN_HIP_CHECK(hipGetLastError());
  const int64_t words = cdiv(H, 32);
  LGNN_CALL(fc.act_out[0].reserve(size_t(N) * H * 4));
  LGNN_CALL(fc.dact0.reserve(size_t(N) * H * 4));
  uint32_t* bits = nullptr;
  if (h->act == LGNN_ACT_RELU) {
    LGNN_CALL(fc.mask_bits[0].reserve(size_t(N) * words * 4 + 32));
    bits = fc.mask_bits[0].as<uint32_t>();
  }
  hipLaunchKernelGGL(hidden_epilogue_kernel, dim3(unsigned(cdiv(N * words * 32, 256))), dim3(256), 0, s, fc.tmp.as<float>(), ks, N,
                     H, fc.rowsum.as<float>(), h->b[0], h->act, fc.act_out[0].as<float>(), fc.dact0.as<float>(), bits);
  LGNN_HIP_CHECK(hipGetLastError());
  fc.hact_p[0] = fc.act_out[0].as<float>();
  fc.hact_ld[0] = H;
  fc.lin_in_p[1] = fc.hact_p[0];
  fc.lin_in_ld[1] = H;
  fc.prop_ld[1] = H;
  LGNN_CALL(fc.prop_in[1].reserve(size_t(N) * H * 4));
  LGNN_CALL(launch_spmm(h->P, N, fc.hact_p[0], H, fc.prop_in[1].as<float>(), H, H, 0, s, lr, nlr));
  LGNN_CALL(fc.out.reserve(size_t(N) * C * 4));
  hipLaunchKernelGGL(rowdot_kernel, dim3(unsigned(cdiv(N * C, 256))), dim3(256), 0, s, fc.prop_in[1].as<float>(), H, N, H, C,
                     h->W[1], fc.rowsum.as<float>(), h->b[1], fc.out.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn
