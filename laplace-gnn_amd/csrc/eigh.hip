// Batched symmetric eigendecomposition of the Kronecker factors (the step right after the accumulation inside
// KronLaplace.fit: laplace/baselaplace.py:1610 -> laplace/utils/matrix.py:118-145 -> utils.py:193-226).
//
// The reference calls torch.linalg.eigh once per factor; on this stack that is one rocSOLVER syevd per matrix, each a
// chain of ~100 single-workgroup kernels (tridiagonalisation panels, divide and conquer): ~3.3 ms per 256 x 256 factor,
// 13 ms for the four distinct factors of the arxiv-shaped model -- strictly serial and using one CU.  Batched through ONE
// strided-batched syevd the decomposition costs what the largest factor costs (5.0 ms), 4 ms of it the Householder
// tridiagonalisation: 64 panel steps of seven tiny launches each plus a 2 ms single-workgroup tail kernel.
//
// n <= 256 (every factor of the BASELINE models) therefore takes a hand-written path:
//   tridiag256_kernel   one 1024-thread workgroup per matrix, the whole matrix in REGISTERS (thread (r, q) owns 64 entries
//                       of row r), the Householder vector / A v / w through 4 KB of LDS, four barriers per column and
//                       no launch in between: A = Q T Q^T, reflectors to a workspace                       (~0.5 ms)
//   rocsolver_sstedc    divide and conquer on T (eigenvectors of the tridiagonal matrix), one call per factor
//   backtransform_kernel  y = H_0 H_1 ... H_{n-2} z, one wave per eigenvector, reflectors streamed from L2
// (The caller pads smaller factors to the common size with a decoupled negative diagonal block, see matrix.py: a column
// whose off-diagonal part is exactly zero gets tau = 0 and costs one barrier.)  LGNN_EIGH_LIBRARY=1 forces the library
// path, which larger factors always take.
#include <cstdio>

#include <rocsolver/rocsolver.h>

#include "lgnn_internal.h"

namespace lgnn {
namespace {
rocblas_handle g_handle = nullptr;
// The library path (factors above 256 rows) has a handle and a workspace of its own: KronLaplace runs it on a side stream
// while the caller's stream still executes queued batches that use g_handle (rocBLAS handles are not meant to serve two
// streams at once: they carry the solver's device workspace).
rocblas_handle g_eig_handle = nullptr;
DevBuf g_e_large;  // off-diagonal workspace of the library path [batch, n]
DevBuf g_e;    // off-diagonal workspace [batch, n]
DevBuf g_v;    // Householder vectors [batch][n][256]
DevBuf g_tau;  // [batch][n]
DevBuf g_d;    // diagonal of T in, eigenvalues out [batch][n]
DevBuf g_z;    // eigenvectors of T [batch][n][n]
DevBuf g_info; // [batch]

// The divide and conquer calls of the factors are independent chains of ~75 tiny kernels each (0.9 ms per 256 x 256
// factor, launch after launch on one CU): each factor's chain goes to a side stream of its own (own rocBLAS handle and
// workspace), forked off and joined back into the caller's stream with events, so the chains run side by side.
// (Capturing the chains into a hipGraph was tried first: the solver performs an operation that is not permitted while a
// stream is capturing, so the fork / join is done with plain streams.)
constexpr int kBranches = 8;
rocblas_handle g_bh[kBranches] = {};
hipStream_t g_side[kBranches] = {};
hipEvent_t g_fork = nullptr, g_join[kBranches] = {};
constexpr int TN = 256;

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Thread (r = tid / 4, q = tid % 4) holds a[4 g + i] = A[r][16 g + 4 q + i], g < 16, i < 4: the four threads of a row read
// 64 contiguous bytes of an LDS vector per step (no bank conflict, rows of a wave broadcast).
__global__ __launch_bounds__(1024) void tridiag256_kernel(const float* __restrict__ A, int n, float* __restrict__ D,
                                                          float* __restrict__ E, float* __restrict__ V,
                                                          float* __restrict__ tau) {
  __shared__ __attribute__((aligned(16))) float xs[TN], vs[TN], ps[TN], ws[TN];
  const int tid = threadIdx.x, lane = tid & 63, r = tid >> 2, q = tid & 3;
  const int64_t b = blockIdx.x;
  const float* __restrict__ Ab = A + b * int64_t(n) * n;
  float* __restrict__ Db = D + b * int64_t(n);
  float* __restrict__ Eb = E + b * int64_t(n);
  float* __restrict__ Vb = V + b * int64_t(n) * TN;
  float* __restrict__ tb = tau + b * int64_t(n);
  float a[64];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * g + 4 * q + i;
      a[4 * g + i] = (r < n && c < n) ? Ab[int64_t(r) * n + c] : 0.f;
    }
  for (int k = 0; k < n - 1; ++k) {
    const int g0 = (k + 1) >> 4;  // column groups below hold columns <= k only
    if (r == k) {
#pragma unroll
      for (int g = 0; g < 16; ++g)
        *reinterpret_cast<float4*>(&xs[16 * g + 4 * q]) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
    }
    __syncthreads();
    // every wave: sigma = sum_{c > k+1} x_c^2 (identical in all waves: same data, same order)
    float sg = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      const float x = xs[c];
      sg += c > k + 1 ? x * x : 0.f;
    }
    sg = wsum64(sg);
    const float alpha = xs[k + 1];
    if (tid == 0) Db[k] = xs[k];
    if (sg == 0.f) {  // nothing to eliminate: H = I
      if (tid == 0) { Eb[k] = alpha; tb[k] = 0.f; }
      __syncthreads();  // xs is rewritten by the next column
      continue;
    }
    const float nrm = sqrtf(alpha * alpha + sg);
    const float beta = alpha >= 0.f ? -nrm : nrm;
    const float t = (beta - alpha) / beta;
    const float inv = 1.f / (alpha - beta);
    if (tid < TN) {
      const float v = tid <= k ? 0.f : (tid == k + 1 ? 1.f : xs[tid] * inv);
      vs[tid] = v;
      Vb[int64_t(k) * TN + tid] = v;
    }
    if (tid == 0) { Eb[k] = beta; tb[k] = t; }
    __syncthreads();
    // p = tau A v on the trailing rows
    if (r > k) {
      float part = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (g >= g0) {
          const float4 v4 = *reinterpret_cast<const float4*>(&vs[16 * g + 4 * q]);
          part = fmaf(a[4 * g], v4.x, part); part = fmaf(a[4 * g + 1], v4.y, part);
          part = fmaf(a[4 * g + 2], v4.z, part); part = fmaf(a[4 * g + 3], v4.w, part);
        }
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      if (q == 0) ps[r] = t * part;
    } else if (q == 0) {
      ps[r] = 0.f;
    }
    __syncthreads();
    // w = p - (tau / 2) (p^T v) v
    float ds = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ds += ps[lane + 64 * j] * vs[lane + 64 * j];
    ds = wsum64(ds);
    const float hd = 0.5f * t * ds;
    if (tid < TN) ws[tid] = ps[tid] - hd * vs[tid];
    __syncthreads();
    // A -= v w^T + w v^T on the trailing block
    if (r > k) {
      const float vr = vs[r], wr = ws[r];
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (g >= g0) {
          const float4 v4 = *reinterpret_cast<const float4*>(&vs[16 * g + 4 * q]);
          const float4 w4 = *reinterpret_cast<const float4*>(&ws[16 * g + 4 * q]);
          a[4 * g] -= vr * w4.x + wr * v4.x; a[4 * g + 1] -= vr * w4.y + wr * v4.y;
          a[4 * g + 2] -= vr * w4.z + wr * v4.z; a[4 * g + 3] -= vr * w4.w + wr * v4.w;
        }
    }
    // (the next column's row goes to xs, which nobody reads after the second barrier of this step)
  }
  if (r == n - 1) {
#pragma unroll
    for (int g = 0; g < 16; ++g)
      *reinterpret_cast<float4*>(&xs[16 * g + 4 * q]) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  }
  __syncthreads();
  if (tid == 0) { Db[n - 1] = xs[n - 1]; Eb[n - 1] = 0.f; tb[n - 1] = 0.f; }
}

// Row j of Z (memory row = eigenvector j of T, the solver's column j) -> Q z = H_0 (H_1 (... H_{n-2} z)) into row j of the
// caller's matrix; one wave per eigenvector, lane = 4 coordinates.  Also hands the eigenvalues and the status over.
__global__ __launch_bounds__(256) void backtransform_kernel(const float* __restrict__ Z, int n, const float* __restrict__ V,
                                                            const float* __restrict__ tau, const float* __restrict__ lam,
                                                            const int32_t* __restrict__ info_in, float* __restrict__ out,
                                                            float* __restrict__ W, int32_t* __restrict__ info) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = blockIdx.y;
  if (j >= n) return;
  if (lane == 0) {
    W[b * n + j] = lam[b * n + j];
    if (j == 0) info[b] = info_in[b];
  }
  const float* __restrict__ zr = Z + (b * n + j) * int64_t(n);
  float* __restrict__ orow = out + (b * n + j) * int64_t(n);
  const float* __restrict__ Vb = V + b * int64_t(n) * TN;
  const float* __restrict__ tb = tau + b * int64_t(n);
  float z[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) z[i] = 4 * lane + i < n ? zr[4 * lane + i] : 0.f;
  for (int k = n - 2; k >= 0; --k) {
    const float t = tb[k];
    if (t == 0.f) continue;
    const float4 v = *reinterpret_cast<const float4*>(Vb + int64_t(k) * TN + 4 * lane);
    float d = v.x * z[0] + v.y * z[1] + v.z * z[2] + v.w * z[3];
    d = wsum64(d) * t;
    z[0] -= d * v.x; z[1] -= d * v.y; z[2] -= d * v.z; z[3] -= d * v.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (4 * lane + i < n) orow[4 * lane + i] = z[i];
}

int stedc_branch(int br, int branches, int64_t n, int64_t batch, hipStream_t st) {
  if (rocblas_set_stream(g_bh[br], st) != rocblas_status_success) { set_error("rocblas_set_stream failed"); return 3; }
  for (int64_t b = br; b < batch; b += branches) {
    const rocblas_status rs = rocsolver_sstedc(g_bh[br], rocblas_evect_tridiagonal, rocblas_int(n), g_d.as<float>() + b * n,
                                               g_e.as<float>() + b * n, g_z.as<float>() + b * n * n, rocblas_int(n),
                                               g_info.as<int32_t>() + b);
    if (rs != rocblas_status_success) { set_error("rocsolver_sstedc failed"); return 3; }
  }
  return 0;
}

// eigenpairs of the tridiagonal matrices in g_d / g_e -> g_d (values), g_z (vectors), g_info; on stream s
int stedc_all(int64_t n, int64_t batch, hipStream_t s) {
  const int branches = int(std::min<int64_t>(batch, kBranches));
  for (int i = 0; i < branches; ++i) {
    if (!g_bh[i] && rocblas_create_handle(&g_bh[i]) != rocblas_status_success) { set_error("rocblas_create_handle failed"); return 3; }
    if (i > 0 && !g_side[i]) LGNN_HIP_CHECK(hipStreamCreateWithFlags(&g_side[i], hipStreamNonBlocking));
    if (!g_join[i]) LGNN_HIP_CHECK(hipEventCreateWithFlags(&g_join[i], hipEventDisableTiming));
  }
  if (!g_fork) LGNN_HIP_CHECK(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
  if (branches == 1 || getenv("LGNN_EIGH_ONE_STREAM") != nullptr) {
    for (int i = 0; i < branches; ++i) LGNN_CALL(stedc_branch(i, branches, n, batch, s));
    return 0;
  }
  LGNN_HIP_CHECK(hipEventRecord(g_fork, s));
  for (int i = 0; i < branches; ++i) {
    hipStream_t st = i == 0 ? s : g_side[i];
    if (i > 0) LGNN_HIP_CHECK(hipStreamWaitEvent(st, g_fork, 0));
    LGNN_CALL(stedc_branch(i, branches, n, batch, st));
    if (i > 0) {
      LGNN_HIP_CHECK(hipEventRecord(g_join[i], st));
      LGNN_HIP_CHECK(hipStreamWaitEvent(s, g_join[i], 0));
    }
  }
  return 0;
}
}  // namespace
}  // namespace lgnn

namespace lgnn {
// the process-wide rocBLAS handle, bound to stream `s` (also used by jacobian.hip); nullptr on failure
void* blas_handle(hipStream_t s) {
  if (!g_handle && rocblas_create_handle(&g_handle) != rocblas_status_success) return nullptr;
  if (rocblas_set_stream(g_handle, s) != rocblas_status_success) return nullptr;
  return g_handle;
}
}  // namespace lgnn

using namespace lgnn;

// A: [batch][n][n] fp32, symmetric, overwritten: row j of matrix b = eigenvector j (unit norm) for eigenvalue W[b][j],
// eigenvalues ascending (column-major eigenvector columns of the solver == rows of the row-major view).
// info: device int32 [batch], 0 = converged.  Asynchronous on `stream` (apart from workspace growth on first use).
extern "C" int lgnn_symeig_batched(float* A, int64_t n, int64_t batch, float* W, int32_t* info, void* stream) {
  if (!A || !W || !info) { set_error("null argument"); return 2; }
  LGNN_REQUIRE(n > 0 && n <= 32768 && batch > 0 && batch <= 65535, "symeig: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (neither branch touches the shared g_handle: the divide and conquer chains have handles of their own, g_bh, and the
  //  library path g_eig_handle -- queued main-stream work that uses g_handle is never rebound to a side stream from here)
  if (n <= TN && n >= 2 && getenv("LGNN_EIGH_LIBRARY") == nullptr) {
    LGNN_CALL(g_e.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_v.reserve(size_t(batch) * n * TN * 4));
    LGNN_CALL(g_tau.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_d.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_z.reserve(size_t(batch) * n * n * 4));
    LGNN_CALL(g_info.reserve(size_t(batch) * 4));
    hipLaunchKernelGGL(tridiag256_kernel, dim3(unsigned(batch)), dim3(1024), 0, s, A, int(n), g_d.as<float>(),
                       g_e.as<float>(), g_v.as<float>(), g_tau.as<float>());
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(stedc_all(n, batch, s));
    hipLaunchKernelGGL(backtransform_kernel, dim3(unsigned(cdiv(n, 4)), unsigned(batch)), dim3(256), 0, s, g_z.as<float>(),
                       int(n), g_v.as<float>(), g_tau.as<float>(), g_d.as<float>(), g_info.as<int32_t>(), A, W, info);
    LGNN_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (!g_eig_handle && rocblas_create_handle(&g_eig_handle) != rocblas_status_success) { set_error("rocBLAS handle"); return 3; }
  if (rocblas_set_stream(g_eig_handle, s) != rocblas_status_success) { set_error("rocBLAS stream setup failed"); return 3; }
  LGNN_CALL(g_e_large.reserve(size_t(batch) * n * 4));
  const rocblas_status st = rocsolver_ssyevd_strided_batched(
      g_eig_handle, rocblas_evect_original, rocblas_fill_upper, rocblas_int(n), A, rocblas_int(n), rocblas_stride(n * n), W,
      rocblas_stride(n), g_e_large.as<float>(), rocblas_stride(n), info, rocblas_int(batch));
  if (st != rocblas_status_success) { set_error("rocsolver_ssyevd_strided_batched failed"); return 3; }
  return 0;
}
