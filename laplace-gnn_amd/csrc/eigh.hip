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

#include "gram256.h"  // f32x16
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
// (nfull, off, vld: the matrix is the trailing n x n block at offset `off` of an nfull x nfull matrix whose first `off` columns
//  tridiag_stream_kernel has eliminated; reflector k goes to row off + k of V [nfull][vld] at coordinates off..off + n.
//  Stand-alone use: nfull = n, off = 0, vld = TN.)
__global__ __launch_bounds__(1024) void tridiag256_kernel(const float* __restrict__ A, int n, float* __restrict__ D,
                                                          float* __restrict__ E, float* __restrict__ V,
                                                          float* __restrict__ tau, int nfull, int off, int vld) {
  __shared__ __attribute__((aligned(16))) float xs[TN], vs[TN], ps[TN], ws[TN];
  const int tid = threadIdx.x, lane = tid & 63, r = tid >> 2, q = tid & 3;
  const int64_t b = blockIdx.x;
  const float* __restrict__ Ab = A + b * int64_t(nfull) * nfull + int64_t(off) * nfull + off;
  float* __restrict__ Db = D + b * int64_t(nfull) + off;
  float* __restrict__ Eb = E + b * int64_t(nfull) + off;
  float* __restrict__ Vb = V + b * int64_t(nfull) * vld + int64_t(off) * vld + off;
  float* __restrict__ tb = tau + b * int64_t(nfull) + off;
  float a[64];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * g + 4 * q + i;
      a[4 * g + i] = (r < n && c < n) ? Ab[int64_t(r) * nfull + c] : 0.f;
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
      Vb[int64_t(k) * vld + tid] = v;  // (embedded use: n == TN, so off + tid < nfull)
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

// ---- 256 < n <= 512 (n % 4 == 0): the first n - 256 columns ---------------------------------------------------------------
// 512 x 512 floats are two CUs' register files, so the register-resident kernel above cannot hold the matrix.  One 1024-thread
// workgroup per matrix eliminates the first n - 256 columns with the matrix in global memory (1 MB: L2 resident) by the blocked
// scheme of LAPACK's latrd: panels of 32 columns whose reflectors V and companions W live in LDS ([32][n] each, 128 KiB at
// n = 512), the trailing matrix is READ once per column (p = A v, every thread owns four columns and an eighth of the rows:
// row-contiguous 16-byte loads, no cross-lane reduction) and WRITTEN once per panel (A -= V W^T + W V^T).  The trailing 256 x 256
// block then goes to tridiag256_kernel.  Per column: x = A[k, :] - (V W^T + W V^T)[k, :]; Householder v, tau;
// w = tau (A v - V (W^T v) - W (V^T v)), w -= (tau / 2)(w^T v) v.
constexpr int SNB = 32;   // panel width
constexpr int SRG = 8;    // row groups of the matrix-vector product
__global__ __launch_bounds__(1024) void tridiag_stream_kernel(float* __restrict__ A, int n, int ncols, float* __restrict__ D,
                                                              float* __restrict__ E, float* __restrict__ V,
                                                              float* __restrict__ tau) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* __restrict__ Vp = sm;                    // [SNB][n]
  float* __restrict__ Wp = Vp + SNB * n;          // [SNB][n]
  float* __restrict__ pp = Wp + SNB * n;          // [SRG][n] partial products
  float* __restrict__ xs = pp + SRG * n;          // [n]
  float* __restrict__ vs = xs + n;                // [n]
  float* __restrict__ ps = vs + n;                // [n]
  float* __restrict__ tv = ps + n;                // [SNB] W^T v
  float* __restrict__ tw = tv + SNB;              // [SNB] V^T v
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & 127, rg = tid >> 7, c4 = 4 * cg;
  const int64_t b = blockIdx.x;
  float* __restrict__ Ab = A + b * int64_t(n) * n;
  float* __restrict__ Db = D + b * int64_t(n);
  float* __restrict__ Eb = E + b * int64_t(n);
  float* __restrict__ Vb = V + b * int64_t(n) * n;
  float* __restrict__ tb = tau + b * int64_t(n);
  const int per_lane = (n + 63) >> 6;
  for (int k0 = 0; k0 < ncols; k0 += SNB) {
    const int jn = min(SNB, ncols - k0);
    for (int j = 0; j < jn; ++j) {
      const int k = k0 + j;
      // (1) the current column with the panel's pending update applied (row k of the symmetric matrix: contiguous)
      if (tid < n) {
        float val = 0.f;
        if (tid >= k) {
          val = Ab[int64_t(k) * n + tid];
          float v2 = 0.f;  // (two chains, unrolled: the LDS reads of several panel columns are in flight together)
#pragma unroll 4
          for (int i = 0; i < j; ++i) {
            val -= Vp[i * n + tid] * Wp[i * n + k];
            v2 += Wp[i * n + tid] * Vp[i * n + k];
          }
          val -= v2;
        }
        xs[tid] = val;
      }
      __syncthreads();
      // (2) Householder vector (every wave: the same data in the same order)
      float sg = 0.f;
      for (int q = 0; q < per_lane; ++q) {
        const int c = lane + 64 * q;
        const float x = c < n ? xs[c] : 0.f;
        sg += c > k + 1 ? x * x : 0.f;
      }
      sg = wsum64(sg);
      const float alpha = xs[k + 1];
      float t = 0.f, beta = alpha, inv = 0.f;
      if (sg != 0.f) {
        const float nrm = sqrtf(alpha * alpha + sg);
        beta = alpha >= 0.f ? -nrm : nrm;
        t = (beta - alpha) / beta;
        inv = 1.f / (alpha - beta);
      }
      if (tid < n) {
        const float v = (sg == 0.f || tid <= k) ? 0.f : (tid == k + 1 ? 1.f : xs[tid] * inv);
        vs[tid] = v;
        Vp[j * n + tid] = v;
        Vb[int64_t(k) * n + tid] = v;
      }
      if (tid == 0) { Db[k] = xs[k]; Eb[k] = beta; tb[k] = t; }
      __syncthreads();
      if (sg == 0.f) {  // H = I: w = 0
        if (tid < n) Wp[j * n + tid] = 0.f;
        __syncthreads();
        continue;
      }
      // (3) p = A v over the trailing block (rows and columns > k), W^T v and V^T v of the panel's earlier columns
      {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 + 3 > k && c4 < n) {
          int r = k + 1 + rg;
          // eight rows per step: the loads of a step are all in flight before the first product (the loop is bound by the
          // L2 latency otherwise: four loads per thread gave 91 GB/s into the CU)
          for (; r + 7 * SRG < n; r += 8 * SRG) {
            float4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(Ab + int64_t(r + u * SRG) * n + c4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float vv = vs[r + u * SRG];
              acc.x += a[u].x * vv; acc.y += a[u].y * vv; acc.z += a[u].z * vv; acc.w += a[u].w * vv;
            }
          }
          for (; r < n; r += SRG) {
            const float4 a0 = *reinterpret_cast<const float4*>(Ab + int64_t(r) * n + c4);
            const float v0 = vs[r];
            acc.x += a0.x * v0; acc.y += a0.y * v0; acc.z += a0.z * v0; acc.w += a0.w * v0;
          }
        }
        if (c4 < n) *reinterpret_cast<float4*>(pp + rg * n + c4) = acc;
        for (int i = wave; i < j; i += 16) {
          float s1 = 0.f, s2 = 0.f;
          for (int q = 0; q < per_lane; ++q) {
            const int c = lane + 64 * q;
            if (c < n) { const float v = vs[c]; s1 += Wp[i * n + c] * v; s2 += Vp[i * n + c] * v; }
          }
          s1 = wsum64(s1); s2 = wsum64(s2);
          if (lane == 0) { tv[i] = s1; tw[i] = s2; }
        }
      }
      __syncthreads();
      if (tid < n) {
        float p = 0.f;
        if (tid > k) {
#pragma unroll
          for (int g = 0; g < SRG; ++g) p += pp[g * n + tid];
          float p2 = 0.f;
#pragma unroll 4
          for (int i = 0; i < j; ++i) {
            p -= Vp[i * n + tid] * tv[i];
            p2 += Wp[i * n + tid] * tw[i];
          }
          p -= p2;
        }
        ps[tid] = t * p;
      }
      __syncthreads();
      // (4) w = p - (tau / 2)(p^T v) v
      float ds = 0.f;
      for (int q = 0; q < per_lane; ++q) {
        const int c = lane + 64 * q;
        if (c < n) ds += ps[c] * vs[c];
      }
      ds = wsum64(ds);
      const float hd = 0.5f * t * ds;
      if (tid < n) Wp[j * n + tid] = ps[tid] - hd * vs[tid];
      __syncthreads();
    }
    // trailing update A -= V W^T + W V^T = [V | W] [W | V]^T on rows / columns >= kb: a rank-64 product on the matrix cores
    // (v_mfma_f32_32x32x2_f32, both operands straight from the LDS panels: lanes along rows resp. columns are consecutive
    // addresses), 32 x 32 output tiles dealt to the 16 waves, read-modify-write of the L2-resident matrix.  (As VALU code with
    // the panel operands re-read from LDS per row this update was bound by the LDS pipe: 200 us per panel.)
    const int kb = k0 + jn;
    if (jn < SNB) {  // a partial last panel: the unused panel columns multiply as zeros
      for (int q = tid; q < (SNB - jn) * n; q += 1024) { Vp[jn * n + q] = 0.f; Wp[jn * n + q] = 0.f; }
      __syncthreads();
    }
    {
      const int base = kb & ~31;
      const int nt = (n - base + 31) >> 5;
      const int l31 = lane & 31, lhi = lane >> 5;
      for (int tile = wave; tile < nt * nt; tile += 16) {
        const int tr = tile / nt, tc = tile - tr * nt;
        const int row = base + 32 * tr + l31, col = base + 32 * tc + l31;
        const int rr = row < n ? row : n - 1, cc = col < n ? col : n - 1;  // (clamped: out-of-range lanes are not stored)
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll 8
        for (int kk = 0; kk < SNB / 2; ++kk) {
          const int i = 2 * kk + lhi;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Vp[i * n + rr], Wp[i * n + cc], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wp[i * n + rr], Vp[i * n + cc], acc, 0, 0, 0);
        }
        if (col < n) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int r = base + 32 * tr + (q & 3) + 8 * (q >> 2) + 4 * lhi;
            if (r < n) Ab[int64_t(r) * n + col] -= acc[q];
          }
        }
      }
    }
    __syncthreads();
  }
}

// the back-transform for n <= 512: lane holds coordinates 4 lane + 256 q + i, q < 2
__global__ __launch_bounds__(256) void backtransform512_kernel(const float* __restrict__ Z, int n, const float* __restrict__ V,
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
  const float* __restrict__ Vb = V + b * int64_t(n) * n;
  const float* __restrict__ tb = tau + b * int64_t(n);
  const bool hi = 256 + 4 * lane < n;  // (n % 4 == 0)
  const bool lo = 4 * lane < n;
  float4 z0 = lo ? *reinterpret_cast<const float4*>(zr + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 z1 = hi ? *reinterpret_cast<const float4*>(zr + 256 + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = n - 2; k >= 0; --k) {
    const float t = tb[k];
    if (t == 0.f) continue;
    const float* __restrict__ vk = Vb + int64_t(k) * n;
    const float4 v0 = lo ? *reinterpret_cast<const float4*>(vk + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = hi ? *reinterpret_cast<const float4*>(vk + 256 + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    float d = v0.x * z0.x + v0.y * z0.y + v0.z * z0.z + v0.w * z0.w + v1.x * z1.x + v1.y * z1.y + v1.z * z1.z + v1.w * z1.w;
    d = wsum64(d) * t;
    z0.x -= d * v0.x; z0.y -= d * v0.y; z0.z -= d * v0.z; z0.w -= d * v0.w;
    z1.x -= d * v1.x; z1.y -= d * v1.y; z1.z -= d * v1.z; z1.w -= d * v1.w;
  }
  if (lo) *reinterpret_cast<float4*>(orow + 4 * lane) = z0;
  if (hi) *reinterpret_cast<float4*>(orow + 256 + 4 * lane) = z1;
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
                       g_e.as<float>(), g_v.as<float>(), g_tau.as<float>(), int(n), 0, TN);
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(stedc_all(n, batch, s));
    hipLaunchKernelGGL(backtransform_kernel, dim3(unsigned(cdiv(n, 4)), unsigned(batch)), dim3(256), 0, s, g_z.as<float>(),
                       int(n), g_v.as<float>(), g_tau.as<float>(), g_d.as<float>(), g_info.as<int32_t>(), A, W, info);
    LGNN_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (n > TN && n <= 2 * TN && n % 4 == 0 && getenv("LGNN_EIGH_LIBRARY") == nullptr) {
    // 256 < n <= 512: n - 256 columns by the streaming kernel, the trailing 256 x 256 block by the register-resident one,
    // divide and conquer on the tridiagonal matrix (library), back-transform: four launches + the library's chain
    LGNN_CALL(g_e.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_v.reserve(size_t(batch) * n * n * 4));
    LGNN_CALL(g_tau.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_d.reserve(size_t(batch) * n * 4));
    LGNN_CALL(g_z.reserve(size_t(batch) * n * n * 4));
    LGNN_CALL(g_info.reserve(size_t(batch) * 4));
    const size_t smem = size_t(2 * SNB * n + SRG * n + 3 * n + 2 * SNB) * 4;
    LGNN_REQUIRE(smem <= 160 * 1024, "symeig: panel does not fit the LDS");
    static bool attr_set = false;
    if (!attr_set) {
      LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tridiag_stream_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    LGNN_HIP_CHECK(hipMemsetAsync(g_v.p, 0, size_t(batch) * n * n * 4, s));  // (the trailing block's reflectors start at column n - 256)
    hipLaunchKernelGGL(tridiag_stream_kernel, dim3(unsigned(batch)), dim3(1024), smem, s, A, int(n), int(n - TN),
                       g_d.as<float>(), g_e.as<float>(), g_v.as<float>(), g_tau.as<float>());
    hipLaunchKernelGGL(tridiag256_kernel, dim3(unsigned(batch)), dim3(1024), 0, s, A, TN, g_d.as<float>(), g_e.as<float>(),
                       g_v.as<float>(), g_tau.as<float>(), int(n), int(n - TN), int(n));
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(stedc_all(n, batch, s));
    hipLaunchKernelGGL(backtransform512_kernel, dim3(unsigned(cdiv(n, 4)), unsigned(batch)), dim3(256), 0, s, g_z.as<float>(),
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
