// Gradient of the negative log marginal likelihood of a KronLaplace fit w.r.t. the adjacency ("next" row 8(f)-4).
//
// Reference: gnn/marglik_training.py:197-216 calls ``neg_marglik.backward()`` and steps an optimiser on ``model.adj``:
// autograd walks back through log_marginal_likelihood (laplace/baselaplace.py:938-973), the eigendecomposed factors
// (laplace/utils/matrix.py:118-145, 371-394), the KFAC accumulation that the fork keeps attached to the graph
// (curvlinops/kfac.py:637-661 non-detached Hessian square root + create_graph=True, :789-790 / :836-837 clone instead
// of detach), three dense full-graph forwards, normalize_adj (gnn/models/utils.py:106-112) and the straight-through
// binarisation (gnn/models/utils.py:42-86, gnn/models/models.py:103-118).
//
// Here the same chain runs in reverse as explicit kernels on the stored sparsity pattern (2-layer GCN, ReLU):
//   host (python):   Gamma_B_l = d(neg marglik)/dB_l, Gamma_A_l = d/dA_l from the eigenpairs of the fitted factors
//   per batch:       seeds V, g1 = P^T scatter(V), u = mask * (g1 W1), g0 = P^T u               (kfac.hip's chain)
//                    g0bar = 2 g0 Gamma_B0            gradP[(a,b)] += <u[a], g0bar[b]>             SDDMM, 256 wide
//                    ubar  = mask * (P g0bar)         g1bar = ubar W1^T + 2 g1 Gamma_B1
//                    gradP[(a,b)] += <G[a], g1bar[b]> (a in batch)      Vbar = (P g1bar)[batch]  SDDMM + row gather
//                    fbar = dV/df . Vbar (the fork's attached square root) + softmax - onehot      outbar[batch] += fbar
//   once per fit:    gradP += <outbar[a], Z1[b]>;  H1bar = (P^T outbar) W1 + 2 (T/N) H1 Gamma_A1
//                    gradP += <(mask * H1bar)[a], Z0[b]>
//                    normalize_adj backward: P[a,b] = d_a A[b,a] d_b, d = rowsum(A)^-1/2; diagonal -> 0 (overwritten
//                    by fill_diagonal_(1) after the STE); symmetric models average (i,j) and (j,i)
// The dense GEMMs with the (small) Gamma / weight matrices are plain library calls (rocBLAS).
//
// GraphSAGE (STEGraphSAGE, gnn/models/models.py:121-183; mean aggregation P = A / rowsum, gnn/models/layers.py:18-24): the
// top layer is not propagated, so everything per batch lives on the batch rows and their neighbours -- sample-major rows
// r = (m, c) instead of planes over all N nodes:
//   per batch:       DCAT = S W1 [M C, 2H] (S = the seed blocks)     dh[v] = DCAT_self[v] + sum_{u in batch} P[u,v] DCAT_neigh[u]
//                    g0 = mask * dh on the ACTIVE rows (batch + neighbours, compacted)         g0bar = 2 g0 Gamma_B0
//                    gradP[(a,b)] += <DCAT_neigh[a], mask_b * g0bar[b]>,  dcatbar[a] = [mask_a g0bar[a] | sum_b P[a,b] mask_b g0bar[b]]
//                    g1bar = dcatbar W1^T + 2 S Gamma_B1 = Vbar                                   (then as above)
//   once per fit:    T1 = outbar W1 + 2 (T/N) cat1 Gamma_A1;  gradP += <T1_neigh[a], H1[b]>;  Z0bar = mask * (T1_self + P^T T1_neigh)
//                    XNbar = Z0bar W0_neigh + 2 (T/N) (cat0 Gamma_A0)_neigh;  gradP += <XNbar[a], X[b]>
//                    mean_agg backward: gA[a,b] = (gradP[a,b] - sum_j gradP[a,j] P[a,j]) / max(rowsum_a, 1)
#include <rocblas/rocblas.h>

#include "device_utils.h"
#include "lgnn_internal.h"

namespace lgnn {

void* blas_handle(hipStream_t s);  // eigh.hip

namespace {

// out[p] += sum_planes <L_c[a, :], R_c[b, :]> for every stored entry p = (a, b) of the CSR, rows a from an optional list.
// LPR lanes (a power of two >= width / 4, at most 64) share an entry; 64 / LPR entries are in flight per wave.
template <int LPR>
__global__ __launch_bounds__(256) void sddmm_planes_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           int64_t nrows, const int32_t* __restrict__ rows,
                                                           const int32_t* __restrict__ nrows_dev,
                                                           const float* __restrict__ Lp, int64_t l_ld, int64_t l_stride,
                                                           const float* __restrict__ Rp, int64_t r_ld, int64_t r_stride,
                                                           int64_t width, int64_t nplanes, float* __restrict__ out) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, sl = lane % LPR;
  const int64_t total = rows ? int64_t(*nrows_dev) : nrows;
  for (int64_t w = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); w < total; w += int64_t(gridDim.x) * 4) {
    const int64_t a = rows ? rows[w] : w;
    const int32_t s = rowptr[a], e = rowptr[a + 1];
    for (int32_t p0 = s; p0 < e; p0 += EPW) {
      const int32_t p = p0 + sub;
      const bool ok = p < e;
      const int64_t b = ok ? col[p] : 0;
      float acc = 0.f;
      // (small graphs with many planes -- the (sample, class) planes of the res / norm diagonal route: blockIdx.y splits the
      //  planes so that the launch fills the device; the parts meet in out[p] through float atomics)
      const int64_t cper = (nplanes + gridDim.y - 1) / gridDim.y;
      const int64_t cb = int64_t(blockIdx.y) * cper, ce = cb + cper < nplanes ? cb + cper : nplanes;
      for (int64_t c = cb; c < ce; ++c) {
        const float* __restrict__ lrow = Lp + c * l_stride + a * l_ld;
        const float* __restrict__ rrow = Rp + c * r_stride + b * r_ld;
        for (int64_t k = sl; k < width; k += LPR) acc += ok ? lrow[k] * rrow[k] : 0.f;
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (ok && sl == 0) {
        if (gridDim.y > 1) atomicAdd(&out[p], acc);
        else out[p] += acc;  // one wave owns row a: a single writer per entry inside a launch
      }
    }
  }
}

int launch_sddmm(const Csr& m, int64_t nrows, const int32_t* rows, const int32_t* nrows_dev, const float* L, int64_t l_ld,
                 int64_t l_stride, const float* R, int64_t r_ld, int64_t r_stride, int64_t width, int64_t nplanes,
                 float* out, hipStream_t s) {
  if (nrows <= 0 || width <= 0 || nplanes <= 0) return 0;
  const unsigned gx = unsigned(std::min<int64_t>(cdiv(nrows, 4), 8192));
  const unsigned gy = unsigned(std::max<int64_t>(1, std::min<int64_t>({nplanes, cdiv(2048, gx), int64_t(1024)})));
  const dim3 grid{gx, gy, 1};
#define LGNN_SDDMM(LPRV)                                                                                           \
  hipLaunchKernelGGL(sddmm_planes_kernel<LPRV>, grid, dim3(256), 0, s, m.rowptr, m.col, nrows, rows, nrows_dev, \
                     L, l_ld, l_stride, R, r_ld, r_stride, width, nplanes, out)
  if (width <= 8) LGNN_SDDMM(8);
  else if (width <= 16) LGNN_SDDMM(16);
  else if (width <= 32) LGNN_SDDMM(32);
  else LGNN_SDDMM(64);
#undef LGNN_SDDMM
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// The two consumers of g0bar's rows in ONE pass over the entries of the active rows a of P (they gather the same rows):
//     out[(a, b)] += sum_c <u_c[a], R_c[b]>                      (the SDDMM of the step g0 = P^T u)
//     U_c[a]       = dact[a] * sum_b P[a, b] R_c[b]               (ubar: the SpMM of the step behind it; overwrites u)
// One wave per row; PC planes at a time: the row's u_c and the running sums stay in registers, every gathered row of R
// is used for both.  Halves the gather traffic of the two steps and skips the 42 % of rows whose u is zero.
template <int PC>
__global__ __launch_bounds__(256) void sddmm_spmm_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, const int32_t* __restrict__ rows,
                                                         const int32_t* __restrict__ nrows_dev, float* __restrict__ U,
                                                         const float* __restrict__ R, int64_t N, int64_t width,
                                                         int64_t nplanes, const float* __restrict__ dact,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int c0 = lane * 4;
  const bool col_ok = c0 < width;  // width % 4 == 0, <= 256 (launcher)
  const bool odd = (lane & 1) != 0;
  const int64_t total = rows ? int64_t(*nrows_dev) : N;
  const int64_t plane = N * width;
  for (int64_t w = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); w < total; w += int64_t(gridDim.x) * 4) {
    const int64_t a = rows ? int64_t(rows[w]) : w;
    const int32_t s = rowptr[a], e = rowptr[a + 1];
    float4 dm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok) dm = *reinterpret_cast<const float4*>(dact + a * width + c0);
    for (int64_t pc0 = 0; pc0 < nplanes; pc0 += PC) {
      float4 u[PC], acc[PC];
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        u[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && pc0 + c < nplanes) u[c] = *reinterpret_cast<const float4*>(U + (pc0 + c) * plane + a * width + c0);
      }
      // two entries per step: 2 * PC row slices in flight per wave, and ONE shuffle tree for both dot products (after the
      // first exchange even lanes carry entry p, odd lanes entry p + 1)
      for (int32_t p = s; p < e; p += 2) {
        const bool two = p + 1 < e;
        const int64_t b0 = col[p], b1 = two ? col[p + 1] : b0;
        const float v0 = val[p], v1 = two ? val[p + 1] : 0.f;
        float4 x0[PC], x1[PC];
#pragma unroll
        for (int c = 0; c < PC; ++c) {
          x0[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          x1[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col_ok && pc0 + c < nplanes) {
            x0[c] = *reinterpret_cast<const float4*>(R + (pc0 + c) * plane + b0 * width + c0);
            x1[c] = *reinterpret_cast<const float4*>(R + (pc0 + c) * plane + b1 * width + c0);
          }
        }
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int c = 0; c < PC; ++c) {
          acc[c].x = fmaf(v0, x0[c].x, acc[c].x); acc[c].y = fmaf(v0, x0[c].y, acc[c].y);
          acc[c].z = fmaf(v0, x0[c].z, acc[c].z); acc[c].w = fmaf(v0, x0[c].w, acc[c].w);
          acc[c].x = fmaf(v1, x1[c].x, acc[c].x); acc[c].y = fmaf(v1, x1[c].y, acc[c].y);
          acc[c].z = fmaf(v1, x1[c].z, acc[c].z); acc[c].w = fmaf(v1, x1[c].w, acc[c].w);
          d0 += u[c].x * x0[c].x + u[c].y * x0[c].y + u[c].z * x0[c].z + u[c].w * x0[c].w;
          d1 += u[c].x * x1[c].x + u[c].y * x1[c].y + u[c].z * x1[c].z + u[c].w * x1[c].w;
        }
        float d = (odd ? d1 : d0) + __shfl_xor(odd ? d0 : d1, 1);
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) d += __shfl_xor(d, o);
        if (lane == 0) out[p] += d;
        if (lane == 1 && two) out[p + 1] += d;
      }
#pragma unroll
      for (int c = 0; c < PC; ++c)
        if (col_ok && pc0 + c < nplanes)
          *reinterpret_cast<float4*>(U + (pc0 + c) * plane + a * width + c0) =
              make_float4(dm.x * acc[c].x, dm.y * acc[c].y, dm.z * acc[c].z, dm.w * acc[c].w);
    }
  }
}

// Plane-major form of the kernel above: grid (row groups, planes), one wave per (row, plane).  The kernel above gathers
// PC planes at once -- a working set of PC * 173 MB at the arxiv shape, far beyond the 256 MiB Infinity Cache, so its
// gathers go to HBM (3.3 TB/s measured); here all CUs work on ONE plane at a time (blocks are dispatched x first), which
// stays cache resident while its rows are gathered ~15 times each.  Eight entries per step; their eight dot products
// go through one shuffle tree (three halving exchanges leave lane l with entry l % 8, three more finish it); the
// planes' contributions to out[p] meet through float atomics.
__global__ __launch_bounds__(256) void sddmm_spmm_pm_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ val, const int32_t* __restrict__ rows,
                                                            const int32_t* __restrict__ nrows_dev, float* __restrict__ U,
                                                            int64_t u_plane_stride, const float* __restrict__ R, int64_t N,
                                                            int64_t width, const float* __restrict__ dact,
                                                            float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int c0 = lane * 4;
  const bool col_ok = c0 < width;
  const int64_t total = rows ? int64_t(*nrows_dev) : N;
  const int64_t w = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (w >= total) return;
  const int64_t a = rows ? int64_t(rows[w]) : w;
  const int64_t plane = blockIdx.y;
  float* __restrict__ Up = U + plane * u_plane_stride;
  const float* __restrict__ Rp = R + plane * N * width;
  const int32_t s = rowptr[a], e = rowptr[a + 1];
  float4 u = make_float4(0.f, 0.f, 0.f, 0.f), dm = u, acc = u;
  if (col_ok) {
    u = *reinterpret_cast<const float4*>(Up + a * width + c0);
    dm = *reinterpret_cast<const float4*>(dact + a * width + c0);
  }
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  for (int32_t base = s; base < e; base += 64) {
    const int32_t p = base + lane;
    const int32_t cj = p < e ? col[p] : int32_t(a);  // padding entries re-read the row's own slice (cached), weight 0
    const float cv = p < e ? val[p] : 0.f;
    const int n = min(64, int(e - base));
    for (int u0 = 0; u0 < n; u0 += 8) {
      float4 x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t b = __builtin_amdgcn_readlane(cj, min(u0 + k, 63));
        x[k] = col_ok ? *reinterpret_cast<const float4*>(Rp + b * width + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), min(u0 + k, 63)));
        acc.x = fmaf(v, x[k].x, acc.x); acc.y = fmaf(v, x[k].y, acc.y);
        acc.z = fmaf(v, x[k].z, acc.z); acc.w = fmaf(v, x[k].w, acc.w);
        d[k] = u.x * x[k].x + u.y * x[k].y + u.z * x[k].z + u.w * x[k].w;
      }
      float e4[4], f2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) e4[i] = (b0 ? d[2 * i + 1] : d[2 * i]) + __shfl_xor(b0 ? d[2 * i] : d[2 * i + 1], 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) f2[i] = (b1 ? e4[2 * i + 1] : e4[2 * i]) + __shfl_xor(b1 ? e4[2 * i] : e4[2 * i + 1], 2);
      float g = (b2 ? f2[1] : f2[0]) + __shfl_xor(b2 ? f2[0] : f2[1], 4);
      g += __shfl_xor(g, 8);
      g += __shfl_xor(g, 16);
      g += __shfl_xor(g, 32);
      // lane l < 8 holds the dot product of entry u0 + (l & 7) = u0 + 4 b2 + 2 b1 + b0
      if (lane < 8 && u0 + lane < n) atomicAdd(&out[base + u0 + lane], g);
    }
  }
  if (col_ok)
    *reinterpret_cast<float4*>(Up + a * width + c0) = make_float4(dm.x * acc.x, dm.y * acc.y, dm.z * acc.z, dm.w * acc.w);
}

// y[plane][r] = sum_p val[p] in[plane][col[p]] for rows of up to 256 columns, one wave per row, ENTRIES WITH A ZERO VALUE
// ARE NOT GATHERED (the per-batch copy of P^T's values has zeros at the columns of all-zero source rows: 42 % of the entries
// at the arxiv shape).  The plane is a buffer resource and the row a part of the vector offset: a dead slot's offset is out
// of range, the load returns zeros without touching memory, so the loop has no branch; 8 rows in flight per wave.
using srd_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__global__ __launch_bounds__(256) void spmm256_skip_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ val, int64_t N, const float* __restrict__ in,
                                                           int64_t in_plane_stride, int64_t width, float* __restrict__ out) {
  constexpr int UNR = 8;
  constexpr uint32_t kDead = 0xfffff000u;
  const int lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const int64_t plane = blockIdx.y;
  const uint32_t plane_bytes = uint32_t(N * width * 4);
  const srd_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + plane * in_plane_stride), 0, plane_bytes, 0x00020000);
  const uint32_t lane_off = lane * 4 < width ? uint32_t(lane) * 16u : kDead;
  const uint32_t row_bytes = uint32_t(width) * 4u;
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t base = s; base < e; base += 64) {
    const int32_t p = base + lane;
    float cv = 0.f;
    uint32_t cj = kDead;
    if (p < e) {
      cv = val[p];
      if (cv != 0.f) cj = uint32_t(col[p]) * row_bytes;
    }
    const int n = min(64, int(e - base));
    for (int u0 = 0; u0 < n; u0 += UNR) {
      float4 x[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const uint32_t so = uint32_t(__builtin_amdgcn_readlane(int(cj), min(u0 + u, 63)));
        // (a dead slot or a dead lane: kDead + anything below 4 KiB stays out of range of a < 4 GiB - 8 KiB plane)
        const uint32_t voff = so == kDead || lane_off == kDead ? kDead : so + lane_off;
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(srd, int(voff), 0, 0);
        x[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cv), min(u0 + u, 63)));
        y.x = fmaf(v, x[u].x, y.x); y.y = fmaf(v, x[u].y, y.y); y.z = fmaf(v, x[u].z, y.z); y.w = fmaf(v, x[u].w, y.w);
      }
    }
  }
  if (lane * 4 < width) *reinterpret_cast<float4*>(out + (plane * N + row) * width + lane * 4) = y;
}

// rows that are not active: ubar = 0 there (u was zero and stays zero: nothing to do; the planes were written in full by
// the GEMM, whose inactive rows are zeros already)

__global__ void mask_values_by_flag_kernel(const int32_t* __restrict__ col, const float* __restrict__ val, int64_t nnz,
                                           const uint8_t* __restrict__ active, float* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < nnz; p += stride) out[p] = active[col[p]] ? val[p] : 0.f;
}

// The same contraction on an arbitrary list of (a, b) pairs (candidate edges that are NOT stored: the reference's dense
// adj.grad has an entry for every pair, which is how its structure learning proposes new edges).
template <int LPR>
__global__ __launch_bounds__(256) void sddmm_coo_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb, int64_t K,
                                                        const float* __restrict__ Lp, int64_t l_ld, int64_t l_stride,
                                                        const float* __restrict__ Rp, int64_t r_ld, int64_t r_stride,
                                                        int64_t width, int64_t nplanes, const uint8_t* __restrict__ a_active,
                                                        float* __restrict__ out) {
  constexpr int EPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, sl = lane % LPR;
  const int64_t k = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * EPW + sub;
  const bool ok = k < K;
  const int64_t a = ok ? ca[k] : 0, b = ok ? cb[k] : 0;
  float acc = 0.f;
  if (ok && (a_active == nullptr || a_active[a])) {
    for (int64_t c = 0; c < nplanes; ++c) {
      const float* __restrict__ lrow = Lp + c * l_stride + a * l_ld;
      const float* __restrict__ rrow = Rp + c * r_stride + b * r_ld;
      for (int64_t q = sl; q < width; q += LPR) acc += lrow[q] * rrow[q];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (ok && sl == 0) out[k] += acc;
}

int launch_sddmm_coo(const int32_t* ca, const int32_t* cb, int64_t K, const float* L, int64_t l_ld, int64_t l_stride,
                     const float* R, int64_t r_ld, int64_t r_stride, int64_t width, int64_t nplanes, const uint8_t* a_active,
                     float* out, hipStream_t s) {
  if (K <= 0 || width <= 0 || nplanes <= 0) return 0;
#define LGNN_SDDMM_COO(LPRV)                                                                                               \
  hipLaunchKernelGGL(sddmm_coo_kernel<LPRV>, dim3(unsigned(cdiv(K, 4 * (64 / LPRV)))), dim3(256), 0, s, ca, cb, K, L, l_ld, \
                     l_stride, R, r_ld, r_stride, width, nplanes, a_active, out)
  if (width <= 8) LGNN_SDDMM_COO(8);
  else if (width <= 16) LGNN_SDDMM_COO(16);
  else if (width <= 32) LGNN_SDDMM_COO(32);
  else LGNN_SDDMM_COO(64);
#undef LGNN_SDDMM_COO
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// candidate pairs against the scattered seed planes: only pairs whose a is a batch node contribute
__global__ __launch_bounds__(256) void sddmm_coo_seed_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb,
                                                             int64_t K, int64_t N, int64_t C, const int32_t* __restrict__ pos,
                                                             const float* __restrict__ seeds, const float* __restrict__ g1bar,
                                                             int64_t c0, int64_t cc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t k = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int64_t a = ca[k], b = cb[k];
  const int32_t m = pos[a];
  if (m == INT32_MAX) return;
  const float* __restrict__ sd = seeds + int64_t(m) * C * C + c0 * C;
  const int64_t nq = cc * C;
  float acc = 0.f;
  for (int64_t q = lane; q < nq; q += 64) {
    const int64_t c = q / C, kk = q - c * C;
    acc += sd[q] * g1bar[(c * N + b) * C + kk];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[k] += acc;
}

// candidate (a, b) of P = entry (i = b, j = a) of A:  gA = gP d_i d_j - 1/2 d_i^2 (rs[i] + cs[i]);  i == j -> 0
__global__ void cand_adj_grad_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb, int64_t K,
                                     const int32_t* __restrict__ a_rowptr, const float* __restrict__ gPc,
                                     const float* __restrict__ rs, const float* __restrict__ cs, float* __restrict__ out) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t j = ca[k], i = cb[k];
  if (i == j) { out[k] = 0.f; return; }
  const float di = rsqrtf(float(a_rowptr[i + 1] - a_rowptr[i])), dj = rsqrtf(float(a_rowptr[j + 1] - a_rowptr[j]));
  out[k] = gPc[k] * di * dj - 0.5f * di * di * (rs[i] + cs[i]);
}

// gradP[(a, b)] += sum_{c in [c0, c0+cc)} sum_k seeds[m][c][k] * g1bar[c - c0][b][k]  for the first occurrence m of every
// batch node a = idx[m] (the scattered seed planes G_c are non-zero on those rows only and hold the accumulated seeds of
// duplicated node ids).  One wave per sample; lanes over (c, k).
__global__ __launch_bounds__(256) void sddmm_seed_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const int64_t* __restrict__ idx, int64_t M, int64_t N, int64_t C,
                                                         const int32_t* __restrict__ pos, const float* __restrict__ seeds,
                                                         const float* __restrict__ g1bar, int64_t c0, int64_t cc,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t a = idx[m];
  if (a < 0 || a >= N || pos[a] != m) return;  // duplicates: the first occurrence owns the accumulated seed row
  const float* __restrict__ sd = seeds + m * C * C + c0 * C;
  const int64_t nq = cc * C;
  for (int32_t p = rowptr[a]; p < rowptr[a + 1]; ++p) {
    const int64_t b = col[p];
    float acc = 0.f;
    for (int64_t q = lane; q < nq; q += 64) {
      const int64_t c = q / C, k = q - c * C;
      acc += sd[q] * g1bar[(c * N + b) * C + k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[p] += acc;
  }
}

// Vbar[m][c0 + c][k] = sum_b P[a, b] * g1bar[c][b][k], a = idx[m] (every sample, duplicates included)
__global__ __launch_bounds__(256) void seed_adjoint_gather_kernel(const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col,
                                                                  const float* __restrict__ val,
                                                                  const int64_t* __restrict__ idx, int64_t M, int64_t N,
                                                                  int64_t C, const float* __restrict__ g1bar, int64_t c0,
                                                                  int64_t cc, float* __restrict__ vbar) {
  const int64_t nq = cc * C;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= M * nq) return;
  const int64_t m = t / nq, q = t - m * nq;
  const int64_t c = q / C, k = q - c * C;
  const int64_t a = idx[m];
  if (a < 0 || a >= N) return;
  float acc = 0.f;
  for (int32_t p = rowptr[a]; p < rowptr[a + 1]; ++p) acc += val[p] * g1bar[(c * N + col[p]) * C + k];
  vbar[m * C * C + (c0 + c) * C + k] = acc;
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// fbar_m = d/df sum_{k,c} Vbar[k,c] V[k,c](f) + softmax(f) - onehot(y); outbar[idx[m]] += fbar_m.  One wave per sample.
// V[k,c] = alpha_c d_kc - beta_c u_k - gamma_c p_k (alpha = s (1 + t/2), beta = s, gamma = s t / 2, s = sqrt(p),
// t = f - mbar, u = p (1 + t)).  With dp_k/df_m = p_k (d_km - p_m), dmbar/df_m = u_m, ds_c/df_m = s_c (d_cm - p_m) / 2:
//   fbar_m = e1_m + e2_m - p_m sum(e1) - u_m sum(e2) - rho_m (u_m + p_m) + p_m <rho, u> + u_m <rho, p> - sigma_m p_m + p_m <sigma, p>
//   e1 = (Psi_cc alpha - r s - q gamma) / 2, e2 = s (Psi_cc - q) / 2, r_c = sum_k Psi_kc u_k, q_c = sum_k Psi_kc p_k,
//   rho_k = sum_c beta_c Psi_kc, sigma_k = sum_c gamma_c Psi_kc      (Psi = Vbar).
__global__ __launch_bounds__(256) void seed_adjoint_kernel(const float* __restrict__ logits, const float* __restrict__ probs,
                                                           const int64_t* __restrict__ idx, const int64_t* __restrict__ y,
                                                           int64_t M, int64_t N, int64_t C, const float* __restrict__ vbar,
                                                           int fork_exact, float loss_scale, float* __restrict__ outbar) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m = int64_t(blockIdx.x) * 4 + wave;
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;
  float* __restrict__ p_s = sm + size_t(wave) * 8 * C;  // p, u, s, alpha, gamma, rho, sigma, diag
  float* __restrict__ u_s = p_s + C;
  float* __restrict__ s_s = u_s + C;
  float* __restrict__ al_s = s_s + C;
  float* __restrict__ ga_s = al_s + C;
  float* __restrict__ rho_s = ga_s + C;
  float* __restrict__ sig_s = rho_s + C;
  float mb = 0.f;
  for (int64_t k = lane; k < C; k += 64) mb += probs[m * C + k] * logits[n * C + k];
  mb = wsum(mb);
  for (int64_t k = lane; k < C; k += 64) {
    const float p = probs[m * C + k], t = logits[n * C + k] - mb, s = sqrtf(p);
    p_s[k] = p; u_s[k] = p * (1.f + t); s_s[k] = s; al_s[k] = s * (1.f + 0.5f * t); ga_s[k] = 0.5f * s * t;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* __restrict__ vb = vbar + m * C * C;  // vb[c * C + k] = Psi[k, c]
  const int64_t yy = y[m];
  float se1 = 0.f, se2 = 0.f, sru = 0.f, srp = 0.f, ssp = 0.f;
  if (fork_exact) {
    // rho_k = sum_c beta_c Psi_kc, sigma_k = sum_c gamma_c Psi_kc  (lane = k)
    for (int64_t k = lane; k < C; k += 64) {
      float rho = 0.f, sig = 0.f;
      for (int64_t c = 0; c < C; ++c) {
        const float ps = vb[c * C + k];
        rho += s_s[c] * ps;
        sig += ga_s[c] * ps;
      }
      rho_s[k] = rho; sig_s[k] = sig;
      sru += rho * u_s[k]; srp += rho * p_s[k]; ssp += sig * p_s[k];
    }
    sru = wsum(sru); srp = wsum(srp); ssp = wsum(ssp);
  }
  // e1_c, e2_c (lane = c): r_c = sum_k Psi_kc u_k, q_c = sum_k Psi_kc p_k
  float e1_l[4], e2_l[4];  // C <= 256
  int nl = 0;
  for (int64_t c = lane; c < C; c += 64, ++nl) {
    float e1 = 0.f, e2 = 0.f;
    if (fork_exact) {
      float r = 0.f, q = 0.f;
      for (int64_t k = 0; k < C; ++k) {
        const float ps = vb[c * C + k];
        r += ps * u_s[k];
        q += ps * p_s[k];
      }
      const float dg = vb[c * C + c];
      e1 = 0.5f * (dg * al_s[c] - r * s_s[c] - q * ga_s[c]);
      e2 = 0.5f * s_s[c] * (dg - q);
    }
    e1_l[nl] = e1; e2_l[nl] = e2;
    se1 += e1; se2 += e2;
  }
  se1 = wsum(se1); se2 = wsum(se2);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  nl = 0;
  for (int64_t k = lane; k < C; k += 64, ++nl) {
    const float p = p_s[k], u = u_s[k];
    float fb = loss_scale * (p - (k == yy ? 1.f : 0.f));  // d (H_factor * CE) / d f
    if (fork_exact)
      fb += e1_l[nl] + e2_l[nl] - p * se1 - u * se2 - rho_s[k] * (u + p) + p * sru + u * srp - sig_s[k] * p + p * ssp;
    atomicAdd(&outbar[n * C + k], fb);
  }
}

__global__ void relu_mask_inplace_kernel(float* __restrict__ x, const float* __restrict__ h, int64_t n) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride) x[q] = h[q] > 0.f ? x[q] : 0.f;
}

// rs[a] = sum_b gP[a,b] P[a,b], cs[b] += the same (column sums): d J / d d_i = (rs[i] + cs[i]) / d_i
__global__ void gp_rowcol_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                 const float* __restrict__ val, const float* __restrict__ gP, int64_t N,
                                 float* __restrict__ rs, float* __restrict__ cs) {
  const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= N) return;
  float acc = 0.f;
  for (int32_t p = rowptr[a]; p < rowptr[a + 1]; ++p) {
    const float t = gP[p] * val[p];
    acc += t;
    atomicAdd(&cs[col[p]], t);
  }
  rs[a] = acc;
}

__device__ __forceinline__ int32_t find_col(const int32_t* __restrict__ col, int32_t s, int32_t e, int32_t want) {
  while (s < e) {  // sorted columns inside a row
    const int32_t mid = (s + e) >> 1;
    if (col[mid] < want) s = mid + 1;
    else e = mid;
  }
  return s;
}

// Entry p = (i, j) of the stored 0/1 adjacency A (row-major, lgnn_export_adj order):
//   gA[p] = gP[(j, i)] d_i d_j - 1/2 d_i^2 (rs[i] + cs[i]),   d = rowsum(A)^-1/2,   P = D A^T D;   diagonal -> 0
__global__ void adj_grad_kernel(const int32_t* __restrict__ a_rowptr, const int32_t* __restrict__ a_col,
                                const int32_t* __restrict__ p_rowptr, const int32_t* __restrict__ p_col,
                                const float* __restrict__ gP, const float* __restrict__ rs, const float* __restrict__ cs,
                                int64_t N, float* __restrict__ gA) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t s = a_rowptr[i], e = a_rowptr[i + 1];
  if (s == e) return;
  const float di = rsqrtf(float(e - s));
  const float rowterm = -0.5f * di * di * (rs[i] + cs[i]);
  for (int32_t p = s; p < e; ++p) {
    const int32_t j = a_col[p];
    if (j == i) { gA[p] = 0.f; continue; }
    const int32_t js = p_rowptr[j], je = p_rowptr[j + 1];
    const int32_t q = find_col(p_col, js, je, int32_t(i));  // P has the entry (j, i) exactly when A has (i, j)
    const float dj = rsqrtf(float(a_rowptr[j + 1] - a_rowptr[j]));
    gA[p] = gP[q] * di * dj + rowterm;
  }
}

// symmetric models propagate with (adj + adj^T) / 2: the parameter's gradient is the average of (i, j) and (j, i)
__global__ void adj_grad_symmetrize_kernel(const int32_t* __restrict__ a_rowptr, const int32_t* __restrict__ a_col,
                                           const float* __restrict__ gA, int64_t N, float* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int32_t p = a_rowptr[i]; p < a_rowptr[i + 1]; ++p) {
    const int32_t j = a_col[p];
    const int32_t q = find_col(a_col, a_rowptr[j], a_rowptr[j + 1], int32_t(i));
    out[p] = 0.5f * (gA[p] + gA[q]);
  }
}

// GCN top layer, all N rows (kfac.hip's unfused seed SpMM restated here with the planes of a class chunk only)
__global__ void seed_planes_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   const float* __restrict__ val, int64_t N, int64_t C, const int32_t* __restrict__ pos,
                                   const float* __restrict__ seeds, float* __restrict__ g, uint8_t* __restrict__ active) {
  // g[c][n][k] = sum_{v in row n of P^T, v in batch} val * seeds[pos[v]][c][k]; one wave per node
  const int lane = threadIdx.x & 63;
  const int64_t n = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int64_t CC = C * C;
  bool any = false;
  for (int64_t q0 = 0; q0 < CC; q0 += 64) {
    const int64_t q = q0 + lane;
    float acc = 0.f;
    for (int32_t p = rowptr[n]; p < rowptr[n + 1]; ++p) {
      const int32_t mp = pos[col[p]];
      if (mp != INT32_MAX) {
        any = true;
        if (q < CC) acc += val[p] * seeds[int64_t(mp) * CC + q];
      }
    }
    if (q < CC) {
      const int64_t c = q / C, k = q - c * C;
      g[(c * N + n) * C + k] = acc;
    }
  }
  if (lane == 0) active[n] = any ? 1 : 0;
}

// C_rm[R, Nout] = alpha * A_rm[R, K] * B_rm[K, Nout] + beta * C_rm   (row major through the column-major library call)
int sgemm_rm(hipStream_t s, int64_t R, int64_t Nout, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
             int64_t ldb, float beta, float* Cm, int64_t ldc) {
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  LGNN_REQUIRE(R < (int64_t(1) << 31) && Nout < (int64_t(1) << 31) && K < (int64_t(1) << 31), "sgemm: dimension too large");
  const rocblas_status st = rocblas_sgemm(blas, rocblas_operation_none, rocblas_operation_none, rocblas_int(Nout),
                                          rocblas_int(R), rocblas_int(K), &alpha, B, rocblas_int(ldb), A, rocblas_int(lda),
                                          &beta, Cm, rocblas_int(ldc));
  if (st != rocblas_status_success) { set_error("rocblas_sgemm failed"); return 3; }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GraphSAGE
// ---------------------------------------------------------------------------------------------------------------------------
// active[v] = 1 for the batch nodes and every column of their P rows (the rows where dh can be non-zero)
__global__ void sage_mark_active_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, const int32_t* __restrict__ rowptr,
                                        const int32_t* __restrict__ col, uint8_t* __restrict__ active) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t a = idx[m];
  if (a < 0 || a >= N) return;
  if (lane == 0) active[a] = 1;
  for (int32_t p = rowptr[a] + lane; p < rowptr[a + 1]; p += 64) active[col[p]] = 1;
}

__global__ void sage_index_active_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ count,
                                         int32_t* __restrict__ widx) {
  const int64_t w = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w < *count) widx[list[w]] = int32_t(w);
}

// G0[(w, c)] = act'(h_1[v]) * (DCAT[(pos[v], c), :H] + sum_{u in row v of P^T, u in batch} P^T[v,u] DCAT[(pos[u], c), H:]),
// v = list[w], classes c0 <= c < c0 + cc.  One wave per active node, PC classes at a time in registers; H % 4 == 0, <= 256.
template <int PC>
__global__ __launch_bounds__(256) void sage_dh_kernel(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_col,
                                                      const float* __restrict__ t_val, const int32_t* __restrict__ list,
                                                      const int32_t* __restrict__ count, const int32_t* __restrict__ pos,
                                                      const float* __restrict__ dcat, const float* __restrict__ dact,
                                                      int64_t C, int64_t H, int64_t c0, int64_t cc, float* __restrict__ G0) {
  const int lane = threadIdx.x & 63;
  const int k4 = lane * 4;
  const bool col_ok = k4 < H;
  const int64_t total = *count;
  for (int64_t w = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); w < total; w += int64_t(gridDim.x) * 4) {
    const int64_t v = list[w];
    const int32_t mv = pos[v];
    const int32_t s = t_rowptr[v], e = t_rowptr[v + 1];
    float4 dm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok) dm = *reinterpret_cast<const float4*>(dact + v * H + k4);
    for (int64_t pc0 = 0; pc0 < cc; pc0 += PC) {
      float4 acc[PC];
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && pc0 + c < cc && mv != INT32_MAX)
          acc[c] = *reinterpret_cast<const float4*>(dcat + (int64_t(mv) * C + c0 + pc0 + c) * 2 * H + k4);
      }
      for (int32_t p = s; p < e; ++p) {
        const int32_t mu = pos[t_col[p]];
        if (mu == INT32_MAX) continue;  // wave-uniform
        const float a = t_val[p];
#pragma unroll
        for (int c = 0; c < PC; ++c) {
          if (col_ok && pc0 + c < cc) {
            const float4 x = *reinterpret_cast<const float4*>(dcat + (int64_t(mu) * C + c0 + pc0 + c) * 2 * H + H + k4);
            acc[c].x = fmaf(a, x.x, acc[c].x); acc[c].y = fmaf(a, x.y, acc[c].y);
            acc[c].z = fmaf(a, x.z, acc[c].z); acc[c].w = fmaf(a, x.w, acc[c].w);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < PC; ++c)
        if (col_ok && pc0 + c < cc)
          *reinterpret_cast<float4*>(G0 + (w * cc + pc0 + c) * H + k4) =
              make_float4(dm.x * acc[c].x, dm.y * acc[c].y, dm.z * acc[c].z, dm.w * acc[c].w);
    }
  }
}

// For the first occurrence m of every batch node a = idx[m] and the classes of the chunk, in one pass over row a of P:
//     out[(a, b)]        += sum_c <DCAT[(m, c), H:], x_c[b]>,     x_c[b] = act'(h_1[b]) * G0B[(widx[b], c)]
//     DCATB[(m, c), H:]   = sum_b P[a, b] x_c[b]                    DCATB[(m, c), :H] = x_c[a]
// (later occurrences of a node: zero rows -- their seed rows are zero and Vbar is read at the first occurrence.)
template <int PC>
__global__ __launch_bounds__(256) void sage_sddmm_spmm_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const float* __restrict__ val, const int64_t* __restrict__ idx,
                                                              int64_t M, int64_t N, const int32_t* __restrict__ pos,
                                                              const int32_t* __restrict__ widx, const float* __restrict__ dcat,
                                                              const float* __restrict__ G0B, const float* __restrict__ dact,
                                                              int64_t C, int64_t H, int64_t c0, int64_t cc,
                                                              float* __restrict__ dcatb, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int k4 = lane * 4;
  const bool col_ok = k4 < H;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); m < M; m += int64_t(gridDim.x) * 4) {
    const int64_t a = idx[m];
    const bool first = a >= 0 && a < N && pos[a] == m;
    if (!first) {
      for (int64_t c = 0; c < cc; ++c)
        if (col_ok) {
          float* __restrict__ r = dcatb + (m * C + c0 + c) * 2 * H;
          *reinterpret_cast<float4*>(r + k4) = zero4;
          *reinterpret_cast<float4*>(r + H + k4) = zero4;
        }
      continue;
    }
    const int32_t s = rowptr[a], e = rowptr[a + 1];
    const int64_t wa = widx[a];
    float4 da = zero4;
    if (col_ok) da = *reinterpret_cast<const float4*>(dact + a * H + k4);
    for (int64_t pc0 = 0; pc0 < cc; pc0 += PC) {
      float4 D[PC], acc[PC];
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        acc[c] = zero4;
        D[c] = zero4;
        if (col_ok && pc0 + c < cc) D[c] = *reinterpret_cast<const float4*>(dcat + (m * C + c0 + pc0 + c) * 2 * H + H + k4);
      }
      for (int32_t p = s; p < e; ++p) {
        const int64_t b = col[p];
        const int64_t wb = widx[b];
        const float v = val[p];
        float4 db = zero4;
        if (col_ok) db = *reinterpret_cast<const float4*>(dact + b * H + k4);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < PC; ++c) {
          if (col_ok && pc0 + c < cc) {
            float4 x = *reinterpret_cast<const float4*>(G0B + (wb * cc + pc0 + c) * H + k4);
            x.x *= db.x; x.y *= db.y; x.z *= db.z; x.w *= db.w;
            acc[c].x = fmaf(v, x.x, acc[c].x); acc[c].y = fmaf(v, x.y, acc[c].y);
            acc[c].z = fmaf(v, x.z, acc[c].z); acc[c].w = fmaf(v, x.w, acc[c].w);
            d += D[c].x * x.x + D[c].y * x.y + D[c].z * x.z + D[c].w * x.w;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) out[p] += d;  // one wave owns row a (first occurrence): a single writer per entry inside a launch
      }
#pragma unroll
      for (int c = 0; c < PC; ++c)
        if (col_ok && pc0 + c < cc) {
          float* __restrict__ r = dcatb + (m * C + c0 + pc0 + c) * 2 * H;
          const float4 x = *reinterpret_cast<const float4*>(G0B + (wa * cc + pc0 + c) * H + k4);
          *reinterpret_cast<float4*>(r + k4) = make_float4(da.x * x.x, da.y * x.y, da.z * x.z, da.w * x.w);
          *reinterpret_cast<float4*>(r + H + k4) = acc[c];
        }
    }
  }
}

// candidate pairs (a, b): only a in the batch and b active contribute
__global__ __launch_bounds__(256) void sage_cand_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb, int64_t K,
                                                        const int32_t* __restrict__ pos, const int32_t* __restrict__ widx,
                                                        const float* __restrict__ dcat, const float* __restrict__ G0B,
                                                        const float* __restrict__ dact, int64_t C, int64_t H, int64_t c0,
                                                        int64_t cc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t k = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int64_t a = ca[k], b = cb[k];
  const int32_t m = pos[a];
  const int32_t wb = widx[b];
  if (m == INT32_MAX || wb < 0) return;
  float acc = 0.f;
  for (int64_t c = 0; c < cc; ++c) {
    const float* __restrict__ D = dcat + (int64_t(m) * C + c0 + c) * 2 * H + H;
    const float* __restrict__ x = G0B + (int64_t(wb) * cc + c) * H;
    for (int64_t q = lane; q < H; q += 64) acc += D[q] * dact[b * H + q] * x[q];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[k] += acc;
}

// Vbar[m][c][k] = g1bar[(first occurrence of idx[m], c)][k]   (every sample, duplicates included)
__global__ void sage_vbar_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int64_t CC,
                                 const int32_t* __restrict__ pos, const float* __restrict__ g1bar, float* __restrict__ vbar) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= M * CC) return;
  const int64_t m = t / CC, q = t - m * CC;
  const int64_t a = idx[m];
  if (a < 0 || a >= N) return;
  vbar[t] = g1bar[int64_t(pos[a]) * CC + q];
}

// mean_agg backward: rowdot[a] = sum_j gP[a,j] P[a,j];  gA[(a,b)] = (gP[(a,b)] - rowdot[a]) / rowsum_a  (stored rows: rowsum >= 1)
__global__ void sage_adj_grad_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ val,
                                     const float* __restrict__ gP, int64_t N, float* __restrict__ rowdot,
                                     float* __restrict__ gA) {
  const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const int32_t s = rowptr[a], e = rowptr[a + 1];
  float acc = 0.f;
  for (int32_t p = s; p < e; ++p) acc += gP[p] * val[p];
  rowdot[a] = acc;
  for (int32_t p = s; p < e; ++p) gA[p] = (gP[p] - acc) * val[p];
}
// candidate (a, b) = entry (a, b) of A; an empty row's divisor is the constant 1 (layers.py:20) and carries no row term
__global__ void sage_cand_adj_grad_kernel(const int32_t* __restrict__ ca, int64_t K, const int32_t* __restrict__ rowptr,
                                          const float* __restrict__ gPc, const float* __restrict__ rowdot,
                                          float* __restrict__ out) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t a = ca[k];
  const int32_t deg = rowptr[a + 1] - rowptr[a];
  out[k] = deg > 0 ? (gPc[k] - rowdot[a]) / float(deg) : gPc[k];
}

__global__ void relu_mask_ld_kernel(float* __restrict__ x, const float* __restrict__ dact, int64_t n) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride) x[q] *= dact[q];
}

int check_model(const lgnn_ctx* h) {
  LGNN_REQUIRE(h->L == 2, "adjacency gradient: 2-layer models (SURVEY.md 8(f)-4)");
  LGNN_REQUIRE(!h->extras(), "adjacency gradient: models without res / norm");
  LGNN_REQUIRE(h->act == LGNN_ACT_RELU && h->lik == LGNN_LIK_CLASSIFICATION, "adjacency gradient: ReLU, classification");
  LGNN_REQUIRE(h->dims[2] <= 256, "adjacency gradient: at most 256 classes");
  LGNN_REQUIRE(h->kind == LGNN_KIND_GCN || (h->dims[1] % 4 == 0 && h->dims[1] <= 256),
               "adjacency gradient, GraphSAGE: hidden width a multiple of 4, at most 256");
  return 0;
}

int sage_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, bool fork_exact, const float* gamma_B0,
                       const float* gamma_B1, float loss_scale, float* grad_P, float* out_bar, const int32_t* cand_a,
                       const int32_t* cand_b, int64_t K, float* grad_cand, hipStream_t s) {
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], CC = C * C;
  LGNN_CALL(batch_prologue(h, idx, y, M, true, fork_exact, nullptr, s));
  const float* S = h->ws.seeds.as<float>();  // [M][C][C]: row (m, c) = column c of the seed block, zero for later occurrences
  LGNN_REQUIRE(M * C < (int64_t(1) << 31), "adjacency gradient: batch too large");
  // DCAT / DCATB [M C, 2H], g1bar [M C, C], Vbar [M][C][C]
  LGNN_CALL(h->ws.top.reserve(size_t(M) * C * 2 * H * 4 * 2 + size_t(M) * CC * 4 * 2));
  float* dcat = h->ws.top.as<float>();
  float* dcatb = dcat + M * C * 2 * H;
  float* g1b = dcatb + M * C * 2 * H;
  float* vbar = g1b + M * CC;
  LGNN_CALL(sgemm_rm(s, M * C, 2 * H, C, 1.f, S, C, h->W[1], 2 * H, 0.f, dcat, 2 * H));
  // active rows (batch + neighbours), their list and the inverse map
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  LGNN_HIP_CHECK(hipMemsetAsync(h->ws.active.p, 0, size_t(N), s));
  hipLaunchKernelGGL(sage_mark_active_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr, h->P.col,
                     h->ws.active.as<uint8_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
  LGNN_CALL(h->ws.act_count.reserve(64));
  LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                          h->ws.select_tmp, s));
  LGNN_CALL(h->ws.misc.reserve(size_t(N) * 4));
  int32_t* widx = h->ws.misc.as<int32_t>();
  LGNN_HIP_CHECK(hipMemsetAsync(widx, 0xFF, size_t(N) * 4, s));
  hipLaunchKernelGGL(sage_index_active_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->ws.act_list.as<int32_t>(),
                     h->ws.act_count.as<int32_t>(), widx);
  LGNN_HIP_CHECK(hipGetLastError());
  // the number of active rows stays on the device: buffers and grids are sized for the bound N, the kernels and the GEMM over
  // the active rows read the count themselves (no host round trip per batch)
  const int64_t na = N;
  {
    // class chunks: G0 and g0bar [na * cc, H] under the workspace cap
    const int64_t per_class = int64_t(na) * H * 4 * 2;
    const int64_t cc_max = std::max<int64_t>(1, std::min<int64_t>(C, h->ws_limit / std::max<int64_t>(per_class, 1)));
    LGNN_REQUIRE(int64_t(na) * cc_max < (int64_t(1) << 31), "adjacency gradient: too many active rows per chunk");
    LGNN_CALL(h->ws.planes_a.reserve(size_t(na) * cc_max * H * 4));
    LGNN_CALL(h->ws.planes_b.reserve(size_t(na) * cc_max * H * 4));
    h->ws.planes_a_zero_ptr = nullptr;
    float* G0 = h->ws.planes_a.as<float>();
    float* G0B = h->ws.planes_b.as<float>();
    const float* dact = h->fc.dact0.as<float>();
    for (int64_t c0 = 0; c0 < C; c0 += cc_max) {
      const int64_t cc = std::min(cc_max, C - c0);
      hipLaunchKernelGGL(sage_dh_kernel<8>, dim3(unsigned(std::min<int64_t>(cdiv(na, 4), 16384))), dim3(256), 0, s,
                         h->PT.rowptr, h->PT.col, h->PT.val, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                         h->ws.pos.as<int32_t>(), dcat, dact, C, H, c0, cc, G0);
      LGNN_HIP_CHECK(hipGetLastError());
      LGNN_CALL(launch_gemm_devrows(G0, H, gamma_B0, H, G0B, H, na * cc, h->ws.act_count.as<int32_t>(), cc, H, H, 2.f, s));
      if (K > 0)
        hipLaunchKernelGGL(sage_cand_kernel, dim3(unsigned(cdiv(K, 4))), dim3(256), 0, s, cand_a, cand_b, K,
                           h->ws.pos.as<int32_t>(), widx, dcat, G0B, dact, C, H, c0, cc, grad_cand);
      hipLaunchKernelGGL(sage_sddmm_spmm_kernel<8>, dim3(unsigned(std::min<int64_t>(cdiv(M, 4), 16384))), dim3(256), 0, s,
                         h->P.rowptr, h->P.col, h->P.val, idx, M, N, h->ws.pos.as<int32_t>(), widx, dcat, G0B, dact, C, H, c0,
                         cc, dcatb, grad_P);
      LGNN_HIP_CHECK(hipGetLastError());
    }
    // g1bar = dcatbar W1^T + 2 S Gamma_B1     [M C, C]
    LGNN_CALL(sgemm_rm(s, M * C, C, 2 * H, 1.f, dcatb, 2 * H, h->Wt[1].as<float>(), C, 0.f, g1b, C));
    LGNN_CALL(sgemm_rm(s, M * C, C, C, 2.f, S, C, gamma_B1, C, 1.f, g1b, C));
    hipLaunchKernelGGL(sage_vbar_kernel, dim3(unsigned(cdiv(M * CC, 256))), dim3(256), 0, s, idx, M, N, CC,
                       h->ws.pos.as<int32_t>(), g1b, vbar);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_REQUIRE(size_t(4) * 8 * C * 4 <= 64 * 1024, "too many classes for the seed adjoint kernel");
  hipLaunchKernelGGL(seed_adjoint_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), size_t(4) * 8 * C * 4, s,
                     h->fc.out.as<float>(), h->ws.probs.as<float>(), idx, static_cast<const int64_t*>(y), M, N, C, vbar,
                     fork_exact ? 1 : 0, loss_scale, out_bar);
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

// (diagonal posterior: gamma_A0 / gamma_A1 are null; h1_bar [N, H] = direct adjoint of H1, px_bar [N, px_ld] = of P X)
__global__ void add_strided_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t ldy, int64_t rows,
                                   int64_t width);
int sage_adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* gamma_A0, const float* gamma_A1, float a_scale,
                        float* grad_P, float* grad_adj, const int32_t* cand_a, const int32_t* cand_b, int64_t K,
                        float* grad_cand, float* grad_cand_adj, hipStream_t s, const float* h1_bar = nullptr,
                        const float* px_bar = nullptr, int64_t px_ld = 0) {
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], F = h->dims[0];
  LGNN_CALL(forward_ensure_aux(h, s));
  LGNN_CALL(ensure_wt(h, s));
  const float* cat0 = h->fc.lin_in_p[0];  // [N, 2F]: X | P X
  const float* cat1 = h->fc.lin_in_p[1];  // [N, 2H]: H1 | P H1
  LGNN_REQUIRE(h->fc.lin_in_ld[0] == 2 * F && h->fc.lin_in_ld[1] == 2 * H, "internal: unexpected cat row stride");
  LGNN_CALL(h->ws.planes_a.reserve(size_t(N) * (2 * H + H + F) * 4));
  h->ws.planes_a_zero_ptr = nullptr;
  float* T1 = h->ws.planes_a.as<float>();  // [N, 2H]
  float* Z0b = T1 + N * 2 * H;              // [N, H]
  float* XNb = Z0b + N * H;                 // [N, F]
  // T1 = outbar W1 + 2 a cat1 Gamma_A1: the self half feeds H1bar directly, the neighbour half is HNbar
  LGNN_CALL(sgemm_rm(s, N, 2 * H, C, 1.f, out_bar, C, h->W[1], 2 * H, 0.f, T1, 2 * H));
  if (gamma_A1) LGNN_CALL(sgemm_rm(s, N, 2 * H, 2 * H, 2.f * a_scale, cat1, 2 * H, gamma_A1, 2 * H, 1.f, T1, 2 * H));
  if (h1_bar) {  // joins the self half before the mask
    hipLaunchKernelGGL(add_strided_kernel, dim3(unsigned(std::min<int64_t>(cdiv(N * H, 256), 4096))), dim3(256), 0, s, T1, 2 * H,
                       h1_bar, H, N, H);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, T1 + H, 2 * H, 0, cat1, 2 * H, 0, H, 1, grad_P, s));
  LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, T1 + H, 2 * H, 0, cat1, 2 * H, 0, H, 1, nullptr, grad_cand, s));
  // Z0bar = mask * (T1_self + P^T T1_neigh)
  SpmmArgs sa{};
  sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
  sa.in = T1 + H; sa.in_ld = 2 * H; sa.in_plane_stride = 0;
  sa.self = T1; sa.self_ld = 2 * H; sa.self_plane_stride = 0;
  sa.out = Z0b; sa.out_ld = H; sa.out_plane_stride = 0; sa.width = H; sa.out_act = -1;
  LGNN_CALL(launch_spmm_ex(sa, 1, s));
  hipLaunchKernelGGL(relu_mask_ld_kernel, dim3(unsigned(std::min<int64_t>(cdiv(N * H, 256), 4096))), dim3(256), 0, s, Z0b,
                     h->fc.dact0.as<float>(), N * H);
  LGNN_HIP_CHECK(hipGetLastError());
  // XNbar = Z0bar W0[:, F:] + 2 a (cat0 Gamma_A0)[:, F:]
  LGNN_CALL(sgemm_rm(s, N, F, H, 1.f, Z0b, H, h->W[0] + F, 2 * F, 0.f, XNb, F));
  if (gamma_A0) LGNN_CALL(sgemm_rm(s, N, F, 2 * F, 2.f * a_scale, cat0, 2 * F, gamma_A0 + F, 2 * F, 1.f, XNb, F));
  if (px_bar) {
    hipLaunchKernelGGL(add_strided_kernel, dim3(unsigned(std::min<int64_t>(cdiv(N * F, 256), 4096))), dim3(256), 0, s, XNb, F, px_bar,
                       px_ld, N, F);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, XNb, F, 0, cat0, 2 * F, 0, F, 1, grad_P, s));
  LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, XNb, F, 0, cat0, 2 * F, 0, F, 1, nullptr, grad_cand, s));
  // mean_agg backward + the straight-through binarisation (P has the pattern and the row order of A)
  LGNN_CALL(h->ws.misc.reserve(size_t(N) * 4 + size_t(std::max<int64_t>(h->nnz, 1)) * 4));
  float* rowdot = h->ws.misc.as<float>();
  float* tmp = rowdot + N;
  float* first = h->sym ? tmp : grad_adj;
  hipLaunchKernelGGL(sage_adj_grad_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->P.rowptr, h->P.val, grad_P, N,
                     rowdot, first);
  if (h->sym)
    hipLaunchKernelGGL(adj_grad_symmetrize_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->P.rowptr, h->P.col, tmp, N,
                       grad_adj);
  if (K > 0)
    hipLaunchKernelGGL(sage_cand_adj_grad_kernel, dim3(unsigned(cdiv(K, 256))), dim3(256), 0, s, cand_a, K, h->P.rowptr, grad_cand,
                       rowdot, grad_cand_adj);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------------
// Models with res / norm (gnn/models/base_gnn.py:141-149; the STE-GCN configurations of Cornell / Texas / Wisconsin / Circle,
// gnn/configs/original/stegcn_config.yaml:54-105, 129-145):  s = P Z0 + res(X),  n = norm(s),  H1 = relu(n).
// The norm's row-local backward y = rstd Pi(xhat) x (Pi x = x - mean(x) - xhat mean(x xhat); symmetric) appears in two places
// of the second-order sweep, each time with an adjoint ybar of y:
//   KRON  u = norm_bwd(dn):        x = gamma * dn,  ybar = ubar;         needed: dn_bar = mask * gamma * (rstd Pi ybar)
//   !KRON ndot = norm_tangent(sdot): x = sdot,      ybar = gamma * nbar;   needed: sdot_bar = rstd Pi ybar
// and, both times, the adjoint of the pre-norm rows s through the norm's own statistics (x held fixed):
//   rstd_bar = <ybar, y> / rstd,   xhat_bar_j = -rstd (ybar_j mean(x xhat) + mean(ybar xhat) x_j),
//   s_bar    = rstd Pi xhat_bar - rstd_bar rstd^2 xhat / W          (d rstd / d s_j = -rstd^2 xhat_j / W).
// In place: YB <- the first result, X <- s_bar (LayerNorm only; untouched otherwise).  One wave per plane row, node = row % N.
// BatchNorm1d in eval mode is a fixed affine map (no second-order term), Identity passes through.
template <bool KRON>
__global__ __launch_bounds__(256) void norm_pair_kernel(float* __restrict__ X, float* __restrict__ YB, int64_t rows, int64_t N,
                                                        int64_t W, int norm, const float* __restrict__ gamma,
                                                        const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                        const float* __restrict__ hact, int64_t hact_ld, int act) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = int64_t(gridDim.x) * 4;
  for (int64_t r = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); r < rows; r += stride) {
    const int64_t n = r % N;
    float* __restrict__ x = X + r * W;
    float* __restrict__ yb = YB + r * W;
    const float* __restrict__ hn = (KRON && hact) ? hact + n * hact_ld : nullptr;
    if (norm != LGNN_NORM_LAYER) {
      for (int64_t j = lane; j < W; j += 64) {
        float t = yb[j];
        if (norm == LGNN_NORM_BATCH) t *= gamma[j] * rstd[j];
        if (hn) t *= act_deriv_from_out(hn[j], act);
        yb[j] = t;
      }
      continue;
    }
    const float* __restrict__ xh = xhat + n * W;
    float sx = 0.f, sxx = 0.f, sy = 0.f, syx = 0.f, sxy = 0.f;
    for (int64_t j = lane; j < W; j += 64) {
      const float g = gamma[j], h = xh[j];
      const float xv = KRON ? x[j] * g : x[j];
      const float yv = KRON ? yb[j] : yb[j] * g;
      sx += xv; sxx += xv * h; sy += yv; syx += yv * h; sxy += xv * yv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sx += __shfl_xor(sx, o); sxx += __shfl_xor(sxx, o); sy += __shfl_xor(sy, o); syx += __shfl_xor(syx, o);
      sxy += __shfl_xor(sxy, o);
    }
    const float iw = 1.f / float(W);
    const float mx1 = sx * iw, mx = sxx * iw, my1 = sy * iw, my = syx * iw;
    const float rs = rstd[n];
    const float rstd_bar = sxy - float(W) * (my1 * mx1 + my * mx);
    const float m_xb = -rs * (my1 * mx + my * mx1);  // mean(xhat_bar)
    const float m_xbx = -2.f * rs * my * mx;          // mean(xhat_bar * xhat)
    for (int64_t j = lane; j < W; j += 64) {
      const float g = gamma[j], h = xh[j];
      const float xv = KRON ? x[j] * g : x[j];
      const float yv = KRON ? yb[j] : yb[j] * g;
      const float xb = -rs * (yv * mx + my * xv);
      x[j] = rs * (xb - m_xb - h * m_xbx) - rstd_bar * rs * rs * h * iw;
      float o1 = rs * (yv - my1 - h * my);
      if (KRON) {
        o1 *= g;
        if (hn) o1 *= act_deriv_from_out(hn[j], act);
      }
      yb[j] = o1;
    }
  }
}

// ndot = mask * norm_tangent(sdot): out[r] = act'(h[n]) * gamma * rstd * Pi sdot[r] (LayerNorm), * gamma * rstd_channel
// (BatchNorm), sdot itself (Identity); one wave per plane row
__global__ __launch_bounds__(256) void norm_tangent_kernel(const float* __restrict__ S, float* __restrict__ out, int64_t rows,
                                                           int64_t N, int64_t W, int norm, const float* __restrict__ gamma,
                                                           const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                           const float* __restrict__ hact, int64_t hact_ld, int act) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = int64_t(gridDim.x) * 4;
  for (int64_t r = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); r < rows; r += stride) {
    const int64_t n = r % N;
    const float* __restrict__ sd = S + r * W;
    const float* __restrict__ hn = hact + n * hact_ld;
    float m1 = 0.f, m2 = 0.f, rs = 1.f;
    if (norm == LGNN_NORM_LAYER) {
      const float* __restrict__ xh = xhat + n * W;
      for (int64_t j = lane; j < W; j += 64) { m1 += sd[j]; m2 += sd[j] * xh[j]; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
      m1 /= float(W); m2 /= float(W);
      rs = rstd[n];
    }
    for (int64_t j = lane; j < W; j += 64) {
      float t = sd[j];
      if (norm == LGNN_NORM_LAYER) t = gamma[j] * rs * (t - m1 - xhat[n * W + j] * m2);
      else if (norm == LGNN_NORM_BATCH) t *= gamma[j] * rstd[j];
      out[r * W + j] = t * act_deriv_from_out(hn[j], act);
    }
  }
}

// Y[q][r][:] += bias_q[:]  (bias_q = base + q * bias_stride); planes [Q][rows][W]
__global__ void plane_bias_kernel(float* __restrict__ Y, int64_t rows, int64_t W, const float* __restrict__ base,
                                  int64_t bias_stride, int64_t total) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t q = t / (rows * W), j = t % W;
    Y[t] += base[q * bias_stride + j];
  }
}

// Diagonal posterior: K_m = J_m diag(gamma) J_m^T [C, C] of every sample of the chunk; one workgroup per (sample, a <= b)
__global__ __launch_bounds__(256) void diag_ext_gram_kernel(int64_t C, int64_t P, const float* __restrict__ J,
                                                            const float* __restrict__ gamma, float* __restrict__ Kout) {
  __shared__ float red[4];
  const int64_t m = blockIdx.x;
  // pair index -> (a, b), a <= b
  int64_t a = 0, rest = blockIdx.y;
  while (rest >= C - a) { rest -= C - a; ++a; }
  const int64_t b = a + rest;
  const float* __restrict__ Ja = J + (m * C + a) * P;
  const float* __restrict__ Jb = J + (m * C + b) * P;
  float acc = 0.f;
  for (int64_t p = threadIdx.x; p < P; p += 256) acc += Ja[p] * gamma[p] * Jb[p];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = red[0] + red[1] + red[2] + red[3];
    Kout[(m * C + a) * C + b] = v;
    Kout[(m * C + b) * C + a] = v;
  }
}

// per sample m of the chunk (one wave): p = softmax(f_n);  out_bar[n] += Lambda (diag K - 2 K p) + loss_scale (p - onehot(y));
// probs[m] = p (kept for the direction kernel)
__global__ __launch_bounds__(64) void diag_ext_sample_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ y,
                                                             int64_t N, int64_t C, const float* __restrict__ Kin,
                                                             const float* __restrict__ logits, float loss_scale,
                                                             float* __restrict__ probs, float* __restrict__ out_bar) {
  extern __shared__ float sh[];  // K [C*C] | p [C]
  float* Ks = sh;
  float* ps = sh + C * C;
  const int64_t m = blockIdx.x;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;
  for (int64_t q = threadIdx.x; q < C * C; q += 64) Ks[q] = Kin[m * C * C + q];
  if (threadIdx.x == 0) {
    float mx = -INFINITY, sum = 0.f;
    for (int64_t c = 0; c < C; ++c) mx = fmaxf(mx, logits[n * C + c]);
    for (int64_t c = 0; c < C; ++c) { ps[c] = expf(logits[n * C + c] - mx); sum += ps[c]; }
    for (int64_t c = 0; c < C; ++c) { ps[c] /= sum; probs[m * C + c] = ps[c]; }
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < C; c += 64) {
    // t = diag K - 2 K p;  (Lambda t)_c = p_c (t_c - <p, t>)
    float pt = 0.f, tc = 0.f;
    for (int64_t a = 0; a < C; ++a) {
      float kp = 0.f;
      for (int64_t b = 0; b < C; ++b) kp += Ks[a * C + b] * ps[b];
      const float ta = Ks[a * C + a] - 2.f * kp;
      pt += ps[a] * ta;
      if (a == c) tc = ta;
    }
    const float v = ps[c] * (tc - pt) + loss_scale * (ps[c] - (y[m] == c ? 1.f : 0.f));
    atomicAdd(&out_bar[n * C + c], v);
  }
}

// R[m][c][p] = 2 gamma_p (Lambda_m J_m)[c, p] = 2 gamma_p p_c (J[m][c][p] - sum_k p_k J[m][k][p])
__global__ void diag_ext_direction_kernel(const float* __restrict__ J, const float* __restrict__ gamma,
                                          const float* __restrict__ probs, int64_t mc, int64_t C, int64_t P,
                                          float* __restrict__ R) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < mc * P; t += stride) {
    const int64_t m = t / P, p = t - m * P;
    const float* __restrict__ Jm = J + m * C * P + p;
    const float* __restrict__ pm = probs + m * C;
    float jb = 0.f;
    for (int64_t k = 0; k < C; ++k) jb += pm[k] * Jm[k * P];
    const float g2 = 2.f * gamma[p];
    for (int64_t c = 0; c < C; ++c) R[(m * C + c) * P + p] = g2 * pm[c] * (Jm[c * P] - jb);
  }
}

// plane q = (m, c) of the chunk, one wave: the directional derivative is qv = sum_b P[n, b] z1dot_q[b, c], n = idx[m]:
//   gradP[(n, b)] += z1dot_q[b, c];   nbar_q[b, :] = act'(h[b]) P[n, b] W1[c, :]  (rows outside the row of n stay 0);
//   h1_bar[b, :] += P[n, b] dW1_q[c, :]
__global__ __launch_bounds__(256) void diag_ext_row_kernel(const int64_t* __restrict__ idx, int64_t planes, int64_t N, int64_t C,
                                                           int64_t H, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, const float* __restrict__ val,
                                                           const float* __restrict__ Z1d, const float* __restrict__ W1,
                                                           const float* __restrict__ dW1, int64_t dir_stride,
                                                           const float* __restrict__ hact, int64_t hact_ld, int act,
                                                           float* __restrict__ nbar, float* __restrict__ h1_bar,
                                                           float* __restrict__ grad_P) {
  const int lane = threadIdx.x & 63;
  const int64_t q = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (q >= planes) return;
  const int64_t m = q / C, c = q - m * C;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;
  const float* __restrict__ w = W1 + c * H;
  const float* __restrict__ dw = dW1 + q * dir_stride + c * H;
  for (int32_t p = rowptr[n]; p < rowptr[n + 1]; ++p) {
    const int64_t b = col[p];
    const float pv = val[p];
    if (lane == 0) atomicAdd(&grad_P[p], Z1d[(q * N + b) * C + c]);
    for (int64_t j = lane; j < H; j += 64) {
      nbar[(q * N + b) * H + j] = act_deriv_from_out(hact[b * hact_ld + j], act) * pv * w[j];
      atomicAdd(&h1_bar[b * H + j], pv * dw[j]);
    }
  }
}

// candidate pairs (a, b) of P: grad_cand[k] += z1dot_q[b, c] for every plane q = (m, c) with idx[m] == a
__global__ void diag_ext_cand_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb, int64_t K,
                                     const int64_t* __restrict__ idx, int64_t planes, int64_t N, int64_t C,
                                     const float* __restrict__ Z1d, float* __restrict__ grad_cand) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= K * planes) return;
  const int64_t k = t / planes, q = t - k * planes;
  const int64_t m = q / C, c = q - m * C;
  if (idx[m] != ca[k]) return;
  atomicAdd(&grad_cand[k], Z1d[(q * N + cb[k]) * C + c]);
}

// row-major C_q[R, Nout] = alpha * A[R, K] (shared by all planes) * B_q[Nout, K]^T + beta * C_q,  B_q = B + q * strideB
int sgemm_rm_nt_shared(hipStream_t s, int64_t R, int64_t Nout, int64_t K, const float* A, int64_t lda, const float* B,
                       int64_t strideB, float beta, float* Cm, int64_t strideC, int64_t batch) {
  rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
  LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
  const float one = 1.f;
  const rocblas_status st = rocblas_sgemm_strided_batched(
      blas, rocblas_operation_transpose, rocblas_operation_none, rocblas_int(Nout), rocblas_int(R), rocblas_int(K), &one, B,
      rocblas_int(K), rocblas_stride(strideB), A, rocblas_int(lda), 0, &beta, Cm, rocblas_int(Nout), rocblas_stride(strideC),
      rocblas_int(batch));
  if (st != rocblas_status_success) { set_error("rocblas_sgemm_strided_batched failed"); return 3; }
  return 0;
}

int check_model_ext(const lgnn_ctx* h) {
  LGNN_REQUIRE(h->L == 2, "adjacency gradient: 2-layer models (SURVEY.md 8(f)-4)");
  LGNN_REQUIRE(h->kind == LGNN_KIND_GCN, "adjacency gradient with res / norm: GCN models (the reference's STEGCN configurations)");
  LGNN_REQUIRE(h->act == LGNN_ACT_RELU && h->lik == LGNN_LIK_CLASSIFICATION, "adjacency gradient: ReLU, classification");
  LGNN_REQUIRE(h->dims[2] <= 256, "adjacency gradient: at most 256 classes");
  return 0;
}

int launch_norm_pair(lgnn_ctx* h, bool kron, float* X, float* YB, int64_t rows, hipStream_t s) {
  const int64_t N = h->N, H = h->dims[1];
  const bool nrm = h->norm != LGNN_NORM_NONE;
  const float* g = nrm ? h->norm_w[0] : nullptr;
  const float* xh = nrm ? h->fc.xhat[0].as<float>() : nullptr;
  const float* rs = nrm ? h->fc.rstd[0].as<float>() : nullptr;
  const unsigned grid = unsigned(std::min<int64_t>(cdiv(rows, 4), 65536));
  if (kron)
    hipLaunchKernelGGL(norm_pair_kernel<true>, dim3(grid), dim3(256), 0, s, X, YB, rows, N, H, h->norm, g, xh, rs,
                       h->fc.hact_p[0], h->fc.hact_ld[0], h->act);
  else
    hipLaunchKernelGGL(norm_pair_kernel<false>, dim3(grid), dim3(256), 0, s, X, YB, rows, N, H, h->norm, g, xh, rs,
                       static_cast<const float*>(nullptr), int64_t(0), h->act);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// Z0 = X W0^T + b0 [N, H] (the forward keeps only P Z0 + ...): what every adjoint of s = P Z0 + res(X) pairs with
int ext_z0(lgnn_ctx* h, hipStream_t s) {
  const int64_t N = h->N, H = h->dims[1], F = h->dims[0];
  LGNN_CALL(h->ws.adj_z0.reserve(size_t(N) * H * 4));
  GemmEpilogue eb0;
  eb0.bias = h->b[0];
  LGNN_CALL(launch_gemm(h->fc.lin_in_p[0], h->fc.lin_in_ld[0], h->Wt[0].as<float>(), H, h->ws.adj_z0.as<float>(), H, N, F, H,
                        eb0, s));
  return 0;
}

// Kronecker posterior, GCN with res / norm: kfac_adjgrad_batch's chain with the norm's row-local backward between the mask and
// P^T, the res block's B (gamma_Br; u = the gradient at s) and the adjoint of s through the LayerNorm statistics.  All N rows,
// unfused kernels (the configurations that use it are graphs of a few hundred to a few thousand nodes).
int kfac_adjgrad_batch_ext(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, bool fork_exact, const float* gamma_B0,
                           const float* gamma_B1, const float* gamma_Br, float loss_scale, float* grad_P, float* out_bar,
                           const int32_t* cand_a, const int32_t* cand_b, int64_t K, float* grad_cand, hipStream_t s) {
  LGNN_CALL(check_model_ext(h));
  LGNN_REQUIRE(!h->has_res || gamma_Br, "res=True: gamma_B needs a third entry (the res.0 block)");
  LGNN_CALL(forward_ensure(h, s));
  LGNN_CALL(ensure_wt(h, s));
  LGNN_CALL(forward_input_view(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], CC = C * C;
  const bool ln = h->norm == LGNN_NORM_LAYER;
  LGNN_CALL(batch_prologue(h, idx, y, M, true, fork_exact, nullptr, s));
  LGNN_CALL(ext_z0(h, s));
  const float* Z0 = h->ws.adj_z0.as<float>();
  LGNN_CALL(h->ws.top.reserve(size_t(N) * CC * 4 + 16));
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  float* g1 = h->ws.top.as<float>();
  hipLaunchKernelGGL(seed_planes_kernel, dim3(unsigned(cdiv(N, 4))), dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N, C,
                     h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), g1, h->ws.active.as<uint8_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(h->ws.jac.reserve(size_t(M) * CC * 4));  // Vbar [M][C][C]
  float* vbar = h->ws.jac.as<float>();
  const int64_t per_class = N * (4 * H + C) * 4;
  const int64_t cc_max = std::max<int64_t>(1, std::min<int64_t>(C, h->ws_limit / std::max<int64_t>(per_class, 1)));
  LGNN_CALL(h->ws.planes_a.reserve(size_t(cc_max) * N * 2 * H * 4));
  LGNN_CALL(h->ws.planes_b.reserve(size_t(cc_max) * N * (2 * H + C) * 4));
  h->ws.planes_a_zero_ptr = nullptr;
  for (int64_t c0 = 0; c0 < C; c0 += cc_max) {
    const int64_t cc = std::min(cc_max, C - c0);
    float* DN = h->ws.planes_a.as<float>();   // mask * (g1 W1); later the adjoint of s through the norm's statistics
    float* U = DN + cc_max * N * H;           // u = norm_bwd(dn)
    float* UB = h->ws.planes_b.as<float>();   // g0, then ubar, then dn_bar
    float* G0B = UB + cc_max * N * H;
    float* G1B = G0B + cc_max * N * H;
    const float* g1c = g1 + c0 * N * C;
    GemmEpilogue ep;
    ep.hact = h->fc.hact_p[0]; ep.hact_ld = h->fc.hact_ld[0]; ep.act = h->act; ep.hact_row_mod = N;
    LGNN_CALL(launch_gemm(g1c, C, h->W[1], H, DN, H, cc * N, C, H, ep, s));
    LGNN_HIP_CHECK(hipMemcpyAsync(U, DN, size_t(cc) * N * H * 4, hipMemcpyDeviceToDevice, s));
    if (h->norm != LGNN_NORM_NONE) LGNN_CALL(launch_resnorm_backward(h, 0, U, H, cc * N, nullptr, false, s));
    SpmmArgs sa{};
    sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = h->PT.val; sa.nrows = N;
    sa.in = U; sa.in_ld = H; sa.in_plane_stride = N * H; sa.out = UB; sa.out_ld = H; sa.out_plane_stride = N * H;
    sa.width = H; sa.out_act = -1;
    LGNN_CALL(launch_spmm_ex(sa, cc, s));                                          // g0 = P^T u
    LGNN_CALL(sgemm_rm(s, cc * N, H, H, 2.f, UB, H, gamma_B0, H, 0.f, G0B, H));    // g0bar = 2 g0 Gamma_B0
    LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, U, H, N * H, G0B, H, N * H, H, cc, grad_P, s));
    LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, U, H, N * H, G0B, H, N * H, H, cc, nullptr, grad_cand, s));
    SpmmArgs sb{};
    sb.rowptr = h->P.rowptr; sb.col = h->P.col; sb.val = h->P.val; sb.nrows = N;
    sb.in = G0B; sb.in_ld = H; sb.in_plane_stride = N * H; sb.out = UB; sb.out_ld = H; sb.out_plane_stride = N * H;
    sb.width = H; sb.out_act = -1;
    LGNN_CALL(launch_spmm_ex(sb, cc, s));                                          // ubar = P g0bar
    if (h->has_res) LGNN_CALL(sgemm_rm(s, cc * N, H, H, 2.f, U, H, gamma_Br, H, 1.f, UB, H));  // + 2 u Gamma_Br
    LGNN_CALL(launch_norm_pair(h, true, DN, UB, cc * N, s));                       // UB = dn_bar, DN = s_bar (LayerNorm)
    if (ln) {
      LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, DN, H, N * H, Z0, H, 0, H, cc, grad_P, s));
      LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, DN, H, N * H, Z0, H, 0, H, cc, nullptr, grad_cand, s));
    }
    // g1bar = dn_bar W1^T + 2 g1 Gamma_B1     [cc * N, C]
    LGNN_CALL(sgemm_rm(s, cc * N, C, H, 1.f, UB, H, h->Wt[1].as<float>(), C, 0.f, G1B, C));
    LGNN_CALL(sgemm_rm(s, cc * N, C, C, 2.f, g1c, C, gamma_B1, C, 1.f, G1B, C));
    hipLaunchKernelGGL(sddmm_seed_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, h->P.rowptr, h->P.col, idx, M, N, C,
                       h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), G1B, c0, cc, grad_P);
    LGNN_HIP_CHECK(hipGetLastError());
    if (K > 0)
      hipLaunchKernelGGL(sddmm_coo_seed_kernel, dim3(unsigned(cdiv(K, 4))), dim3(256), 0, s, cand_a, cand_b, K, N, C,
                         h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), G1B, c0, cc, grad_cand);
    hipLaunchKernelGGL(seed_adjoint_gather_kernel, dim3(unsigned(cdiv(M * cc * C, 256))), dim3(256), 0, s, h->P.rowptr,
                       h->P.col, h->P.val, idx, M, N, C, G1B, c0, cc, vbar);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_REQUIRE(size_t(4) * 8 * C * 4 <= 64 * 1024, "too many classes for the seed adjoint kernel");
  hipLaunchKernelGGL(seed_adjoint_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), size_t(4) * 8 * C * 4, s,
                     h->fc.out.as<float>(), h->ws.probs.as<float>(), idx, static_cast<const int64_t*>(y), M, N, C, vbar,
                     fork_exact ? 1 : 0, loss_scale, out_bar);
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

// Diagonal posterior, GCN with res / norm (no closed form of the diagonal GGN with a LayerNorm between the layers):
//     d sum_p gamma_p H_p = sum_n <K_n, d Lambda_n> + sum_{n,c} <R_n[c, :], d J_n[c, :]>,   K_n = J_n diag(gamma) J_n^T,
//     R_n = 2 Lambda_n J_n diag(gamma)
// (laplace/curvature/curvature.py:412-432 with the fork's attached Jacobians, :89-130).  <R, d grad_theta f_{n,c}> = the
// derivative of the directional derivative of f_{n,c} along the parameter direction R: per (sample, class) plane one tangent
// forward pass with the weights replaced by R's blocks, then its reverse pass.  Chunks of samples under the workspace cap.
int diag_adjgrad_batch_ext(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, const float* gamma, float loss_scale,
                           float* grad_P, float* out_bar, float* h1_bar, const int32_t* cand_a, const int32_t* cand_b,
                           int64_t K, float* grad_cand, hipStream_t s) {
  LGNN_CALL(check_model_ext(h));
  LGNN_REQUIRE(M > 0 && idx && y && gamma && grad_P && out_bar && h1_bar, "empty batch or null pointers");
  LGNN_REQUIRE(K == 0 || (cand_a && cand_b && grad_cand), "candidate pairs without their buffers");
  LGNN_CALL(forward_ensure(h, s));
  LGNN_CALL(ensure_wt(h, s));
  LGNN_CALL(forward_input_view(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], F = h->dims[0], P = h->n_params;
  const bool ln = h->norm == LGNN_NORM_LAYER, nrm = h->norm != LGNN_NORM_NONE;
  const int64_t oW0 = 0, ob0 = H * F, oW1 = ob0 + H, ob1 = oW1 + C * H, oR0 = ob1 + C, or0 = oR0 + H * F;
  LGNN_REQUIRE(P == (h->has_res ? or0 + H : oR0), "internal: parameter count");
  LGNN_REQUIRE(size_t(C * C + C + 4) * 4 <= 64 * 1024, "too many classes");
  LGNN_CALL(ext_z0(h, s));
  const float* Z0 = h->ws.adj_z0.as<float>();
  const float* X = h->fc.lin_in_p[0];
  const int64_t ldx = h->fc.lin_in_ld[0];
  // per sample: J and R rows, three [C][N][H] plane sets + [C][N][C], and the Jacobian pass's own two plane sets
  const int64_t per_sample = C * (2 * P + N * (3 * H + C) + 2 * N * std::max(H, C)) * 4;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(M, h->ws_limit / std::max<int64_t>(per_sample, 1)));
  LGNN_REQUIRE(chunk * C < (int64_t(1) << 31), "diag adjacency gradient: chunk too large");
  LGNN_CALL(h->ws.jac.reserve(size_t(chunk) * C * P * 4));
  LGNN_CALL(h->ws.adj_dir.reserve(size_t(chunk) * C * P * 4));
  LGNN_CALL(h->ws.probs.reserve(size_t(chunk) * (C + C * C) * 4));
  LGNN_CALL(h->ws.planes_c.reserve(size_t(chunk) * C * N * (3 * H + C) * 4));
  float* J = h->ws.jac.as<float>();
  float* R = h->ws.adj_dir.as<float>();
  float* probs = h->ws.probs.as<float>();
  float* Kn = probs + chunk * C;  // K_n [chunk][C][C]
  const int64_t* yy = static_cast<const int64_t*>(y);
  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t mc = std::min(chunk, M - m0);
    const int64_t Q = mc * C;
    LGNN_CALL(jacobians(h, idx + m0, mc, J, nullptr, s));
    hipLaunchKernelGGL(diag_ext_gram_kernel, dim3(unsigned(mc), unsigned(C * (C + 1) / 2)), dim3(256), 0, s, C, P, J, gamma, Kn);
    hipLaunchKernelGGL(diag_ext_sample_kernel, dim3(unsigned(mc)), dim3(64), size_t(C * C + C) * 4, s, idx + m0, yy + m0, N, C,
                       Kn, h->fc.out.as<float>(), loss_scale, probs, out_bar);
    hipLaunchKernelGGL(diag_ext_direction_kernel, dim3(unsigned(std::min<int64_t>(cdiv(mc * P, 256), 8192))), dim3(256), 0, s, J,
                       gamma, probs, mc, C, P, R);
    LGNN_HIP_CHECK(hipGetLastError());
    float* Z0d = h->ws.planes_c.as<float>();  // [Q][N][H] tangent of Z0
    float* Sd = Z0d + Q * N * H;              // tangent of s; later the adjoint of s through the norm's statistics
    float* Hd = Sd + Q * N * H;               // tangent of H1; later nbar, then the adjoint of sdot
    float* Z1d = Hd + Q * N * H;              // [Q][N][C] tangent of Z1
    const unsigned gq = unsigned(std::min<int64_t>(cdiv(Q * N * H, 256), 16384));
    // tangent forward: Z0dot = X dW0^T + db0;  sdot = P Z0dot (+ X dR0^T + dr0);  H1dot = mask * norm_tangent(sdot);
    //                  Z1dot = H1dot W1^T + H1 dW1^T + db1
    LGNN_CALL(sgemm_rm_nt_shared(s, N, H, F, X, ldx, R + oW0, P, 0.f, Z0d, N * H, Q));
    hipLaunchKernelGGL(plane_bias_kernel, dim3(gq), dim3(256), 0, s, Z0d, N, H, R + ob0, P, Q * N * H);
    LGNN_HIP_CHECK(hipGetLastError());
    SpmmArgs sa{};
    sa.rowptr = h->P.rowptr; sa.col = h->P.col; sa.val = h->P.val; sa.nrows = N;
    sa.in = Z0d; sa.in_ld = H; sa.in_plane_stride = N * H; sa.out = Sd; sa.out_ld = H; sa.out_plane_stride = N * H;
    sa.width = H; sa.out_act = -1;
    LGNN_CALL(launch_spmm_ex(sa, Q, s));
    if (h->has_res) {
      LGNN_CALL(sgemm_rm_nt_shared(s, N, H, F, X, ldx, R + oR0, P, 1.f, Sd, N * H, Q));
      hipLaunchKernelGGL(plane_bias_kernel, dim3(gq), dim3(256), 0, s, Sd, N, H, R + or0, P, Q * N * H);
      LGNN_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(norm_tangent_kernel, dim3(unsigned(std::min<int64_t>(cdiv(Q * N, 4), 65536))), dim3(256), 0, s, Sd, Hd,
                       Q * N, N, H, h->norm, nrm ? h->norm_w[0] : static_cast<const float*>(nullptr),
                       nrm ? h->fc.xhat[0].as<float>() : static_cast<const float*>(nullptr),
                       nrm ? h->fc.rstd[0].as<float>() : static_cast<const float*>(nullptr), h->fc.hact_p[0], h->fc.hact_ld[0],
                       h->act);
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(sgemm_rm(s, Q * N, C, H, 1.f, Hd, H, h->Wt[1].as<float>(), C, 0.f, Z1d, C));
    LGNN_CALL(sgemm_rm_nt_shared(s, N, C, H, h->fc.hact_p[0], h->fc.hact_ld[0], R + oW1, P, 1.f, Z1d, N * C, Q));
    hipLaunchKernelGGL(plane_bias_kernel, dim3(unsigned(std::min<int64_t>(cdiv(Q * N * C, 256), 16384))), dim3(256), 0, s, Z1d, N,
                       C, R + ob1, P, Q * N * C);
    LGNN_HIP_CHECK(hipGetLastError());
    // reverse: the row of n_q
    LGNN_HIP_CHECK(hipMemsetAsync(Hd, 0, size_t(Q) * N * H * 4, s));
    hipLaunchKernelGGL(diag_ext_row_kernel, dim3(unsigned(cdiv(Q, 4))), dim3(256), 0, s, idx + m0, Q, N, C, H, h->P.rowptr,
                       h->P.col, h->P.val, Z1d, h->W[1], R + oW1, P, h->fc.hact_p[0], h->fc.hact_ld[0], h->act, Hd, h1_bar,
                       grad_P);
    if (K > 0)
      hipLaunchKernelGGL(diag_ext_cand_kernel, dim3(unsigned(cdiv(K * Q, 256))), dim3(256), 0, s, cand_a, cand_b, K, idx + m0, Q,
                         N, C, Z1d, grad_cand);
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(launch_norm_pair(h, false, Sd, Hd, Q * N, s));  // Hd = adjoint of sdot, Sd = adjoint of s (LayerNorm)
    LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, Hd, H, N * H, Z0d, H, N * H, H, Q, grad_P, s));
    LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, Hd, H, N * H, Z0d, H, N * H, H, Q, nullptr, grad_cand, s));
    if (ln) {
      LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, Sd, H, N * H, Z0, H, 0, H, Q, grad_P, s));
      LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, Sd, H, N * H, Z0, H, 0, H, Q, nullptr, grad_cand, s));
    }
  }
  h->ws.planes_a_zero_ptr = nullptr;
  return 0;
}
}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------------
// Diagonal posterior, 2-layer GraphSAGE (STEGraphSAGE + DiagLaplace: the driver offers the pair, gnn/utils.py:55-59, 81):
//     s = [X | P X] W0^T + b0,  H1 = relu(s),  out = [H1 | P H1] W1^T + b1.
// Same identity as diag_adjgrad_batch_ext -- d sum_p gamma_p H_p = sum_n <K_n, d Lambda_n> + sum_{n,c} <R_n[c,:], d J_n[c,:]> --
// but everything of a (sample n, class c) pair is local to the rows {n} + N(n), so there are no planes: one workgroup per pair
// runs the tangent along R = (dW0, db0, dW1, db1) and its reverse on those rows,
//     h1dot[b] = act'(h1[b]) * ([X | P X][b] dW0^T + db0)                     (neighbours b of n)
//     gradP[(n, b)] += <h1dot[b], W1n[c]> + <H1[b], dW1n[c]>                   (W1 = [W1s | W1n]: self / neighbour halves)
//     h1_bar[b] += P[n, b] dW1n[c],  h1_bar[n] += dW1s[c]
//     px_bar[r] += (act'(h1[r]) * coef_r) dW0n,  coef_b = P[n, b] W1n[c],  coef_n = W1s[c]        (adjoint of P X, dW0 = [dW0s | dW0n])
// candidate pairs (n, b') get the same first line for their b'.
__global__ __launch_bounds__(256) void sage_diag_pair_kernel(const int64_t* __restrict__ idx, int64_t N, int64_t C, int64_t H,
                                                             int64_t F, int64_t P, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, const float* __restrict__ val,
                                                             const float* __restrict__ cat0, const float* __restrict__ cat1,
                                                             const float* __restrict__ dact, const float* __restrict__ W1,
                                                             const float* __restrict__ R, const int32_t* __restrict__ ca,
                                                             const int32_t* __restrict__ cb, int64_t K,
                                                             float* __restrict__ grad_P, float* __restrict__ grad_cand,
                                                             float* __restrict__ h1_bar, float* __restrict__ px_bar,
                                                             int64_t px_ld) {
  extern __shared__ float sh[];  // w1s | w1n | dw1s | dw1n | cv | hd  [H each] | red [4]
  float* w1s = sh;
  float* w1n = w1s + H;
  float* dw1s = w1n + H;
  float* dw1n = dw1s + H;
  float* cv = dw1n + H;
  float* hd = cv + H;
  float* red = hd + H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t q = blockIdx.x;
  const int64_t m = q / C, c = q - m * C;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;
  const float* __restrict__ Rq = R + q * P;
  const float* __restrict__ dW0 = Rq;                  // [H][2F]
  const float* __restrict__ db0 = Rq + H * 2 * F;      // [H]
  const float* __restrict__ dW1c = db0 + H + c * 2 * H;  // row c of dW1 [C][2H]
  for (int64_t j = tid; j < H; j += 256) {
    w1s[j] = W1[c * 2 * H + j]; w1n[j] = W1[c * 2 * H + H + j];
    dw1s[j] = dW1c[j]; dw1n[j] = dW1c[H + j];
  }
  __syncthreads();
  // h1dot of row b -> hd[], then the pair's dot product (all threads get it)
  auto pair_dot = [&](int64_t b) -> float {
    for (int64_t j = wave; j < H; j += 4) {
      const float* __restrict__ wrow = dW0 + j * 2 * F;
      const float* __restrict__ xrow = cat0 + b * 2 * F;
      float acc = 0.f;
      for (int64_t i = lane; i < 2 * F; i += 64) acc += xrow[i] * wrow[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) hd[j] = dact[b * H + j] * (acc + db0[j]);
    }
    __syncthreads();
    float part = 0.f;
    for (int64_t j = tid; j < H; j += 256) part += hd[j] * w1n[j] + cat1[b * 2 * H + j] * dw1n[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();  // hd / red are rewritten by the next row
    return tot;
  };
  // rows {n} + N(n): slot -1 = the node itself
  const int32_t s0 = rowptr[n], s1 = rowptr[n + 1];
  for (int32_t p = s0 - 1; p < s1; ++p) {
    const bool self = p < s0;
    const int64_t row = self ? n : int64_t(col[p]);
    const float pv = self ? 1.f : val[p];
    for (int64_t j = tid; j < H; j += 256) {
      cv[j] = dact[row * H + j] * pv * (self ? w1s[j] : w1n[j]);
      atomicAdd(&h1_bar[row * H + j], pv * (self ? dw1s[j] : dw1n[j]));
    }
    __syncthreads();
    for (int64_t f = tid; f < F; f += 256) {
      float acc = 0.f;
      for (int64_t j = 0; j < H; ++j) acc += cv[j] * dW0[j * 2 * F + F + f];
      atomicAdd(&px_bar[row * px_ld + f], acc);
    }
    if (!self) {
      const float d = pair_dot(row);  // (starts with a barrier-free phase on hd, ends with barriers)
      if (tid == 0) atomicAdd(&grad_P[p], d);
    } else {
      __syncthreads();
    }
  }
  // candidate pairs that start at n (a = n in the propagation matrix's coordinates)
  for (int64_t k0 = 0; k0 < K; k0 += 256) {
    const int64_t k = k0 + tid;
    const bool mine = k < K && ca[k] == n;
    // every wave walks its own matches; the dot product needs the whole workgroup, so collect the matches first
    __shared__ int32_t match[256];
    __shared__ int nmatch;
    if (tid == 0) nmatch = 0;
    __syncthreads();
    if (mine) match[atomicAdd(&nmatch, 1)] = int32_t(k);
    __syncthreads();
    const int nm = nmatch;
    for (int i = 0; i < nm; ++i) {
      const int32_t kk = match[i];
      const float d = pair_dot(cb[kk]);
      if (tid == 0) atomicAdd(&grad_cand[kk], d);
    }
    __syncthreads();
  }
}

__global__ void add_strided_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ y, int64_t ldy, int64_t rows,
                                   int64_t width) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < rows * width; t += stride) {
    const int64_t r = t / width, j = t - r * width;
    x[r * ldx + j] += y[r * ldy + j];
  }
}

int diag_adjgrad_batch_sage(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, const float* gamma, float loss_scale,
                            float* grad_P, float* out_bar, float* h1_bar, float* e_bar, const int32_t* cand_a,
                            const int32_t* cand_b, int64_t K, float* grad_cand, hipStream_t s) {
  LGNN_REQUIRE(h->L == 2 && !h->extras(), "adjacency gradient, diagonal posterior, GraphSAGE: plain 2-layer models");
  LGNN_REQUIRE(h->act == LGNN_ACT_RELU && h->lik == LGNN_LIK_CLASSIFICATION, "adjacency gradient: ReLU, classification");
  LGNN_REQUIRE(M > 0 && idx && y && gamma && grad_P && out_bar && h1_bar && e_bar, "empty batch or null pointers");
  LGNN_CALL(forward_ensure_aux(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], F = h->dims[0], P = h->n_params;
  LGNN_REQUIRE(P == H * 2 * F + H + C * 2 * H + C, "internal: parameter count");
  LGNN_REQUIRE(h->fc.lin_in_ld[0] == 2 * F && h->fc.lin_in_ld[1] == 2 * H, "internal: unexpected cat row stride");
  LGNN_REQUIRE(size_t(C * C + C) * 4 <= 64 * 1024 && size_t(6 * H + 4) * 4 <= 60 * 1024, "too many classes / hidden units");
  const int64_t per_sample = C * 2 * P * 4;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(M, h->ws_limit / std::max<int64_t>(per_sample, 1)));
  LGNN_CALL(h->ws.jac.reserve(size_t(chunk) * C * P * 4));
  LGNN_CALL(h->ws.adj_dir.reserve(size_t(chunk) * C * P * 4));
  LGNN_CALL(h->ws.probs.reserve(size_t(chunk) * (C + C * C) * 4));
  float* J = h->ws.jac.as<float>();
  float* R = h->ws.adj_dir.as<float>();
  float* probs = h->ws.probs.as<float>();
  float* Kn = probs + chunk * C;
  const int64_t* yy = static_cast<const int64_t*>(y);
  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t mc = std::min(chunk, M - m0);
    LGNN_CALL(jacobians(h, idx + m0, mc, J, nullptr, s));
    hipLaunchKernelGGL(diag_ext_gram_kernel, dim3(unsigned(mc), unsigned(C * (C + 1) / 2)), dim3(256), 0, s, C, P, J, gamma, Kn);
    hipLaunchKernelGGL(diag_ext_sample_kernel, dim3(unsigned(mc)), dim3(64), size_t(C * C + C) * 4, s, idx + m0, yy + m0, N, C,
                       Kn, h->fc.out.as<float>(), loss_scale, probs, out_bar);
    hipLaunchKernelGGL(diag_ext_direction_kernel, dim3(unsigned(std::min<int64_t>(cdiv(mc * P, 256), 8192))), dim3(256), 0, s, J,
                       gamma, probs, mc, C, P, R);
    hipLaunchKernelGGL(sage_diag_pair_kernel, dim3(unsigned(mc * C)), dim3(256), size_t(6 * H + 4) * 4, s, idx + m0, N, C, H, F, P,
                       h->P.rowptr, h->P.col, h->P.val, h->fc.lin_in_p[0], h->fc.lin_in_p[1], h->fc.dact0.as<float>(), h->W[1], R,
                       cand_a, cand_b, K, grad_P, grad_cand, h1_bar, e_bar, F + 1);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  return 0;
}
}  // namespace

int kfac_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags, const float* gamma_B0,
                       const float* gamma_B1, const float* gamma_Br, float loss_scale, float* grad_P, float* out_bar,
                       const int32_t* cand_a, const int32_t* cand_b, int64_t K, float* grad_cand, hipStream_t s) {
  LGNN_REQUIRE(K == 0 || (cand_a && cand_b && grad_cand), "candidate pairs without their buffers");
  LGNN_REQUIRE(M > 0 && idx && y && gamma_B0 && gamma_B1 && grad_P && out_bar, "empty batch or null pointers");
  if (h->extras())
    return kfac_adjgrad_batch_ext(h, idx, y, M, (flags & LGNN_FLAG_FORK_EXACT_SEED) != 0, gamma_B0, gamma_B1, gamma_Br,
                                  loss_scale, grad_P, out_bar, cand_a, cand_b, K, grad_cand, s);
  LGNN_CALL(check_model(h));
  LGNN_CALL(forward_ensure_aux(h, s));  // act'(h_1) is part of the auxiliary forward products
  LGNN_CALL(ensure_wt(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], CC = C * C;
  const bool fork_exact = (flags & LGNN_FLAG_FORK_EXACT_SEED) != 0;
  if (h->kind == LGNN_KIND_SAGE)
    return sage_adjgrad_batch(h, idx, y, M, fork_exact, gamma_B0, gamma_B1, loss_scale, grad_P, out_bar, cand_a, cand_b, K,
                              grad_cand, s);
  LGNN_CALL(batch_prologue(h, idx, y, M, true, fork_exact, nullptr, s));

  // top layer: g1 planes [C][N][C] + flags / list of the rows that are not identically zero.  Without candidate pairs only
  // ACTIVE rows are ever read below (the zero-skipping SpMM, the active-row SDDMM + SpMM, the batch rows' gathers), so the
  // fit's own top-layer kernel writes just those rows (0.3 ms instead of 3.5 ms over all N rows) and the backward GEMM runs
  // over the compacted list (0.9 ms instead of 4.4 ms): rows that are not active hold whatever the buffers held and are
  // never looked at.  Candidate pairs can name any row: then everything is defined on all N rows as before.
  const bool compact = K == 0 && H == 256 && h->fc.mask_bits[0].p != nullptr && backgemm_supported(C, H, true) &&
                       (N + 1) * H * 4 < (int64_t(1) << 31) && getenv("LGNN_ADJ_ALL_ROWS") == nullptr &&
                       getenv("LGNN_ADJ_CHUNKED") == nullptr;
  const int64_t u_stride = compact ? (N + 1) * H : N * H;  // the compacted backward GEMM wants a spare row per plane
  LGNN_CALL(h->ws.top.reserve(size_t(N) * CC * 4 + 16));
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  float* g1 = h->ws.top.as<float>();
  if (compact) {
    LGNN_CALL(kfac_top_planes(h, idx, M, fork_exact, g1, s));
  } else {
    hipLaunchKernelGGL(seed_planes_kernel, dim3(unsigned(cdiv(N, 4))), dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N,
                       C, h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), g1, h->ws.active.as<uint8_t>());
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
    LGNN_CALL(h->ws.act_count.reserve(64));
    LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                            h->ws.select_tmp, s));
  }
  LGNN_CALL(h->ws.jac.reserve(size_t(M) * CC * 4));  // Vbar [M][C][C]
  float* vbar = h->ws.jac.as<float>();
  // values of P^T with the columns of inactive (all-zero) source rows removed
  LGNN_CALL(h->ws.val_act.reserve(size_t(std::max<int64_t>(h->nnz, 1)) * 4));
  float* val_act = h->ws.val_act.as<float>();
  hipLaunchKernelGGL(mask_values_by_flag_kernel, dim3(unsigned(std::min<int64_t>(cdiv(std::max<int64_t>(h->nnz, 1), 256), 4096))),
                     dim3(256), 0, s, h->PT.col, h->PT.val, h->nnz, h->ws.active.as<uint8_t>(), val_act);
  LGNN_HIP_CHECK(hipGetLastError());

  // class chunks: three [cc][N][H] plane buffers (u / ubar, g0, g0bar) + g1bar [cc][N][C] under the workspace cap
  const int64_t per_class = N * (3 * H + C) * 4;
  const int64_t cc_max = std::max<int64_t>(1, std::min<int64_t>(C, h->ws_limit / std::max<int64_t>(per_class, 1)));
  LGNN_CALL(h->ws.planes_a.reserve(size_t(cc_max) * (u_stride + N * H) * 4));
  LGNN_CALL(h->ws.planes_b.reserve(size_t(cc_max) * N * (H + C) * 4));
  h->ws.planes_a_zero_ptr = nullptr;
  for (int64_t c0 = 0; c0 < C; c0 += cc_max) {
    const int64_t cc = std::min(cc_max, C - c0);
    float* U = h->ws.planes_a.as<float>();
    float* G0 = U + cc_max * u_stride;
    float* G0B = h->ws.planes_b.as<float>();
    float* G1B = G0B + cc_max * N * H;
    const float* g1c = g1 + c0 * N * C;
    // u = mask * (g1 W1)     [cc * N, H]
    if (compact) {
      // the compacted backward GEMM of the fit: active rows only (row N of every plane takes its padding stores)
      BackGemmArgs bg{};
      bg.G = g1c; bg.W = h->W[1]; bg.ldw = H; bg.U = U; bg.u_plane_stride = u_stride; bg.N = N; bg.K = C; bg.Nout = H;
      bg.planes = cc;
      bg.rows = h->ws.act_list.as<int32_t>(); bg.na_dev = h->ws.act_count.as<int32_t>();
      bg.mask_bits = h->fc.mask_bits[0].as<uint32_t>(); bg.mask_words = cdiv(H, 32);
      LGNN_CALL(launch_backgemm(bg, s));
    } else {
      GemmEpilogue ep;
      ep.hact = h->fc.hact_p[0]; ep.hact_ld = h->fc.hact_ld[0]; ep.act = h->act; ep.hact_row_mod = N;
      LGNN_CALL(launch_gemm(g1c, C, h->W[1], H, U, H, cc * N, C, H, ep, s));
    }
    // g0 = P^T u  (source rows with u = 0 are not gathered: their values are zeroed)
    SpmmArgs sa{};
    sa.rowptr = h->PT.rowptr; sa.col = h->PT.col; sa.val = val_act; sa.nrows = N;
    sa.in = U; sa.in_ld = H; sa.in_plane_stride = u_stride; sa.out = G0; sa.out_ld = H; sa.out_plane_stride = N * H;
    sa.width = H; sa.out_act = -1;
    if (H % 4 == 0 && H <= 256 && N * H * 4 < (int64_t(1) << 32) - (int64_t(1) << 14) && cc < 65536) {
      hipLaunchKernelGGL(spmm256_skip_kernel, dim3(unsigned(cdiv(N, 4)), unsigned(cc)), dim3(256), 0, s, h->PT.rowptr, h->PT.col,
                         val_act, N, U, u_stride, H, G0);
      LGNN_HIP_CHECK(hipGetLastError());
    } else {
      LGNN_CALL(launch_spmm_ex(sa, cc, s));
    }
    // g0bar = 2 g0 Gamma_B0
    LGNN_CALL(sgemm_rm(s, cc * N, H, H, 2.f, G0, H, gamma_B0, H, 0.f, G0B, H));
    // gradP[(a,b)] += sum_c <u_c[a], g0bar_c[b]> and ubar = mask * (P g0bar) (overwrites u), both over the active rows only
    // (u, hence ubar's consumers, vanish elsewhere); candidates first: they read u before it is overwritten
    LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, U, H, N * H, G0B, H, N * H, H, cc, h->ws.active.as<uint8_t>(), grad_cand, s));
    if (H % 4 == 0 && H <= 256) {
      // (candidate pairs reach g1bar at arbitrary rows, so ubar is then needed everywhere, not on the active rows only)
      const int32_t* rows = K > 0 ? nullptr : h->ws.act_list.as<int32_t>();
      static const bool chunked = getenv("LGNN_ADJ_CHUNKED") != nullptr;  // dev: the 8-planes-per-wave kernel
      if (!chunked && cc < 65536)
        hipLaunchKernelGGL(sddmm_spmm_pm_kernel, dim3(unsigned(cdiv(N, 4)), unsigned(cc)), dim3(256), 0, s, h->P.rowptr, h->P.col,
                           h->P.val, rows, h->ws.act_count.as<int32_t>(), U, u_stride, G0B, N, H, h->fc.dact0.as<float>(), grad_P);
      else
        hipLaunchKernelGGL(sddmm_spmm_kernel<8>, dim3(unsigned(std::min<int64_t>(cdiv(N, 4), 8192))), dim3(256), 0, s,
                           h->P.rowptr, h->P.col, h->P.val, rows, h->ws.act_count.as<int32_t>(), U, G0B, N, H, cc,
                           h->fc.dact0.as<float>(), grad_P);
      LGNN_HIP_CHECK(hipGetLastError());
    } else {
      LGNN_CALL(launch_sddmm(h->P, N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(), U, H, N * H, G0B, H, N * H,
                             H, cc, grad_P, s));
      SpmmArgs sb{};
      sb.rowptr = h->P.rowptr; sb.col = h->P.col; sb.val = h->P.val; sb.nrows = N;
      sb.in = G0B; sb.in_ld = H; sb.in_plane_stride = N * H; sb.out = U; sb.out_ld = H; sb.out_plane_stride = N * H;
      sb.width = H; sb.out_act = -1;
      sb.hact = h->fc.hact_p[0]; sb.hact_ld = h->fc.hact_ld[0]; sb.act = h->act;
      LGNN_CALL(launch_spmm_ex(sb, cc, s));
    }
    // g1bar = ubar W1^T + 2 g1 Gamma_B1     [cc * N, C]
    if (compact) {  // planes of U are (N + 1) rows apart: one batched call, plane by plane
      rocblas_handle blas = static_cast<rocblas_handle>(blas_handle(s));
      LGNN_REQUIRE(blas != nullptr, "rocBLAS handle");
      const float one = 1.f, zero = 0.f;
      const rocblas_status st = rocblas_sgemm_strided_batched(
          blas, rocblas_operation_none, rocblas_operation_none, rocblas_int(C), rocblas_int(N), rocblas_int(H), &one,
          h->Wt[1].as<float>(), rocblas_int(C), 0, U, rocblas_int(H), rocblas_stride(u_stride), &zero, G1B, rocblas_int(C),
          rocblas_stride(N * C), rocblas_int(cc));
      if (st != rocblas_status_success) { set_error("rocblas_sgemm_strided_batched failed"); return 3; }
    } else {
      LGNN_CALL(sgemm_rm(s, cc * N, C, H, 1.f, U, H, h->Wt[1].as<float>(), C, 0.f, G1B, C));
    }
    LGNN_CALL(sgemm_rm(s, cc * N, C, C, 2.f, g1c, C, gamma_B1, C, 1.f, G1B, C));
    // gradP[(a,b)] += sum_c <G_c[a], g1bar_c[b]> for the batch rows a;  Vbar = (P g1bar)[batch rows]
    hipLaunchKernelGGL(sddmm_seed_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, h->P.rowptr, h->P.col, idx, M, N, C,
                       h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), G1B, c0, cc, grad_P);
    LGNN_HIP_CHECK(hipGetLastError());
    if (K > 0)
      hipLaunchKernelGGL(sddmm_coo_seed_kernel, dim3(unsigned(cdiv(K, 4))), dim3(256), 0, s, cand_a, cand_b, K, N, C,
                         h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), G1B, c0, cc, grad_cand);
    hipLaunchKernelGGL(seed_adjoint_gather_kernel, dim3(unsigned(cdiv(M * cc * C, 256))), dim3(256), 0, s, h->P.rowptr,
                       h->P.col, h->P.val, idx, M, N, C, G1B, c0, cc, vbar);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_REQUIRE(size_t(4) * 8 * C * 4 <= 64 * 1024, "too many classes for the seed adjoint kernel");
  hipLaunchKernelGGL(seed_adjoint_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), size_t(4) * 8 * C * 4, s,
                     h->fc.out.as<float>(), h->ws.probs.as<float>(), idx, static_cast<const int64_t*>(y), M, N, C, vbar,
                     fork_exact ? 1 : 0, loss_scale, out_bar);
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

namespace {
// ---------------------------------------------------------------------------------------------------------------------------
// Diagonal posterior (DiagLaplace on an STE-GCN: the shipped gnn/configs/original/stegcn_config.yaml:7).  With
// gamma_p = f / (2 (H_p f + delta_p)) held fixed (f = the likelihood's H_factor), d(-marglik) = f dCE + sum_p gamma_p dH_p, and the diagonal GGN of a
// 2-layer GCN is closed form per batch sample n (laplace/curvature/curvature.py:412-432 with the fork's
// attached Jacobians, :89-130), Ee = [P X | rowsum(P)], H1e = [H_1 | 1], g0e [H, F+1] / g1e [C, H+1] = gamma by (W | b) rows:
//     T_n[j, :] = sum_v P[n, v] mask[v, j] Ee[v, :]      H_{W0|b0}[j, :] += q_n[j] T_n[j, :]^2
//     phi_n     = (P H1e)[n]                             H_{W1|b1}[c, :] += p_c (1 - p_c) phi_n^2
//     q_n[j]    = sum_c p_c W1[c, j]^2 - (sum_c p_c W1[c, j])^2,   p_n = softmax(f_n)
// Reverse, per sample (oracle: diag_marglik_adj_grad, pinned to the reference's model.adj.grad):
//     r[j] = sum_i g0e[j, i] T[j, i]^2      a[c] = sum_j g1e[c, j] phi[j]^2
//     pbar[c] = (1 - 2 p_c) a[c] + sum_j r[j] (W1[c, j]^2 - 2 W1[c, j] m[j]),  m = p W1;   fbar = p * pbar - p (p . pbar) + CE'
//     phibar = 2 phi * (sum_c p_c (1 - p_c) g1e[c, :])        Tbar[j, :] = 2 q[j] g0e[j, :] T[j, :]
//     gradP[(n, v)] += sum_j mask[v, j] <Tbar[j, :], Ee[v, :]> + <phibar, H1e[v]>          (dadj_entry_kernel; candidates: dadj_cand_kernel)
//     h1_bar[v] += P[n, v] phibar[:H]        e_bar[v, :] += P[n, v] sum_j mask[v, j] Tbar[j, :]
// (h1_bar, e_bar and out_bar are propagated once per fit by adjgrad_finish.)  T of a chunk of samples lives in the workspace
// ([chunk][H][F + 1 rounded up to 4] floats: 0.5 GB for a Cora-shaped batch); one workgroup per sample throughout.

__global__ void add_inplace_kernel(float* __restrict__ x, const float* __restrict__ y, int64_t n) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < n; q += stride) x[q] += y[q];
}

// gradP[p] += c[a] for every entry p of row a (c has row stride ld);  candidates (a, b): grad_cand[k] += c[a]
__global__ __launch_bounds__(256) void row_const_kernel(const int32_t* __restrict__ rowptr, int64_t N, const float* __restrict__ c,
                                                        int64_t ld, float* __restrict__ gradP, const int32_t* __restrict__ ca,
                                                        int64_t K, float* __restrict__ grad_cand) {
  const int lane = threadIdx.x & 63;
  const int64_t a = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (a < N) {
    const float v = c[a * ld];
    for (int32_t p = rowptr[a] + lane; p < rowptr[a + 1]; p += 64) gradP[p] += v;
  }
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < K; k += stride) grad_cand[k] += c[int64_t(ca[k]) * ld];
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {  // red: 4 floats of LDS; every thread gets the sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();  // (red may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// The per-sample arrays of the tile kernel in LDS: r, q, mv [H], phi [H + 1], p, a, pb [C], red [4].
struct DadjSample {
  float *r, *q, *mv, *phi, *p, *a, *pb, *red;
  __device__ DadjSample(float* s, int64_t H, int64_t C) {
    r = s; q = r + H; mv = q + H; phi = mv + H; p = phi + H + 1; a = p + C; pb = a + C; red = pb + C;
  }
};

// T[m][j][:] = sum_v P[n, v] mask[v, j] Ee[v, :], n = idx[m0 + m] -- and everything else of the sample, in one pass over the
// tile: p, q (which needs the softmax and W_1 only) first; the row's entries staged VCH at a time, thread <-> 4 x 4 sub-tiles
// in registers (two 16-byte LDS reads per 16 FMAs and staged entry); with the LAST group of entries a sub-tile leaves as
// Tbar = 2 q g0e T and adds its share of r[j] = sum_i g0e T^2 (LDS atomics); then the last layer and the logits' adjoint.
__global__ __launch_bounds__(256) void dadj_tile_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ y,
                                                        int64_t m0, int64_t N, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, const float* __restrict__ val,
                                                        const float* __restrict__ mask, int64_t H, const float* __restrict__ PX,
                                                        int64_t ldx, const float* __restrict__ rowsum, int64_t F, int VCH,
                                                        const float* __restrict__ probs, int64_t C, const float* __restrict__ W1,
                                                        const float* __restrict__ PH, int64_t ldp,
                                                        const float* __restrict__ gamma, float loss_scale, float* __restrict__ T,
                                                        float* __restrict__ phibar, float* __restrict__ out_bar) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int64_t F1 = F + 1, FP = (F1 + 3) & ~int64_t(3), HP = (H + 3) & ~int64_t(3), H1 = H + 1;
  float* __restrict__ mk = sm;                     // [VCH][HP]  w_u * mask rows, zero padded
  float* __restrict__ ev = mk + int64_t(VCH) * HP; // [VCH][FP]  rows of Ee, zero padded
  const DadjSample S(ev + int64_t(VCH) * FP, H, C);
  const float* __restrict__ g0 = gamma;
  const float* __restrict__ gb0 = g0 + H * F;
  const float* __restrict__ g1 = gb0 + H;
  const float* __restrict__ gb1 = g1 + C * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m = m0 + blockIdx.x;
  const int64_t n = idx[m];
  float* __restrict__ Tm = T + int64_t(blockIdx.x) * H * FP;
  float* __restrict__ pbm = phibar + int64_t(blockIdx.x) * H1;
  if (n < 0 || n >= N) {  // flagged by the prologue: contributes nothing
    for (int64_t e = tid; e < H * FP; e += 256) Tm[e] = 0.f;
    for (int64_t j = tid; j < H1; j += 256) pbm[j] = 0.f;
    return;
  }
  for (int64_t c = tid; c < C; c += 256) S.p[c] = probs[m * C + c];
  for (int64_t j = tid; j < H1; j += 256) S.phi[j] = j < H ? PH[n * ldp + j] : rowsum[n];
  __syncthreads();
  for (int64_t j = tid; j < H; j += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float w = W1[c * H + j];
      s1 = fmaf(S.p[c], w, s1);
      s2 = fmaf(S.p[c] * w, w, s2);
    }
    S.mv[j] = s1;
    S.q[j] = s2 - s1 * s1;
    S.r[j] = 0.f;
  }
  const int32_t ps = rowptr[n], pe = rowptr[n + 1];
  if (ps == pe)
    for (int64_t e = tid; e < H * FP; e += 256) Tm[e] = 0.f;
  const int nI = int(FP / 4), nSub = int(HP / 4) * nI;
  for (int32_t p0 = ps; p0 < pe; p0 += VCH) {
    const int un = min(VCH, pe - p0);
    const bool last = p0 + VCH >= pe;
    __syncthreads();
    for (int64_t t = tid; t < int64_t(un) * (HP + FP); t += 256) {
      const int u = int(t / (HP + FP));
      const int64_t k = t - int64_t(u) * (HP + FP);
      const int64_t v = col[p0 + u];
      if (k < HP) mk[int64_t(u) * HP + k] = k < H ? val[p0 + u] * mask[v * H + k] : 0.f;
      else {
        const int64_t i = k - HP;
        ev[int64_t(u) * FP + i] = i < F ? PX[v * ldx + i] : (i == F ? rowsum[v] : 0.f);
      }
    }
    __syncthreads();
    for (int st = tid; st < nSub; st += 256) {
      const int jb = st / nI, ib = st - jb * nI;
      float acc[4][4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int64_t j = 4 * jb + rr;
        const float4 t4 = (p0 == ps || j >= H) ? make_float4(0.f, 0.f, 0.f, 0.f)
                                               : *reinterpret_cast<const float4*>(Tm + j * FP + 4 * ib);
        acc[rr][0] = t4.x; acc[rr][1] = t4.y; acc[rr][2] = t4.z; acc[rr][3] = t4.w;
      }
      for (int u = 0; u < un; ++u) {
        const float4 m4 = *reinterpret_cast<const float4*>(mk + int64_t(u) * HP + 4 * jb);
        const float4 e4 = *reinterpret_cast<const float4*>(ev + int64_t(u) * FP + 4 * ib);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) acc[rr][cc] = fmaf(mm[rr], ee[cc], acc[rr][cc]);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int64_t j = 4 * jb + rr;
        if (j >= H) continue;
        if (last) {  // T -> Tbar on the way out, r[j] from the unscaled entries
          const float qj = 2.f * S.q[j];
          float rs = 0.f;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int64_t i = 4 * ib + cc;
            const float g = i < F ? g0[j * F + i] : (i == F ? gb0[j] : 0.f);
            const float t = acc[rr][cc];
            rs = fmaf(g * t, t, rs);
            acc[rr][cc] = qj * g * t;
          }
          atomicAdd(&S.r[j], rs);
        }
        *reinterpret_cast<float4*>(Tm + j * FP + 4 * ib) = make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
      }
    }
  }
  __syncthreads();  // r is complete
  for (int64_t c = wave; c < C; c += 4) {
    float acc = 0.f;
    for (int64_t j = lane; j < H1; j += 64) acc = fmaf((j < H ? g1[c * H + j] : gb1[c]) * S.phi[j], S.phi[j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) S.a[c] = acc;
  }
  __syncthreads();
  for (int64_t c = wave; c < C; c += 4) {
    float acc = 0.f;
    for (int64_t j = lane; j < H; j += 64) {
      const float w = W1[c * H + j];
      acc = fmaf(S.r[j], w * w - 2.f * w * S.mv[j], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) S.pb[c] = (1.f - 2.f * S.p[c]) * S.a[c] + acc;
  }
  __syncthreads();
  float part = 0.f;
  for (int64_t c = tid; c < C; c += 256) part = fmaf(S.p[c], S.pb[c], part);
  const float dot = block_sum_256(part, S.red);
  const int64_t yc = y[m];
  for (int64_t c = tid; c < C; c += 256)
    atomicAdd(&out_bar[n * C + c], S.p[c] * (S.pb[c] - dot) + loss_scale * (S.p[c] - (c == yc ? 1.f : 0.f)));
  for (int64_t j = tid; j < H1; j += 256) {
    float lg = 0.f;
    for (int64_t c = 0; c < C; ++c) lg = fmaf(S.p[c] * (1.f - S.p[c]), j < H ? g1[c * H + j] : gb1[c], lg);
    pbm[j] = 2.f * S.phi[j] * lg;
  }
}

// value of one (sample, column node v) pair: sum_j mask[v, j] <Tbar[j, :], Ee[v, :]> + <phibar, H1e[v]> (candidate pairs; the
// stored entries take dadj_entry_kernel).  mk: [H] floats of LDS.  Every thread returns the value.
__device__ __forceinline__ float dadj_pair(const float* __restrict__ Tm, const float* __restrict__ pbm, int64_t v, int64_t H,
                                           int64_t F, const float* __restrict__ mask, const float* __restrict__ PX, int64_t ldx,
                                           const float* __restrict__ rowsum, const float* __restrict__ H1p, int64_t ldh,
                                           float* __restrict__ mk, float* __restrict__ red) {
  const int tid = threadIdx.x;
  const int64_t F1 = F + 1, FP = (F1 + 3) & ~int64_t(3);
  __syncthreads();
  for (int64_t j = tid; j < H; j += 256) mk[j] = mask[v * H + j];
  __syncthreads();
  float part = 0.f;
  for (int64_t i = tid; i < F1; i += 256) {
    float ca = 0.f;
    for (int64_t j = 0; j < H; ++j)
      if (mk[j] != 0.f) ca = fmaf(mk[j], Tm[j * FP + i], ca);
    part = fmaf(ca, i < F ? PX[v * ldx + i] : rowsum[v], part);
  }
  for (int64_t j = tid; j < H; j += 256) part = fmaf(pbm[j], H1p[v * ldh + j], part);
  if (tid == 0) part += pbm[H];
  return block_sum_256(part, red);
}

// The entries of a sample's row, EV at a time: ca[u][i] = sum_j mask[v_u, j] Tbar[j, i] is an (EV x H) . (H x F1) product per
// group.  Thread <-> (ib, jq): four columns of the tile (one 16-byte load per row) and every NJ-th row; the EV mask values of a
// row are two 16-byte LDS broadcasts; 32 FMAs per row visited.  The row slices meet in LDS (cas [EV][ICH], float atomics when
// NJ > 1), ICH columns at a time; the sums are then contracted with Ee[v_u, :] (gradP) and scattered (e_bar).
constexpr int EV = 8;
constexpr int ICH = 1024;  // columns of the tile per pass (cas: 32 KiB)
__global__ __launch_bounds__(256) void dadj_entry_kernel(const int64_t* __restrict__ idx, int64_t m0, int64_t N,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, const float* __restrict__ mask, int64_t H,
                                                         const float* __restrict__ PX, int64_t ldx,
                                                         const float* __restrict__ rowsum, int64_t F,
                                                         const float* __restrict__ H1p, int64_t ldh, const float* __restrict__ T,
                                                         const float* __restrict__ phibar, float* __restrict__ gradP,
                                                         float* __restrict__ h1_bar, float* __restrict__ e_bar) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int64_t F1 = F + 1, FP = (F1 + 3) & ~int64_t(3);
  const int icw = int(min<int64_t>(ICH, FP));   // columns per pass
  float* __restrict__ mk = sm;                   // [H][EV]
  float* __restrict__ cas = mk + H * EV;         // [EV][icw]
  float* __restrict__ red = cas + EV * icw;      // [4][EV]
  float* __restrict__ wv = red + 4 * EV;         // [EV]
  int32_t* __restrict__ vid = reinterpret_cast<int32_t*>(wv + EV);  // [EV]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = idx[m0 + blockIdx.x];
  if (n < 0 || n >= N) return;
  const float* __restrict__ Tm = T + int64_t(blockIdx.x) * H * FP;
  const float* __restrict__ pbm = phibar + int64_t(blockIdx.x) * (H + 1);
  const int32_t ps = rowptr[n], pe = rowptr[n + 1];
  for (int32_t p0 = ps; p0 < pe; p0 += EV) {
    const int un = min(EV, pe - p0);
    __syncthreads();
    if (tid < EV) {
      vid[tid] = tid < un ? col[p0 + tid] : 0;
      wv[tid] = tid < un ? val[p0 + tid] : 0.f;
    }
    __syncthreads();
    for (int64_t t = tid; t < H * EV; t += 256) {
      const int u = int(t % EV);
      mk[t] = u < un ? mask[int64_t(vid[u]) * H + t / EV] : 0.f;
    }
    float part[EV];
#pragma unroll
    for (int u = 0; u < EV; ++u) part[u] = 0.f;
    for (int64_t c0 = 0; c0 < FP; c0 += icw) {
      const int cw = int(min<int64_t>(icw, FP - c0));  // a multiple of 4
      const int nI = cw / 4;
      const int NJ = nI >= 256 ? 1 : int(min<int64_t>(256 / nI, H));  // row slices
      __syncthreads();  // mk staged / the previous pass's cas consumed
      if (NJ > 1)
        for (int t = tid; t < EV * cw; t += 256) cas[(t / cw) * icw + (t % cw)] = 0.f;
      __syncthreads();
      for (int st = tid; st < nI * NJ; st += 256) {
        const int jq = st / nI, ib = st - jq * nI;
        float ca[EV][4];
#pragma unroll
        for (int u = 0; u < EV; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) ca[u][c] = 0.f;
        for (int64_t j = jq; j < H; j += NJ) {
          const float4 t4 = *reinterpret_cast<const float4*>(Tm + j * FP + c0 + 4 * ib);
          const float4 m0v = *reinterpret_cast<const float4*>(mk + j * EV);
          const float4 m1v = *reinterpret_cast<const float4*>(mk + j * EV + 4);
          const float mm[EV] = {m0v.x, m0v.y, m0v.z, m0v.w, m1v.x, m1v.y, m1v.z, m1v.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
          for (int u = 0; u < EV; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) ca[u][c] = fmaf(mm[u], tt[c], ca[u][c]);
        }
#pragma unroll
        for (int u = 0; u < EV; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (NJ > 1) atomicAdd(&cas[u * icw + 4 * ib + c], ca[u][c]);
            else cas[u * icw + 4 * ib + c] = ca[u][c];
          }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < EV; ++u)
        if (u < un) {
          const int64_t v = vid[u];
          for (int i = tid; i < cw; i += 256) {
            const int64_t gi = c0 + i;
            if (gi >= F1) break;
            const float c = cas[u * icw + i];
            part[u] = fmaf(c, gi < F ? PX[v * ldx + gi] : rowsum[v], part[u]);
            if (c != 0.f) atomicAdd(&e_bar[v * F1 + gi], wv[u] * c);
          }
        }
    }
    for (int64_t j = tid; j < H; j += 256) {
      const float pj = pbm[j];
#pragma unroll
      for (int u = 0; u < EV; ++u)
        if (u < un) {
          const int64_t v = vid[u];
          part[u] = fmaf(pj, H1p[v * ldh + j], part[u]);
          atomicAdd(&h1_bar[v * H + j], wv[u] * pj);
        }
    }
#pragma unroll
    for (int u = 0; u < EV; ++u) {
      float x = part[u];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
      if (lane == 0) red[wave * EV + u] = x;
    }
    __syncthreads();
    if (tid < un)  // (a repeated node id: several samples share the entry, hence the atomic)
      atomicAdd(&gradP[p0 + tid], red[tid] + red[EV + tid] + red[2 * EV + tid] + red[3 * EV + tid] + pbm[H]);
  }
}

// candidate pairs (a, b): a contributes through the samples that are a -- the first one's tile, times its multiplicity
__global__ __launch_bounds__(256) void dadj_cand_kernel(const int32_t* __restrict__ ca, const int32_t* __restrict__ cb, int64_t K,
                                                        const int32_t* __restrict__ pos, const int32_t* __restrict__ mult,
                                                        int64_t m0, int64_t mc, const float* __restrict__ mask, int64_t H,
                                                        const float* __restrict__ PX, int64_t ldx,
                                                        const float* __restrict__ rowsum, int64_t F,
                                                        const float* __restrict__ H1p, int64_t ldh, const float* __restrict__ T,
                                                        const float* __restrict__ phibar, float* __restrict__ grad_cand) {
  extern __shared__ float sm[];
  float* __restrict__ mk = sm;
  float* __restrict__ red = mk + H;
  for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
    const int32_t m = pos[ca[k]];
    if (m == INT32_MAX || m < m0 || m >= m0 + mc) continue;  // (uniform over the workgroup)
    const float* __restrict__ Tm = T + int64_t(m - m0) * H * ((F + 4) & ~int64_t(3));
    const float* __restrict__ pbm = phibar + int64_t(m - m0) * (H + 1);
    const float tot = dadj_pair(Tm, pbm, cb[k], H, F, mask, PX, ldx, rowsum, H1p, ldh, mk, red);
    if (threadIdx.x == 0) grad_cand[k] += float(mult[m]) * tot;
  }
}

}  // namespace

int diag_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, const float* gamma, float loss_scale,
                       float* grad_P, float* out_bar, float* h1_bar, float* e_bar, const int32_t* cand_a, const int32_t* cand_b,
                       int64_t K, float* grad_cand, hipStream_t s) {
  LGNN_REQUIRE(K == 0 || (cand_a && cand_b && grad_cand), "candidate pairs without their buffers");
  if (h->kind == LGNN_KIND_SAGE)  // (e_bar [N, F + 1] carries the adjoint of P X in its first F columns)
    return diag_adjgrad_batch_sage(h, idx, y, M, gamma, loss_scale, grad_P, out_bar, h1_bar, e_bar, cand_a, cand_b, K, grad_cand,
                                   s);
  if (h->extras())  // res / norm: no closed form, (sample, class) planes (e_bar stays untouched)
    return diag_adjgrad_batch_ext(h, idx, y, M, gamma, loss_scale, grad_P, out_bar, h1_bar, cand_a, cand_b, K, grad_cand, s);
  LGNN_CALL(check_model(h));
  LGNN_REQUIRE(h->kind == LGNN_KIND_GCN, "adjacency gradient, diagonal posterior: GCN models");
  LGNN_REQUIRE(M > 0 && idx && y && gamma && grad_P && out_bar && h1_bar && e_bar, "empty batch or null pointers");
  LGNN_CALL(forward_ensure_aux(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], F = h->dims[0], F1 = F + 1;
  LGNN_REQUIRE(h->in_dim[1] == H, "internal: GCN head stride");
  LGNN_CALL(batch_prologue(h, idx, y, M, false, false, nullptr, s));
  // T of a chunk of samples under the workspace cap
  const int64_t FP = (F1 + 3) & ~int64_t(3), HP = (H + 3) & ~int64_t(3);  // tile rows / staged rows padded to 16 bytes
  const int64_t per_sample = H * FP * 4;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(M, h->ws_limit / per_sample));
  // tile kernel: staged entries (as many as fit, at most 8) + the per-sample arrays, within the 64 KiB of a default launch
  const size_t smem_s = size_t(3 * H + (H + 1) + 3 * C + 4) * 4;
  LGNN_REQUIRE(smem_s + size_t(HP + FP) * 4 <= 62 * 1024, "adjacency gradient, diagonal posterior: hidden + input width too large");
  const int vch = int(std::min<int64_t>(8, (62 * 1024 - int64_t(smem_s)) / ((HP + FP) * 4)));
  const size_t smem_t = size_t(vch) * (HP + FP) * 4;
  const size_t smem_e = size_t(H * 8 + 8 * std::min<int64_t>(1024, FP) + 4 * 8 + 8 + 8) * 4;  // (dadj_entry_kernel: EV = 8 mask rows + the column sums of a pass; the candidates' kernel needs H + 4)
  LGNN_REQUIRE(smem_e <= 60 * 1024, "adjacency gradient, diagonal posterior: hidden width too large");
  const float* mask = h->fc.dact0.as<float>();
  const float* PX = h->fc.prop_in[0].as<float>();
  const int64_t ldx = h->fc.prop_ld[0];
  const float* rowsum = h->fc.rowsum.as<float>();
  LGNN_CALL(h->ws.jac.reserve(size_t(chunk) * per_sample));
  LGNN_CALL(h->ws.misc.reserve(size_t(chunk) * (H + 1) * 4));
  float* T = h->ws.jac.as<float>();
  float* phibar = h->ws.misc.as<float>();
  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t mc = std::min(chunk, M - m0);
    hipLaunchKernelGGL(dadj_tile_kernel, dim3(unsigned(mc)), dim3(256), smem_t + smem_s, s, idx,
                       static_cast<const int64_t*>(y), m0, N, h->P.rowptr, h->P.col, h->P.val, mask, H, PX, ldx, rowsum, F, vch,
                       h->ws.probs.as<float>(), C, h->W[1], h->fc.prop_in[1].as<float>(), h->fc.prop_ld[1], gamma, loss_scale, T,
                       phibar, out_bar);
    hipLaunchKernelGGL(dadj_entry_kernel, dim3(unsigned(mc)), dim3(256), smem_e, s, idx, m0, N, h->P.rowptr, h->P.col, h->P.val,
                       mask, H, PX, ldx, rowsum, F, h->fc.hact_p[0], h->fc.hact_ld[0], T, phibar, grad_P, h1_bar, e_bar);
    if (K > 0)
      hipLaunchKernelGGL(dadj_cand_kernel, dim3(unsigned(std::min<int64_t>(K, 65535))), dim3(256), size_t(H + 4) * 4, s, cand_a, cand_b, K,
                         h->ws.pos.as<int32_t>(), h->ws.mult.as<int32_t>(), m0, mc, mask, H, PX, ldx, rowsum, F,
                         h->fc.hact_p[0], h->fc.hact_ld[0], T, phibar, grad_cand);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

// h1_bar [N, H] / e_bar [N, F + 1] (diagonal posterior, GCN: diag_adjgrad_batch): adjoints of H_1 and of [P X | rowsum(P)]
// that take the place of the Kronecker posterior's 2 a_scale H_1 Gamma_A1 (gamma_A1 is null then)
int adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* gamma_A0, const float* gamma_A1, float a1_scale,
                   float* grad_P, float* grad_adj, const int32_t* cand_a, const int32_t* cand_b, int64_t K, float* grad_cand,
                   float* grad_cand_adj, hipStream_t s, const float* h1_bar, const float* e_bar) {
  LGNN_REQUIRE(K == 0 || (cand_a && cand_b && grad_cand && grad_cand_adj), "candidate pairs without their buffers");
  if (h->extras()) LGNN_CALL(check_model_ext(h));
  else LGNN_CALL(check_model(h));
  LGNN_REQUIRE(out_bar && (gamma_A1 || (h1_bar && e_bar)) && grad_P && grad_adj, "null pointers");
  if (h->kind == LGNN_KIND_SAGE) {
    if (h1_bar && e_bar)  // diagonal posterior
      return sage_adjgrad_finish(h, out_bar, nullptr, nullptr, 0.f, grad_P, grad_adj, cand_a, cand_b, K, grad_cand, grad_cand_adj, s,
                                 h1_bar, e_bar, h->dims[0] + 1);
    LGNN_REQUIRE(gamma_A0 != nullptr, "GraphSAGE: the first layer's input covariance depends on the adjacency (gamma_A[0])");
    return sage_adjgrad_finish(h, out_bar, gamma_A0, gamma_A1, a1_scale, grad_P, grad_adj, cand_a, cand_b, K, grad_cand,
                               grad_cand_adj, s);
  }
  LGNN_CALL(forward_ensure(h, s));
  LGNN_CALL(ensure_wt(h, s));
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1], F = h->dims[0];
  // Z1 = H1 W1^T + b1 and Z0 = X W0^T + b0 (the forward keeps neither: one scratch serves both)
  const int64_t zw = std::max(H, C);
  LGNN_CALL(h->ws.planes_a.reserve(size_t(N) * (zw + H + C) * 4));
  h->ws.planes_a_zero_ptr = nullptr;
  float* Z = h->ws.planes_a.as<float>();  // Z1 [N, C], later Z0 [N, H]
  float* Hb = Z + N * zw;                   // H1bar [N, H]
  float* Zb = Hb + N * H;                   // Z1bar [N, C]
  GemmEpilogue eb1;
  eb1.bias = h->b[1];
  LGNN_CALL(launch_gemm(h->fc.hact_p[0], h->fc.hact_ld[0], h->Wt[1].as<float>(), C, Z, C, N, H, C, eb1, s));
  LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, out_bar, C, 0, Z, C, 0, C, 1, grad_P, s));
  LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, out_bar, C, 0, Z, C, 0, C, 1, nullptr, grad_cand, s));
  // Z1bar = P^T outbar;  H1bar = Z1bar W1 + 2 a1_scale H1 Gamma_A1;  P0bar = mask * H1bar
  LGNN_CALL(launch_spmm(h->PT, N, out_bar, C, Zb, C, C, 0, s));
  LGNN_CALL(sgemm_rm(s, N, H, C, 1.f, Zb, C, h->W[1], H, 0.f, Hb, H));
  if (gamma_A1) LGNN_CALL(sgemm_rm(s, N, H, H, 2.f * a1_scale, h->fc.hact_p[0], h->fc.hact_ld[0], gamma_A1, H, 1.f, Hb, H));
  if (h1_bar) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(unsigned(std::min<int64_t>(cdiv(N * H, 256), 4096))), dim3(256), 0, s, Hb, h1_bar,
                       N * H);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  if (h->extras()) {  // s_bar = norm_bwd(mask * H1bar): the mask and the norm's row-local backward (resnorm.hip)
    LGNN_CALL(launch_resnorm_backward(h, 0, Hb, H, N, nullptr, true, s));
  } else {
    hipLaunchKernelGGL(relu_mask_inplace_kernel, dim3(unsigned(std::min<int64_t>(cdiv(N * H, 256), 4096))), dim3(256), 0, s,
                       Hb, h->fc.hact_p[0], N * H);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  GemmEpilogue eb0;
  eb0.bias = h->b[0];
  LGNN_CALL(launch_gemm(h->fc.lin_in_p[0], h->fc.lin_in_ld[0], h->Wt[0].as<float>(), H, Z, H, N, F, H, eb0, s));
  LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, Hb, H, 0, Z, H, 0, H, 1, grad_P, s));
  LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, Hb, H, 0, Z, H, 0, H, 1, nullptr, grad_cand, s));
  if (e_bar) {
    // P X = sum_u P[v, u] X[u]: gradP[(v, u)] += <e_bar[v, :F], X[u]>;  rowsum(P)[v] = sum_u P[v, u]: += e_bar[v, F]
    const int64_t F1 = F + 1;
    LGNN_CALL(forward_input_view(h, s));
    LGNN_CALL(launch_sddmm(h->P, N, nullptr, nullptr, e_bar, F1, 0, h->fc.lin_in_p[0], h->fc.lin_in_ld[0], 0, F, 1, grad_P, s));
    LGNN_CALL(launch_sddmm_coo(cand_a, cand_b, K, e_bar, F1, 0, h->fc.lin_in_p[0], h->fc.lin_in_ld[0], 0, F, 1, nullptr,
                               grad_cand, s));
    hipLaunchKernelGGL(row_const_kernel, dim3(unsigned(cdiv(N, 4))), dim3(256), 0, s, h->P.rowptr, N, e_bar + F, F1, grad_P,
                       cand_a, K, grad_cand);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  // normalize_adj backward + the straight-through binarisation
  LGNN_CALL(h->ws.misc.reserve(size_t(2) * N * 4 + size_t(h->nnz) * 4));
  float* rs = h->ws.misc.as<float>();
  float* cs = rs + N;
  float* tmp = cs + N;
  LGNN_HIP_CHECK(hipMemsetAsync(cs, 0, size_t(N) * 4, s));
  hipLaunchKernelGGL(gp_rowcol_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val, grad_P,
                     N, rs, cs);
  float* first = h->sym ? tmp : grad_adj;
  hipLaunchKernelGGL(adj_grad_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->A.rowptr, h->A.col, h->P.rowptr,
                     h->P.col, grad_P, rs, cs, N, first);
  if (h->sym)
    hipLaunchKernelGGL(adj_grad_symmetrize_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->A.rowptr, h->A.col, tmp,
                       N, grad_adj);
  if (K > 0)  // candidates: un-symmetrised d/dA[i, j]; a symmetric model's caller passes both orientations and averages
    hipLaunchKernelGGL(cand_adj_grad_kernel, dim3(unsigned(cdiv(K, 256))), dim3(256), 0, s, cand_a, cand_b, K, h->A.rowptr,
                       grad_cand, rs, cs, grad_cand_adj);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn

extern "C" int lgnn_kfac_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags,
                                       const float* const* gamma_B, float loss_scale, float* grad_P, float* out_bar,
                                       const int32_t* cand_a, const int32_t* cand_b, int64_t num_cand, float* grad_cand,
                                       void* stream) {
  if (!h || !gamma_B) { lgnn::set_error("null argument"); return 2; }
  return lgnn::kfac_adjgrad_batch(h, idx, y, M, flags, gamma_B[0], gamma_B[1], h->has_res ? gamma_B[2] : nullptr, loss_scale,
                                  grad_P, out_bar, cand_a, cand_b, num_cand, grad_cand, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* const* gamma_A, float a_scale,
                                   float* grad_P, float* grad_adj, const int32_t* cand_a, const int32_t* cand_b,
                                   int64_t num_cand, float* grad_cand, float* grad_cand_adj, void* stream) {
  if (!h || !gamma_A) { lgnn::set_error("null argument"); return 2; }
  return lgnn::adjgrad_finish(h, out_bar, gamma_A[0], gamma_A[1], a_scale, grad_P, grad_adj, cand_a, cand_b, num_cand, grad_cand,
                              grad_cand_adj, static_cast<hipStream_t>(stream), nullptr, nullptr);
}

extern "C" int lgnn_diag_adjgrad_batch(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, const float* gamma,
                                       float loss_scale, float* grad_P, float* out_bar, float* h1_bar, float* e_bar,
                                       const int32_t* cand_a, const int32_t* cand_b, int64_t num_cand, float* grad_cand,
                                       void* stream) {
  if (!h) { lgnn::set_error("null context"); return 2; }
  return lgnn::diag_adjgrad_batch(h, idx, y, M, gamma, loss_scale, grad_P, out_bar, h1_bar, e_bar, cand_a, cand_b, num_cand,
                                  grad_cand, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_diag_adjgrad_finish(lgnn_ctx* h, const float* out_bar, const float* h1_bar, const float* e_bar,
                                        float* grad_P, float* grad_adj, const int32_t* cand_a, const int32_t* cand_b,
                                        int64_t num_cand, float* grad_cand, float* grad_cand_adj, void* stream) {
  if (!h) { lgnn::set_error("null context"); return 2; }
  if (!h1_bar || !e_bar) { lgnn::set_error("lgnn_diag_adjgrad_finish: null adjoint buffers"); return 2; }
  return lgnn::adjgrad_finish(h, out_bar, nullptr, nullptr, 0.f, grad_P, grad_adj, cand_a, cand_b, num_cand, grad_cand,
                              grad_cand_adj, static_cast<hipStream_t>(stream), h1_bar, e_bar);
}
