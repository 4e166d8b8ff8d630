// KFAC factors of one mini-batch for L-layer GCN / GraphSAGE models.
//
// Reference: CurvlinopsInterface.kron (laplace/curvature/curvlinops.py:77-108) ->
// KFACLinearOperator._compute_kfac (curvlinops/kfac.py:540-581): forward pre-hooks accumulate the
// input covariance of every nn.Linear (kfac.py:819-875), then one backward pass per class column c
// of the loss-Hessian square root (kfac.py:653-661) accumulates g^T g at every Linear output
// (kfac.py:777-817).  The reference runs C dense N x N backward passes through autograd; here the
// model family is closed form, so the C right-hand sides travel together as class-major planes
// T[c][n][w] through explicit kernels:
//     seeds  V[m][c][k]                 (softmax + fork-exact v_c, one wave per sample)
//     top    g_{L-1} = P^T scatter(V)   (seed SpMM: only neighbours that are batch nodes contribute)
//     down   up = act'(h) * (g_l W_l)   (fp32 MFMA GEMM, epilogue mask)
//            g_{l-1} = P^T up           (fused SpMM^T -> LDS tile -> MFMA Gram; g never hits HBM
//                                        unless a lower layer needs it)
#include "lgnn_internal.h"

namespace lgnn {

namespace {

__global__ void mark_batch_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int32_t* __restrict__ pos,
                                  int* __restrict__ bad) {
  const int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) { bad[1] = 1; return; }  // sticky flag word 1: node id out of range
  atomicMin(&pos[n], int32_t(m));  // duplicates: the first occurrence owns the accumulated seed row
}
__global__ void unmark_batch_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int32_t* __restrict__ pos) {
  const int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n >= 0 && n < N) pos[n] = INT32_MAX;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per sample.  f = logits[idx[m]], p = softmax(f), mbar = sum_k p_k f_k.
//   fork exact : V[k,c] = sqrt(p_c) [d_kc - p_k (1 + f_k - mbar) + 1/2 (d_kc - p_k)(f_c - mbar)]
//   upstream   : V[k,c] = sqrt(p_c) (d_kc - p_k)                       (kfac_utils.py:122-126)
// written as seeds[first][c][k] (+=: duplicated node ids accumulate like x[x_indices]' backward).
// loss += logsumexp(f) - f[y]   (CrossEntropyLoss(reduction='sum'))
// seed modes: 0 upstream, 1 fork exact, 2 regression (sqrt(2) I), and the single-column "gradient" seeds of the empirical
// / Monte-Carlo Fisher (curvlinops/kfac.py:663-674: ONE backward pass of the loss itself): 3 classification
// V[k, 0] = rs (p_k - [k == y_seed]), 4 regression V[k, 0] = rs (f_k - y_seed_k); columns c > 0 stay zero.
__global__ __launch_bounds__(256) void seed_kernel(const float* __restrict__ logits, int64_t C,
                                                   const int64_t* __restrict__ idx, const void* __restrict__ y_,
                                                   int64_t M, int64_t N, const int32_t* __restrict__ pos, int fork_exact,
                                                   float* __restrict__ seeds, float* __restrict__ probs,
                                                   float* __restrict__ loss, int* __restrict__ bad,
                                                   int32_t* __restrict__ mult, const void* __restrict__ yseed_,
                                                   float resid_scale) {
  extern __shared__ float sm[];
  __shared__ float loss_part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* f_s = sm + size_t(wave) * 2 * C;
  float* p_s = f_s + C;
  float loss_acc = 0.f;  // lane 0: this wave's share of the batch loss (one atomic per workgroup, not per sample)
  for (int64_t m = int64_t(blockIdx.x) * 4 + wave; m < M; m += int64_t(gridDim.x) * 4) {
    const int64_t n = idx[m];
    if (n < 0 || n >= N) continue;  // flagged by mark_batch_kernel
    float mx = -INFINITY;
    for (int64_t k = lane; k < C; k += 64) {
      const float v = logits[n * C + k];
      f_s[k] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int64_t k = lane; k < C; k += 64) {
      const float e = expf(f_s[k] - mx);
      p_s[k] = e;
      se += e;
    }
    se = wave_sum(se);
    const float inv = 1.0f / se;
    float mb = 0.f;
    for (int64_t k = lane; k < C; k += 64) {
      const float p = p_s[k] * inv;
      p_s[k] = p;
      if (probs) probs[m * C + k] = p;
      mb += p * f_s[k];
    }
    mb = wave_sum(mb);
    // f_s / p_s are read across lanes below: same wave, LDS executes a wave's operations in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (fork_exact == 2 || fork_exact == 4) {
      // regression: sum_k (f_k - y_k)^2 with float targets [M, C] (MSELoss(reduction='sum'); the factor 0.5 of the
      // interface is applied by the caller)
      if (loss) {
        const float* __restrict__ yr = static_cast<const float*>(y_) + m * C;
        float sq = 0.f;
        for (int64_t k = lane; k < C; k += 64) { const float dlt = f_s[k] - yr[k]; sq += dlt * dlt; }
        sq = wave_sum(sq);
        if (lane == 0) loss_acc += sq;
      }
    } else if (lane == 0 && loss) {
      const int64_t yy = static_cast<const int64_t*>(y_)[m];
      if (yy < 0 || yy >= C) *bad = 2;
      else loss_acc += logf(se) + mx - f_s[yy];
    }
    if (lane == 0 && mult) atomicAdd(&mult[pos[n]], 1);  // how often the node occurs in the batch, kept at its first place
    if (seeds && fork_exact >= 3) {
      float* __restrict__ dst = seeds + int64_t(pos[n]) * C * C;  // row c = 0 of the block
      if (fork_exact == 3) {
        const int64_t ys = static_cast<const int64_t*>(yseed_)[m];
        if (ys < 0 || ys >= C) { if (lane == 0) *bad = 2; }
        else for (int64_t k = lane; k < C; k += 64) atomicAdd(&dst[k], resid_scale * (p_s[k] - (k == ys ? 1.f : 0.f)));
      } else {
        const float* __restrict__ yr = static_cast<const float*>(yseed_) + m * C;
        for (int64_t k = lane; k < C; k += 64) atomicAdd(&dst[k], resid_scale * (f_s[k] - yr[k]));
      }
    } else if (seeds) {
      const int64_t first = pos[n];
      float* __restrict__ dst = seeds + first * C * C;
      const int64_t CC = C * C;
      for (int64_t q = lane; q < CC; q += 64) {
        const int64_t c = q / C, k = q - c * C;
        const float pc = p_s[c], pk = p_s[k];
        const float d = (k == c) ? 1.f : 0.f;
        float v;
        if (fork_exact == 2) v = d * 1.41421356237309515f;  // regression: Hessian square root sqrt(2) I (kfac_utils.py:116-120)
        else if (fork_exact) v = sqrtf(pc) * (d - pk * (1.f + f_s[k] - mb) + 0.5f * (d - pk) * (f_s[c] - mb));
        else v = sqrtf(pc) * (d - pk);
        atomicAdd(&dst[q], v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the next sample overwrites f_s / p_s
  }
  if (loss) {
    if (lane == 0) loss_part[wave] = loss_acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (loss_part[0] + loss_part[1]) + (loss_part[2] + loss_part[3]));
  }
}

// GCN top layer: g[c][n][k] = sum_{v in row n of P^T} val * [v in batch] * seeds[pos[v]][c][k].
// One wave per node; the (few) batch neighbours are found with a ballot, their seed rows are
// accumulated in an LDS row of C*C floats and written out class-major.
__global__ void seed_spmm_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                 const float* __restrict__ val, int64_t N, int64_t C,
                                 const int32_t* __restrict__ pos, const float* __restrict__ seeds,
                                 float* __restrict__ g, uint8_t* __restrict__ active, int64_t cb, int64_t ce) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = int64_t(blockIdx.x) * (blockDim.x >> 6) + wave;
  if (n >= N) return;
  const int64_t CC = C * C;
  const int64_t q0 = cb * C, nq = (ce - cb) * C;  // only the class planes [cb, ce) are built
  float* buf = sm + int64_t(wave) * nq;
  bool any = false;
  const int32_t s = rowptr[n], e = rowptr[n + 1];
  for (int32_t base = s; base < e; base += 64) {
    const int32_t p = base + lane;
    int32_t mp = INT32_MAX;
    float v = 0.f;
    if (p < e) { mp = pos[col[p]]; v = val[p]; }
    unsigned long long mask = __ballot(mp != INT32_MAX);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int64_t mm = __shfl(mp, b);
      const float vv = __shfl(v, b);
      const float* __restrict__ src = seeds + mm * CC;
      if (!any) {
        for (int64_t q = lane; q < nq; q += 64) buf[q] = vv * src[q0 + q];
        any = true;
      } else {
        for (int64_t q = lane; q < nq; q += 64) buf[q] += vv * src[q0 + q];
      }
    }
  }
  // each lane re-reads only what it wrote itself (q = lane mod 64): no barrier needed
  for (int64_t q = lane; q < nq; q += 64) {
    const int64_t c = (q0 + q) / C, k = (q0 + q) - c * C;
    g[(c * N + n) * C + k] = any ? buf[q] : 0.f;
  }
  if (lane == 0) active[n] = any ? 1 : 0;
}


// active[v] = 1 for every node v that has a batch node among its P^T neighbours, i.e. every column of the P rows of
// the batch nodes: these are the only rows of the top-layer gradient that are not identically zero.  One wave per sample.
__global__ __launch_bounds__(256) void mark_active_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, uint8_t* __restrict__ active) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;  // flagged by mark_batch_kernel
  const int32_t e = rowptr[n + 1];
  for (int32_t p = rowptr[n] + lane; p < e; p += 64) active[col[p]] = 1;
}

// GraphSAGE: the rows of the top-layer gradient that are not identically zero are the batch nodes themselves
__global__ void mark_batch_flags_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, uint8_t* __restrict__ active) {
  const int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int64_t n = idx[m];
  if (n >= 0 && n < N) active[n] = 1;
}

// zero the listed rows of every plane again (16 bytes per thread): U[p][rows[i]][0 .. width)
__global__ void clear_rows_kernel(float* __restrict__ U, int64_t plane_stride, int64_t width, int64_t planes,
                                  const int32_t* __restrict__ rows, const int32_t* __restrict__ count) {
  const int64_t w4 = width / 4, per_plane = int64_t(*count) * w4, total = per_plane * planes;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < total; q += stride) {
    const int64_t p = q / per_plane, r = q - p * per_plane;
    const int64_t i = r / w4, c4 = r - i * w4;
    *reinterpret_cast<float4*>(U + p * plane_stride + int64_t(rows[i]) * width + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

// GCN top layer over the ACTIVE nodes only, with the top-layer Gram fused in:
//     G_n[c][k] = sum_{v in row n of P^T, v in batch} val * V_v[k, c]                 (c in [cb, ce))
//     g[c][n][:] = G_n[c][:]                        (class-major planes for the backward GEMM; skipped when !g)
//     S        += G_n^T G_n = sum_c g_c[n]^T g_c[n]  (v_mfma_f32_16x16x4_f32 on the LDS copy of G_n)
// One wave per active node, persistent: the NBLK*(NBLK+1)/2 upper 16x16 tiles of S stay in registers, are reduced
// through LDS once per workgroup and leave with one float atomic per element.
//
// The seed block V_v of a batch sample is never materialised: it is diagonal plus rank two,
//     fork exact : V[k,c] = sqrt(p_c) [d_kc - p_k (1 + f_k - mbar) + 1/2 (d_kc - p_k)(f_c - mbar)]
//                         = alpha_c d_kc - beta_c u_k - gamma_c p_k
//     upstream   : V[k,c] = sqrt(p_c) (d_kc - p_k)                      (gamma = 0, u = p)
// so the wave loads the sample's C probabilities and logits (320 B instead of the 6.4 KB block), lane c keeps
// (alpha, beta, gamma)_c, lane k keeps (u, p)_k, and row c of G_n is built by lane c from v_readlane broadcasts,
// two columns per step.  The kernel is issue bound (a wave64 VALU instruction occupies its SIMD for 4 cycles), which
// is why the instruction count per node, not the bytes, is what the layout below minimises.
// Before: seed blocks written by seed_kernel (64 MB per arxiv-shaped batch), read back 1.5 times per active node,
// planes for all N rows to HBM, a separate Gram kernel over them: 0.14 + 0.54 + 0.62 ms per batch.
template <int NBLK>
__global__ __launch_bounds__(1024) void seed_spmm_gram_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val, int64_t N, int C,
    const int32_t* __restrict__ pos, const float* __restrict__ probs, const float* __restrict__ logits,
    const int32_t* __restrict__ mult, int fork_exact, float* __restrict__ g, const int32_t* __restrict__ act_list,
    const int32_t* __restrict__ act_count, int cb, int ce, float* __restrict__ scratch, int ldb, int debug_arg,
    // hub rows cut into slices (see top_tasks_* below).  mode 0: act_list holds node ids, a node's task is its whole row;
    // mode 1: act_list holds (node or -1 - node, begin, end) triples -- a negative node marks a SLICE of a hub row, whose
    // partial tile is added to hub_tiles[long_slot[node]] instead of being finished here; mode 2: act_list holds the ids of
    // the sliced hubs, their summed tiles are read back from hub_tiles and finished (planes + Gram)
    int mode, const int32_t* __restrict__ long_slot, float* __restrict__ hub_tiles, const uint8_t* __restrict__ active) {
#ifdef LGNN_DEV  // ablation switches exist in `make DEV=1` builds only
  const int debug = debug_arg;
#else
  (void)debug_arg;
  constexpr int debug = 0;
#endif
  constexpr int NT = NBLK * (NBLK + 1) / 2;
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int nwaves = blockDim.x >> 6;
  const int nrows = ce - cb, rp = (nrows + 3) & ~3;
  // buf[r][k], r = c - cb < rp, k < ldb (>= C rounded up to 8; == 2 mod 4 so that the 16 lanes of an MFMA operand read and
  // the 8-byte row writes of 16 consecutive lanes fall into distinct banks).  Rows >= nrows stay zero; columns >= C are
  // masked when read.
  float* __restrict__ buf = sm + size_t(wave) * rp * ldb;
  for (int q = lane; q < rp * ldb; q += 64) buf[q] = 0.f;
  const bool rowok = lane >= cb && lane < ce;
  float* __restrict__ rowp = buf + (rowok ? lane - cb : 0) * ldb;
  bool colok[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) colok[b] = b * 16 + (lane & 15) < C;
  // plane stores: 16 bytes per lane when C % 4 == 0, else 4
  const bool vec = (C & 3) == 0;
  const int cw = vec ? C >> 2 : C;                // stored units per row
  const int nunits = nrows * cw;
  const int step_r = 64 / cw, step_k = 64 % cw, r0 = lane / cw, k0 = lane % cw;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int total = *act_count;
  // Node j of this wave is act_list[gw + j * S].  list -> rowptr -> col -> pos is a chain of dependent loads; it is
  // taken off the critical path: list and rowptr for 64 of the wave's nodes at once (one per lane), col/val two nodes
  // ahead, pos one node ahead.
  const int gw = blockIdx.x * nwaves + wave, S = gridDim.x * nwaves;
  const int cnt = gw < total ? (total - gw + S - 1) / S : 0;
  for (int c0 = 0; c0 < cnt; c0 += 64) {
    const int cn = min(64, cnt - c0);
    int32_t n_l = 0, s_l = 0, e_l = 0;
    if (lane < cn) {
      const int64_t t = gw + int64_t(c0 + lane) * S;
      if (mode == 1) {
        n_l = act_list[3 * t];
        s_l = act_list[3 * t + 1];
        e_l = act_list[3 * t + 2];
      } else {
        n_l = act_list[t];
        if (mode == 0) { s_l = rowptr[n_l]; e_l = rowptr[n_l + 1]; }  // mode 2: no entries, the tile comes from hub_tiles
      }
    }
    auto load_entries = [&](int j, int32_t& cj, float& vj) {
      const int jj = min(j, 63);
      const int32_t sj = __builtin_amdgcn_readlane(s_l, jj), ej = __builtin_amdgcn_readlane(e_l, jj);
      cj = -1; vj = 0.f;
      if (j < cn && sj + lane < ej) { cj = col[sj + lane]; vj = val[sj + lane]; }
    };
    int32_t cA, cB, cC, mpA, mpB;
    float vA, vB, vC;
    load_entries(0, cA, vA);
    mpA = cA >= 0 ? pos[cA] : INT32_MAX;
    load_entries(1, cB, vB);
    for (int j = 0; j < cn; ++j) {
      mpB = cB >= 0 ? pos[cB] : INT32_MAX;  // node j + 1
      load_entries(j + 2, cC, vC);           // node j + 2
      const int32_t n_enc = __builtin_amdgcn_readlane(n_l, j);
      const bool slice = n_enc < 0;  // a slice of a hub row: its partial tile goes to hub_tiles
      const int64_t n = slice ? -1 - int64_t(n_enc) : int64_t(n_enc);
      const int32_t s = __builtin_amdgcn_readlane(s_l, j), e = __builtin_amdgcn_readlane(e_l, j);
      bool any = false;
      if (mode == 2) {  // the summed tile of a sliced hub
        any = active[n] != 0;
        if (any) {
          const float* __restrict__ src = hub_tiles + int64_t(long_slot[n]) * nrows * C;
          for (int q = lane; q < nrows * C; q += 64) buf[(q / C) * ldb + (q % C)] = src[q];
        }
      }
      for (int32_t base = s; base < e; base += 64) {
        int32_t mp = mpA, cu = cA;
        float v = vA;
        if (base > s) {  // rows with more than 64 stored entries: the rest is not prefetched
          const int32_t p = base + lane;
          mp = INT32_MAX; v = 0.f; cu = 0;
          if (p < e) { cu = col[p]; mp = pos[cu]; v = val[p]; }
        }
        unsigned long long mask = __ballot(mp != INT32_MAX);
        while (mask) {
          const int b = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const int64_t mm = __shfl(mp, b);
          const int64_t u = __shfl(cu, b);
          float pk = 0.f, fk = 0.f;
          if (lane < C && fork_exact != 2) { pk = probs[mm * C + lane]; fk = logits[u * C + lane]; }
          // a node listed t times in the batch counts t times (the dense reference's x[x_indices] backward)
          const float vv = __shfl(v, b) * float(mult[mm]);
          const float mb = wave_sum(pk * fk);  // same summation order as seed_kernel
          const float sp = sqrtf(pk), t = fk - mb;
          // fork_exact: 0 upstream, 1 fork-exact classification seeds; 2 regression: V = sqrt(2) I (only the diagonal term)
          const float alpha = fork_exact == 2 ? 1.41421356237309515f : (fork_exact ? sp * (1.f + 0.5f * t) : sp);
          const float nbeta = -sp, ngamma = fork_exact == 1 ? -0.5f * sp * t : 0.f;
          int uvi = __float_as_int(vv * (fork_exact == 1 ? pk * (1.f + t) : pk));
          int pvi = __float_as_int(fork_exact == 1 ? vv * pk : 0.f);
          // Both are read with v_readlane (which ignores EXEC) inside the rowok region below, from lanes that are not
          // part of it: pin their computation here, for all lanes, so that it is not sunk into that region.
          asm volatile("" : "+v"(uvi), "+v"(pvi));
          // 4 column pairs per trip, no guards inside (ldb covers C rounded up to 8; lanes >= C hold zeros); the
          // readlanes are convergent, so only constant trip counts unroll
          if (rowok) {
            if (!any) {
              for (int k8 = 0; k8 < C; k8 += 8) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                  const int k = k8 + 2 * w;
                  const float u0 = __int_as_float(__builtin_amdgcn_readlane(uvi, k));
                  const float u1 = __int_as_float(__builtin_amdgcn_readlane(uvi, k + 1));
                  const float p0 = __int_as_float(__builtin_amdgcn_readlane(pvi, k));
                  const float p1 = __int_as_float(__builtin_amdgcn_readlane(pvi, k + 1));
                  *reinterpret_cast<float2*>(rowp + k) = make_float2(nbeta * u0 + ngamma * p0, nbeta * u1 + ngamma * p1);
                }
              }
            } else {
              for (int k8 = 0; k8 < C; k8 += 8) {
                float2 o[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) o[w] = *reinterpret_cast<float2*>(rowp + k8 + 2 * w);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                  const int k = k8 + 2 * w;
                  const float u0 = __int_as_float(__builtin_amdgcn_readlane(uvi, k));
                  const float u1 = __int_as_float(__builtin_amdgcn_readlane(uvi, k + 1));
                  const float p0 = __int_as_float(__builtin_amdgcn_readlane(pvi, k));
                  const float p1 = __int_as_float(__builtin_amdgcn_readlane(pvi, k + 1));
                  o[w].x += nbeta * u0 + ngamma * p0; o[w].y += nbeta * u1 + ngamma * p1;
                  *reinterpret_cast<float2*>(rowp + k) = o[w];
                }
              }
            }
          }
          if (rowok) rowp[lane] += vv * alpha;  // the diagonal term, column k = c
          any = true;
        }
      }
      mpA = mpB; cA = cB; vA = vB; cB = cC; vB = vC;
      if (!any) continue;  // a slice without batch neighbours, an inactive hub; cannot happen for a listed whole node
      // the other lanes' LDS writes are read below: same wave, LDS executes a wave's operations in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (slice) {
        float* __restrict__ dst = hub_tiles + int64_t(long_slot[n]) * nrows * C;
        for (int q = lane; q < nrows * C; q += 64) atomicAdd(&dst[q], buf[(q / C) * ldb + (q % C)]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next task overwrites the tile
        continue;
      }
      if (g && debug != 1) {
        int r = r0, k = k0;
        float* __restrict__ gn = g + (int64_t(cb) * N + n) * C;
        const int64_t plane = N * C;
        for (int qb = lane; qb < nunits; qb += 64 * 4) {
          float4 t4[4];
          int64_t off[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const bool ok = qb + 64 * w < nunits;
            if (vec) {
              const float* src = buf + r * ldb + 4 * k;
              const float2 lo = ok ? *reinterpret_cast<const float2*>(src) : make_float2(0.f, 0.f);
              const float2 hi = ok ? *reinterpret_cast<const float2*>(src + 2) : make_float2(0.f, 0.f);
              t4[w] = make_float4(lo.x, lo.y, hi.x, hi.y);
              off[w] = r * plane + 4 * k;
            } else {
              t4[w].x = ok ? buf[r * ldb + k] : 0.f;
              off[w] = r * plane + k;
            }
            r += step_r; k += step_k;
            if (k >= cw) { k -= cw; ++r; }
          }
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (qb + 64 * w < nunits) {
              if (vec) *reinterpret_cast<float4*>(gn + off[w]) = t4[w];
              else gn[off[w]] = t4[w].x;
            }
          }
        }
      }
      const float* __restrict__ xb = buf + (lane >> 4) * ldb + (lane & 15);
      for (int kk = 0; kk < (debug == 2 ? 0 : rp); kk += 4) {
        float x[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) x[b] = colok[b] ? xb[kk * ldb + b * 16] : 0.f;
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
          for (int bj = bi; bj < NBLK; ++bj, ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[bi], x[bj], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // workgroup reduction of the register tiles through LDS, then one atomic per upper-triangular element
  __syncthreads();
  float* __restrict__ red = sm;  // NT * 256 floats (the launcher sizes the allocation for it)
  for (int q = threadIdx.x; q < NT * 256; q += blockDim.x) red[q] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&red[t * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)], acc[t][r]);
  __syncthreads();
  for (int q = threadIdx.x; q < NT * 256; q += blockDim.x) {
    const int t = q >> 8, ii = (q >> 4) & 15, jj = q & 15;
    int bi = 0, bj = 0, tt = t;  // t -> (bi <= bj)
    for (bi = 0; bi < NBLK; ++bi) {
      if (tt < NBLK - bi) { bj = bi + tt; break; }
      tt -= NBLK - bi;
    }
    const int i = bi * 16 + ii, j = bj * 16 + jj;
    if (i <= j && j < C && debug != 3) atomicAdd(&scratch[int64_t(i) * C + j], red[q]);
  }
}

// Hub rows of the top layer.  A node's cost in seed_spmm_gram_kernel is its number of batch neighbours, walked by ONE wave
// (~2 us of issue-bound SIMD time per pair): on a power-law graph the largest hub (arxiv_powerlaw: 5 474 entries, ~330 in a
// batch) finishes alone long after the other waves (0.70 ms per batch against 0.29 ms on the uniform graph).  Rows with
// more than kTopSlice entries are therefore cut into slices that different waves take; a slice adds its partial tile to the
// hub's tile in global memory (float atomics) and a second, small launch finishes the summed tiles (planes + Gram).
// Per batch: one count / scan / fill pass over the active list builds the task triples.
__global__ void top_tasks_count_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ count,
                                       const int32_t* __restrict__ rowptr, int64_t N, int32_t* __restrict__ cnt) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= N) return;
  int32_t c = 0;
  if (k < *count) {
    const int32_t n = list[k];
    const int32_t deg = rowptr[n + 1] - rowptr[n];
    c = deg > kTopSlice ? (deg + kTopSlice - 1) / kTopSlice : 1;
  }
  cnt[k] = c;
}
__global__ void top_tasks_fill_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ count,
                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cnt,
                                      const int32_t* __restrict__ offs, int32_t* __restrict__ tasks,
                                      int32_t* __restrict__ task_count) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int32_t total = *count;
  if (k >= total) return;
  const int32_t n = list[k], c = cnt[k], o = offs[k];
  const int32_t rs = rowptr[n], re = rowptr[n + 1];
  if (c == 1) {
    tasks[3 * o] = n; tasks[3 * o + 1] = rs; tasks[3 * o + 2] = re;
  } else {
    for (int32_t i = 0; i < c; ++i) {
      tasks[3 * (o + i)] = -1 - n;
      tasks[3 * (o + i) + 1] = rs + i * kTopSlice;
      tasks[3 * (o + i) + 2] = min(rs + (i + 1) * kTopSlice, re);
    }
  }
  if (k == total - 1) *task_count = o + c;
}

template <int NBLK>
int seed_spmm_gram_launch(lgnn_ctx* h, bool fork_exact, float* g, int64_t cb, int64_t ce, float* scratch,
                          hipStream_t s) {
  const int C = int(h->dims[h->L]);
  constexpr int NT = NBLK * (NBLK + 1) / 2;
  const int rp = (int(ce - cb) + 3) & ~3;
  const int ldb = ((C + 7) & ~7) + 2;  // rows hold C rounded up to 8 columns; == 2 mod 4 keeps the LDS banks apart
  const size_t per_wave = size_t(rp) * ldb * 4;
  const int waves = int(std::max<size_t>(1, std::min<size_t>(16, (150 * 1024) / per_wave)));
  // + 64 floats: an MFMA operand read of the last row may run past it (those lanes are masked)
  const size_t smem = std::max(per_wave * waves + 256, size_t(NT) * 256 * 4);
  LGNN_REQUIRE(smem <= 158 * 1024, "too many classes for the seed SpMM kernel (use class ranges)");
  static bool attr_set = false;  // more than 64 KiB of dynamic LDS needs an explicit opt-in
  if (!attr_set) {
    LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&seed_spmm_gram_kernel<NBLK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    attr_set = true;
  }
  const int per_cu = int(std::max<size_t>(1, std::min<size_t>(2048 / (64 * waves), (158 * 1024) / smem)));
  int debug = 0;
#ifdef LGNN_DEV  // make DEV=1: ablation switches (1 no plane stores, 2 no MFMA, 3 no global atomics)
  if (const char* dbg = getenv("LGNN_SEED_DEBUG")) debug = atoi(dbg);
#endif
  const int fe = h->lik == LGNN_LIK_REGRESSION ? 2 : (fork_exact ? 1 : 0);
  LGNN_CALL(long_rows_ensure(h, s));
  static const bool no_slices = getenv("LGNN_TOP_NO_SLICES") != nullptr;  // dev: A/B of the sliced hubs
  if (h->n_top_multi <= 0 || no_slices) {
    hipLaunchKernelGGL(seed_spmm_gram_kernel<NBLK>, dim3(unsigned(256 * per_cu)), dim3(64 * waves), smem, s, h->PT.rowptr,
                       h->PT.col, h->PT.val, h->N, C, h->ws.pos.as<int32_t>(), h->ws.probs.as<float>(),
                       h->fc.out.as<float>(), h->ws.mult.as<int32_t>(), fe, g, h->ws.act_list.as<int32_t>(),
                       h->ws.act_count.as<int32_t>(), int(cb), int(ce), scratch, ldb, debug, 0,
                       static_cast<const int32_t*>(nullptr), static_cast<float*>(nullptr), static_cast<const uint8_t*>(nullptr));
    LGNN_HIP_CHECK(hipGetLastError());
    return 0;
  }
  // sliced hubs: task triples of this batch's active nodes, partial tiles, then the finishing launch over the sliced hubs
  const int64_t N = h->N, nrows = ce - cb;
  const int64_t cap = N + h->n_top_slices;
  LGNN_CALL(h->top_cnt.reserve(size_t(N) * 4));
  LGNN_CALL(h->top_offs.reserve(size_t(N) * 4));
  LGNN_CALL(h->top_tasks.reserve(size_t(cap) * 12));
  LGNN_CALL(h->top_task_count.reserve(64));
  LGNN_CALL(h->top_hub_tiles.reserve(size_t(h->n_long) * nrows * C * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(h->top_task_count.p, 0, 4, s));
  LGNN_HIP_CHECK(hipMemsetAsync(h->top_hub_tiles.p, 0, size_t(h->n_long) * nrows * C * 4, s));
  hipLaunchKernelGGL(top_tasks_count_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->ws.act_list.as<int32_t>(),
                     h->ws.act_count.as<int32_t>(), h->PT.rowptr, N, h->top_cnt.as<int32_t>());
  LGNN_CALL(exclusive_scan_i32(h->top_cnt.as<int32_t>(), h->top_offs.as<int32_t>(), N, h->ws.select_tmp, s));
  hipLaunchKernelGGL(top_tasks_fill_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->ws.act_list.as<int32_t>(),
                     h->ws.act_count.as<int32_t>(), h->PT.rowptr, h->top_cnt.as<int32_t>(), h->top_offs.as<int32_t>(),
                     h->top_tasks.as<int32_t>(), h->top_task_count.as<int32_t>());
  hipLaunchKernelGGL(seed_spmm_gram_kernel<NBLK>, dim3(unsigned(256 * per_cu)), dim3(64 * waves), smem, s, h->PT.rowptr,
                     h->PT.col, h->PT.val, h->N, C, h->ws.pos.as<int32_t>(), h->ws.probs.as<float>(),
                     h->fc.out.as<float>(), h->ws.mult.as<int32_t>(), fe, g, h->top_tasks.as<int32_t>(),
                     h->top_task_count.as<int32_t>(), int(cb), int(ce), scratch, ldb, debug, 1, h->long_slot.as<int32_t>(),
                     h->top_hub_tiles.as<float>(), h->ws.active.as<uint8_t>());
  // finishing launch: one wave per sliced hub (its count is a property of the graph: a constant in device memory)
  const unsigned fin_blocks = unsigned(std::min<int64_t>(cdiv(h->n_top_multi, waves), 256 * per_cu));
  hipLaunchKernelGGL(seed_spmm_gram_kernel<NBLK>, dim3(fin_blocks), dim3(64 * waves), smem, s, h->PT.rowptr, h->PT.col,
                     h->PT.val, h->N, C, h->ws.pos.as<int32_t>(), h->ws.probs.as<float>(), h->fc.out.as<float>(),
                     h->ws.mult.as<int32_t>(), fe, g, h->top_multi.as<int32_t>(), h->top_multi.as<int32_t>() + h->n_top_multi, int(cb), int(ce),
                     scratch, ldb, debug, 2, h->long_slot.as<int32_t>(), h->top_hub_tiles.as<float>(),
                     h->ws.active.as<uint8_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// values of P^T with the columns of all-zero source rows removed: the fused SpMM issues no load for them
__global__ void mask_values_kernel(const int32_t* __restrict__ col, const float* __restrict__ val, int64_t nnz,
                                   const uint8_t* __restrict__ active, const int32_t* __restrict__ pos,
                                   float* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < nnz; p += stride) {
    const int32_t j = col[p];
    const bool on = active ? active[j] != 0 : pos[j] != INT32_MAX;
    out[p] = on ? val[p] : 0.f;
  }
}

// GraphSAGE top layer: the last Linear's output is not propagated, so grad wrt its output is the
// scattered seed itself: G[c][idx[m]][k] = seeds[m][c][k] for first occurrences (planes pre-zeroed).
__global__ void scatter_seed_planes_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N, int64_t C,
                                           const int32_t* __restrict__ pos, const float* __restrict__ seeds,
                                           float* __restrict__ g, int64_t cb, int64_t ce) {
  const int64_t m = blockIdx.x;
  const int64_t n = idx[m];
  if (n < 0 || n >= N || pos[n] != m) return;
  const int64_t CC = C * C;
  for (int64_t q = cb * C + threadIdx.x; q < ce * C; q += blockDim.x) {
    const int64_t c = q / C, k = q - c * C;
    g[(c * N + n) * C + k] = seeds[m * CC + q];
  }
}

}  // namespace

int record_event(lgnn_ctx* h, hipStream_t s) {
  if (h->ev_used >= h->ev.size()) {
    hipEvent_t e;
    LGNN_HIP_CHECK(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  LGNN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used++], s));
  return 0;
}

// 256-wide fused kernel: rows of P^T with more than kLongRow stored entries are computed by whole workgroups first
// (longrows.hip) and handed over as finished rows; the list is built on the first call (one stream synchronisation).
static int attach_long_rows(lgnn_ctx* h, FusedArgs& a, hipStream_t s) {
  if (a.width <= 128) return 0;  // the narrower kernels keep their own row loop
  LGNN_CALL(long_rows_ensure(h, s));
  if (h->n_long <= 0) return 0;
  LGNN_CALL(launch_long_rows_spmm(h, a.val, a.in, a.in_ld, a.in_plane_stride, a.nplanes, a.width, s));
  a.long_slot = h->long_slot.as<int32_t>();
  a.hub = h->hub.as<float>();
  a.hub_plane_stride = h->n_long * a.width;
  a.n_long = h->n_long;
  return 0;
}

// Every path decision of kfac_accumulate as a pure function of the shapes (base pointers come from hipMalloc and
// are 256-byte aligned, so alignment follows from the widths).  One helper serves the launch loop, the workspace
// sizing (need_pong, cc_max) and lgnn_kfac_plan, so that the prediction cannot drift from what the loop does.
KfacPlan plan_kfac(int kind, int L, int64_t N, int64_t nnz, const int64_t* dims, int act, bool no_fuse,
                   int64_t ws_limit, bool no_paths) {
  KfacPlan p{};
  const int64_t C = dims[L];
  const bool gcn = kind == LGNN_KIND_GCN;
  p.seeds_on_the_fly = gcn && !no_fuse && C <= 64;
  const int64_t d_top = L > 1 ? dims[L - 1] : 0;
  auto hact_ld = [&](int l) { return gcn ? dims[l + 1] : 2 * dims[l + 1]; };  // row stride of h_{l+1} (context.hip)
  p.sage_compact = !gcn && L > 1 && !no_fuse && nnz > 0 && backgemm_supported(C, 2 * d_top, false) &&
                   (N + 1) * 2 * d_top * 4 < (int64_t(1) << 31) &&
                   fused_supported(d_top, 2 * d_top, (N + 1) * 2 * d_top, nullptr, N + 1) && hact_ld(L - 2) % 4 == 0;
  const bool row_active = gcn && !no_fuse && nnz > 0;  // flags of the non-zero top-layer gradient rows exist
  p.need_pong = L > 2 || no_fuse;
  p.paths = !no_fuse && !no_paths && (gcn ? p.seeds_on_the_fly : true) && paths_supported(kind, L, dims, act, nnz);
  if (p.paths) p.sage_compact = false;  // (GraphSAGE: one-hop paths instead of the compact top level + fused LIST kernel)
  for (int l = L - 1; l >= 1; --l) {
    const int64_t d = dims[l], dout = dims[l + 1];
    const bool top = l == L - 1;
    if (gcn) {
      p.fuse[l] = !no_fuse && fused_supported(d, d, N * d, nullptr, N + 1);
      p.backgemm[l] = top && p.fuse[l] && row_active && backgemm_supported(dout, d, act == LGNN_ACT_RELU) &&
                      (N + 1) * d * 4 < (int64_t(1) << 31);
    } else {
      const bool compact = p.sage_compact && top;
      const int64_t stride = (compact ? N + 1 : N) * 2 * d;
      p.fuse[l] = !no_fuse && fused_supported(d, 2 * d, stride, nullptr, N + 1) && hact_ld(l - 1) % 4 == 0;
      p.backgemm[l] = compact;
    }
    if (!p.fuse[l]) p.need_pong = true;  // the unfused path writes its SpMM output there
    p.maxw = std::max(p.maxw, gcn ? d : 2 * d);  // GEMM output width of layer l
  }
  if (p.paths) {  // Y [N][classes][H] is the only plane-sized buffer; no fused kernel, no backward GEMM, no second buffer
    p.need_pong = false;
    for (int l = 0; l < L; ++l) p.fuse[l] = p.backgemm[l] = false;
  }
  const int64_t per_class = p.paths ? N * p.maxw * 4 : (N + 1) * p.maxw * 4 * (p.need_pong ? 2 : 1);
  p.cc_max = std::max<int64_t>(1, std::min<int64_t>(C, ws_limit / std::max<int64_t>(per_class, 1)));
  return p;
}

// shared by kfac / diag / last layer: mark the batch, compute seeds/probs/loss.
int batch_prologue(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, bool want_seeds, bool fork_exact,
                   float* loss_out, hipStream_t s, const void* y_seed, float resid_scale) {
  const int64_t N = h->N, C = h->dims[h->L];
  // y_seed != null: single-column gradient seeds of the empirical / MC Fisher (labels resp. fp32 targets to seed with)
  const int seed_mode = y_seed ? (h->lik == LGNN_LIK_REGRESSION ? 4 : 3)
                               : (h->lik == LGNN_LIK_REGRESSION ? 2 : (fork_exact ? 1 : 0));
  LGNN_REQUIRE(2 * C * 4 <= 64 * 1024, "too many classes for the seed kernel");
  int* bad = h->ws.flags.as<int>();  // allocated and zeroed by lgnn_create; sticky until lgnn_check_async_errors
  hipLaunchKernelGGL(mark_batch_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, N,
                     h->ws.pos.as<int32_t>(), bad);
  LGNN_CALL(h->ws.probs.reserve(size_t(M) * C * 4));
  LGNN_CALL(h->ws.mult.reserve(size_t(M) * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(h->ws.mult.p, 0, size_t(M) * 4, s));
  float* seeds = nullptr;
  if (want_seeds) {
    LGNN_CALL(h->ws.seeds.reserve(size_t(M) * C * C * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(h->ws.seeds.p, 0, size_t(M) * C * C * 4, s));
    seeds = h->ws.seeds.as<float>();
  }
  LGNN_REQUIRE(8 * C * 4 <= 60 * 1024, "too many classes for the seed kernel");
  hipLaunchKernelGGL(seed_kernel, dim3(unsigned(std::min<int64_t>(cdiv(M, 4), 2048))), dim3(256), size_t(8 * C) * 4, s,
                     h->fc.out.as<float>(), C, idx,
                     y, M, N, h->ws.pos.as<int32_t>(), seed_mode, seeds,
                     h->ws.probs.as<float>(), loss_out, bad, h->ws.mult.as<int32_t>(), y_seed, resid_scale);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int batch_epilogue(lgnn_ctx* h, const int64_t* idx, int64_t M, hipStream_t s) {
  hipLaunchKernelGGL(unmark_batch_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, h->N,
                     h->ws.pos.as<int32_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// The GCN top layer of one batch for a caller outside the accumulate (adjgrad.hip): active rows (flags in ws.active, list in
// ws.act_list / ws.act_count) and the class-major planes g[c][n][:] of ALL C classes for those rows -- rows that are not
// active are NOT written.  Needs batch_prologue's probabilities / multiplicities; the Gram it also forms goes to a scratch.
int kfac_top_planes(lgnn_ctx* h, const int64_t* idx, int64_t M, bool fork_exact, float* g, hipStream_t s) {
  const int64_t N = h->N, C = h->dims[h->L];
  LGNN_REQUIRE(h->kind == LGNN_KIND_GCN, "internal: top-layer planes are a GCN step");
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  LGNN_HIP_CHECK(hipMemsetAsync(h->ws.active.p, 0, size_t(N), s));
  hipLaunchKernelGGL(mark_active_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr, h->P.col,
                     h->ws.active.as<uint8_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
  LGNN_CALL(h->ws.act_count.reserve(64));
  LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                          h->ws.select_tmp, s));
  const int l = h->L - 1;
  LGNN_CALL(h->ws.gram_scratch[l].reserve(size_t(C) * C * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(h->ws.gram_scratch[l].p, 0, size_t(C) * C * 4, s));
  float* sc = h->ws.gram_scratch[l].as<float>();
  switch (int(cdiv(C, 16))) {
    case 1: return seed_spmm_gram_launch<1>(h, fork_exact, g, 0, C, sc, s);
    case 2: return seed_spmm_gram_launch<2>(h, fork_exact, g, 0, C, sc, s);
    case 3: return seed_spmm_gram_launch<3>(h, fork_exact, g, 0, C, sc, s);
    default: return seed_spmm_gram_launch<4>(h, fork_exact, g, 0, C, sc, s);
  }
}

int kfac_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train, uint32_t flags,
                    int64_t cb, int64_t ce, float* const* A_out, float* const* B_out, float* loss_out, hipStream_t s,
                    const KfacFisherOpts* fisher, const KfacShare* share) {
  LGNN_REQUIRE(M > 0 && idx && (y || (fisher && !fisher->add_loss_and_A)), "empty batch or null batch pointers");
  LGNN_REQUIRE(h->L > 0, "no model bound");
  // Share mode (lgnn_kfac_accumulate_share): parts [begin, end) of `count` equal parts of the batch's work; HOW a batch is
  // cut is decided below, once the route is known -- destination-node ranges on the path routes (B_0 = sum_n Y_n^T Y_n:
  // nothing is computed twice), class ranges everywhere else.  Until then the range is "everything".
  LGNN_REQUIRE(!share || (share->count > 0 && share->begin >= 0 && share->begin < share->end && share->end <= share->count && !fisher),
               "share must satisfy 0 <= begin < end <= count");
  if (share) { cb = 0; ce = h->dims[h->L]; }
  LGNN_REQUIRE(cb >= 0 && cb < ce && ce <= h->dims[h->L], "class range must satisfy 0 <= begin < end <= C");
  // B_l = sum over class columns c of g_c^T g_c: a class range is an exact additive share of the batch.
  // The share that contains class 0 also carries what exists once per batch: the loss and the A increment
  // (and, for GraphSAGE, the whole top-layer Gram, which is only M*C rows).
  const bool first = share ? share->begin == 0 : cb == 0;
  // empirical / MC Fisher: one plane (the gradient seed sits in column 0 of the block), B scaled by 1 / mc_samples, the
  // loss of the TRUE labels and the A increment only with the call that is told to add them
  const bool once = fisher ? fisher->add_loss_and_A : first;
  const float b_scale = fisher ? fisher->b_scale : 1.0f;
  LGNN_REQUIRE(!fisher || (cb == 0 && ce == 1 && fisher->y_seed), "internal: Fisher seeds use plane 0 only");
  LGNN_REQUIRE(n_train > 0, "n_train must be positive");
  LGNN_REQUIRE(A_out && B_out && loss_out, "null output pointers");
  // the input Grams feed the A increment, which only the share with class 0 adds (ranks of a multi-GPU job that
  // hold no such share never compute them)
  if (once) LGNN_CALL(forward_ensure_grams(h, s));
  else LGNN_CALL(forward_ensure(h, s));
  const int64_t N = h->N;
  const int L = h->L;
  const int64_t C = h->dims[L];
  const int64_t CC = C * C;
  // models with res / norm (gnn/models/base_gnn.py:141-149) take the unfused route: GEMM, row-local norm backward,
  // SpMM^T and Gram as separate kernels (resnorm.hip)
  const bool no_fuse = (flags & LGNN_FLAG_NO_FUSE) != 0 || h->extras();
  LGNN_REQUIRE(M < INT32_MAX, "batch too large");

  const bool fork_exact = (flags & LGNN_FLAG_FORK_EXACT_SEED) != 0;
  // the two-hop path route (paths.hip) where the shape allows it and the batch's expected number of paths makes it pay
  bool no_paths = (flags & LGNN_FLAG_NO_PATHS) != 0 || fisher != nullptr || no_fuse ||
                  !paths_supported(h->kind, L, h->dims, h->act, h->nnz);
  if (!no_paths && h->kind == LGNN_KIND_GCN) {  // (GraphSAGE's paths are one hop: at most nnz + (C + 1) M of them, always cheaper)
    LGNN_CALL(two_hop_ensure(h, s));
    no_paths = !paths_pay(h, M) && (flags & LGNN_FLAG_FORCE_PATHS) == 0;
  }
  const KfacPlan plan = plan_kfac(h->kind, L, N, h->nnz, h->dims, h->act, no_fuse, h->ws_limit, no_paths);
  // share mode: the cut.  Path routes: nodes [nb, ne), all classes (the top layer's small Gram goes with part 0); otherwise
  // classes [C begin / count, C end / count) -- possibly none, then only what exists once per batch is left to do.
  int64_t nb = 0, ne = N;
  const bool share_nodes = share && plan.paths;
  if (share_nodes) {
    nb = N * share->begin / share->count;
    ne = N * share->end / share->count;
  } else if (share) {
    cb = C * share->begin / share->count;
    ce = C * share->end / share->count;
  }
  const bool no_classes = cb >= ce;  // (share mode with more parts than classes)
  // GCN, fused path: the top-layer kernel rebuilds each sample's C x C seed block from its probabilities and logits,
  // so the blocks are never written (64 MB per arxiv-shaped batch); every other path reads them from ws.seeds
  const bool seeds_on_the_fly = plan.seeds_on_the_fly && !fisher;  // the on-the-fly rebuild knows the GGN blocks only
  LGNN_CALL(batch_prologue(h, idx, y, M, !seeds_on_the_fly, fork_exact, once ? loss_out : nullptr, s,
                           fisher ? fisher->y_seed : nullptr, fisher ? fisher->resid_scale : 1.0f));

  // A_l += in_l^T in_l / n_train   (kfac.py:870 divides by M, curvlinops.py:46-53 multiplies by M/N)
  for (int l = 0; once && l < L; ++l)
    LGNN_CALL(launch_sym_accumulate(h->fc.gram_raw[l].as<float>(), h->in_dim[l], 1.0f / float(n_train), A_out[l], s));
  // res.{l} sees h_l itself: its input covariance is the conv's (GCN) resp. the leading block of cat_l's (GraphSAGE)
  for (int l = 0; once && h->has_res && l < L - 1; ++l)
    LGNN_CALL(launch_sym_accumulate(h->fc.gram_raw[l].as<float>(), h->dims[l], 1.0f / float(n_train), A_out[L + l], s,
                                    h->in_dim[l]));

  for (int l = 0; l < L; ++l) {
    const int64_t D = h->dims[l + 1];
    LGNN_CALL(h->ws.gram_scratch[l].reserve(size_t(D) * D * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(h->ws.gram_scratch[l].p, 0, size_t(D) * D * 4, s));
    if (h->has_res && h->kind == LGNN_KIND_GCN && l < L - 1) {
      LGNN_CALL(h->ws.gram_scratch_res[l].reserve(size_t(D) * D * 4));
      LGNN_HIP_CHECK(hipMemsetAsync(h->ws.gram_scratch_res[l].p, 0, size_t(D) * D * 4, s));
    }
  }

  // a part without a class column: what exists once per batch has been added above.  (GraphSAGE goes on: its whole
  // top-layer Gram belongs to part 0, and every loop over the classes [cb, ce) below is empty)
  if (no_classes && h->kind == LGNN_KIND_GCN) {
    LGNN_CALL(batch_epilogue(h, idx, M, s));
    return 0;
  }

  // ---- top layer ---------------------------------------------------------------------------------
  float* gtop = nullptr;  // planes [C][N][C]
  if ((h->kind == LGNN_KIND_GCN || L > 1) && !(h->kind == LGNN_KIND_SAGE && plan.paths && !fisher)) {
    LGNN_CALL(h->ws.top.reserve(size_t(N) * CC * 4 + 16));  // + 16: the backward GEMM reads rows shifted by one float
    gtop = h->ws.top.as<float>();
  }
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  bool have_act_list = false;
  // GraphSAGE, fused path: the top-level GEMM g W_l runs over the (distinct) batch nodes only -- 6 % of the rows at
  // the arxiv shape -- through the compacted-row backward GEMM; the other rows of its output stay zero (see below)
  const bool sage_compact = plan.sage_compact;
  if (h->kind == LGNN_KIND_GCN) {
    const int64_t nq = (ce - cb) * C;
    if (seeds_on_the_fly) {
      // active rows first (columns of the batch nodes' P rows), then one pass over those rows only
      LGNN_HIP_CHECK(hipMemsetAsync(h->ws.active.p, 0, size_t(N), s));
      hipLaunchKernelGGL(mark_active_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr,
                         h->P.col, h->ws.active.as<uint8_t>());
      LGNN_HIP_CHECK(hipGetLastError());
      LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
      LGNN_CALL(h->ws.act_count.reserve(64));
      LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(),
                              h->ws.act_count.as<int32_t>(), h->ws.select_tmp, s));
      have_act_list = true;
      // a single-layer model needs the Gram only -- and so does the two-hop path route (paths.hip), which never reads planes
      const bool paths_route = plan.paths && !fisher;
      float* gplanes = (L > 1 && !paths_route) ? gtop : nullptr;
      const bool top_here = !share_nodes || first;  // node shares: B_{L-1} (all classes, 0.24 ms) goes with part 0
      if (L > 1 && !plan.fuse[L - 1] && !paths_route)
        // the unfused lower path reads every row of the planes: the rows this kernel skips must be zero
        LGNN_HIP_CHECK(hipMemsetAsync(gtop + cb * N * C, 0, size_t(N) * nq * 4, s));
      float* sc = h->ws.gram_scratch[L - 1].as<float>();
      if (top_here) switch (int(cdiv(C, 16))) {
        case 1: LGNN_CALL(seed_spmm_gram_launch<1>(h, fork_exact, gplanes, cb, ce, sc, s)); break;
        case 2: LGNN_CALL(seed_spmm_gram_launch<2>(h, fork_exact, gplanes, cb, ce, sc, s)); break;
        case 3: LGNN_CALL(seed_spmm_gram_launch<3>(h, fork_exact, gplanes, cb, ce, sc, s)); break;
        default: LGNN_CALL(seed_spmm_gram_launch<4>(h, fork_exact, gplanes, cb, ce, sc, s)); break;
      }
    } else {
      int waves = int(std::max<int64_t>(1, std::min<int64_t>(4, (48 * 1024) / (nq * 4))));
      LGNN_REQUIRE(nq * 4 <= 60 * 1024, "too many classes for the seed SpMM kernel (use class ranges)");
      hipLaunchKernelGGL(seed_spmm_kernel, dim3(unsigned(cdiv(N, waves))), dim3(64 * waves), size_t(waves) * nq * 4, s,
                         h->PT.rowptr, h->PT.col, h->PT.val, N, C, h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(),
                         gtop, h->ws.active.as<uint8_t>(), cb, ce);
      LGNN_HIP_CHECK(hipGetLastError());
      LGNN_CALL(launch_gram(gtop + cb * N * C, C, (ce - cb) * N, C, h->ws.gram_scratch[L - 1].as<float>(), s));
    }
  } else {
    // rows (m, c) of the accumulated seeds; rows of non-first duplicates are zero
    if (first) LGNN_CALL(launch_gram(h->ws.seeds.as<float>(), C, M * C, C, h->ws.gram_scratch[L - 1].as<float>(), s));
    if (L > 1 && !(plan.paths && !fisher)) {  // (the path route reads the samples' probabilities, not seed planes)
      // the compact top level reads the planes at the batch nodes only; every other path reads all rows
      if (!sage_compact) LGNN_HIP_CHECK(hipMemsetAsync(gtop + cb * N * C, 0, size_t(N) * (ce - cb) * C * 4, s));
      hipLaunchKernelGGL(scatter_seed_planes_kernel, dim3(unsigned(M)), dim3(256), 0, s, idx, M, N, C,
                         h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), gtop, cb, ce);
      LGNN_HIP_CHECK(hipGetLastError());
    }
  }

  // ---- 2-layer GCN: B_0 from the batch's 2-hop paths -- no class planes (paths.hip) -------------------------
  const bool paths_route = plan.paths && !fisher && (h->kind == LGNN_KIND_GCN ? seeds_on_the_fly : true);
  h->last_route_paths = paths_route;
  if (paths_route) {
    const int mode = h->lik == LGNN_LIK_REGRESSION ? 2 : (fork_exact ? 1 : 0);
    if (h->kind == LGNN_KIND_GCN)
      LGNN_CALL(kfac_paths_first_layer(h, idx, M, mode, cb, ce, h->ws.gram_scratch[0].as<float>(), s, nb, ne));
    else LGNN_CALL(kfac_paths_first_layer_sage(h, idx, M, mode, cb, ce, h->ws.gram_scratch[0].as<float>(), s, nb, ne));
  }

  // ---- lower layers, chunked over classes ----------------------------------------------------------
  if (L > 1 && !paths_route) {
    // Source rows of the first backward plane set are non-zero only where the top-layer gradient is:
    // GCN: nodes with a batch node among their P^T neighbours (flags from the seed SpMM);
    // GraphSAGE: the batch nodes themselves.  Zeroed values make the fused SpMM skip those gathers.
    const float* val_top = h->PT.val;
    const uint8_t* row_active = nullptr;
    if (!no_fuse && h->nnz > 0) {
      LGNN_CALL(h->ws.val_act.reserve(size_t(h->nnz) * 4));
      const bool gcn = h->kind == LGNN_KIND_GCN;
      hipLaunchKernelGGL(mask_values_kernel, dim3(unsigned(std::min<int64_t>(cdiv(h->nnz, 256), 4096))), dim3(256), 0, s,
                         h->PT.col, h->PT.val, h->nnz, gcn ? h->ws.active.as<uint8_t>() : (const uint8_t*)nullptr,
                         h->ws.pos.as<int32_t>(), h->ws.val_act.as<float>());
      LGNN_HIP_CHECK(hipGetLastError());
      val_top = h->ws.val_act.as<float>();
      if (sage_compact) {
        LGNN_HIP_CHECK(hipMemsetAsync(h->ws.active.p, 0, size_t(N), s));
        hipLaunchKernelGGL(mark_batch_flags_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, N,
                           h->ws.active.as<uint8_t>());
        LGNN_HIP_CHECK(hipGetLastError());
        LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
        LGNN_CALL(h->ws.act_count.reserve(64));
        LGNN_CALL(compact_flags(h->ws.active.as<uint8_t>(), N, h->ws.act_list.as<int32_t>(),
                                h->ws.act_count.as<int32_t>(), h->ws.select_tmp, s));
      }
      if (gcn) {
        row_active = h->ws.active.as<uint8_t>();
        if (!have_act_list) {
          LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
          LGNN_CALL(h->ws.act_count.reserve(64));
          LGNN_CALL(compact_flags(row_active, N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                                  h->ws.select_tmp, s));
        }
      }
    }
    // The pong buffer holds g_{l-1} when a lower layer needs it (L > 2) and the SpMM output of the unfused path;
    // the fused two-layer path never touches it (plan_kfac decides, from the same per-layer flags the loop uses).
    const int64_t maxw = plan.maxw;
    const bool need_pong = plan.need_pong;
    const int64_t cc_max = plan.cc_max;
    for (int l = 1; l < L; ++l)
      LGNN_REQUIRE(h->fc.hact_ld[l - 1] == (h->kind == LGNN_KIND_GCN ? h->dims[l] : 2 * h->dims[l]),
                   "internal: activation row stride differs from the plan's");
    LGNN_CALL(h->ws.planes_a.reserve(size_t(cc_max) * (N + 1) * maxw * 4));  // + 1: the backward GEMM's spare row per plane
    if (need_pong) LGNN_CALL(h->ws.planes_b.reserve(size_t(cc_max) * N * maxw * 4));
    const bool gcn_res_deep = h->has_res && h->kind == LGNN_KIND_GCN && L > 2;  // dh_l = g_l W_l + u_l Wr_l below the top level
    if (gcn_res_deep) LGNN_CALL(h->ws.planes_c.reserve(size_t(cc_max) * N * maxw * 4));
    for (int64_t c0 = cb; c0 < ce; c0 += cc_max) {
      const int64_t cc = std::min(cc_max, ce - c0);
      const float* g = gtop + c0 * N * C;  // planes [cc][N][dims[l+1]] of layer l
      float* ping = h->ws.planes_a.as<float>();
      float* pong = need_pong ? h->ws.planes_b.as<float>() : nullptr;
      for (int l = L - 1; l >= 1; --l) {
        const int64_t dout = h->dims[l + 1];  // width of g
        const int64_t d = h->dims[l];         // width of g_{l-1}
        const bool store = (l - 1) > 0;
        float* scratch = h->ws.gram_scratch[l - 1].as<float>();
        const bool dominant = (l - 1) == 0;
        if (h->kind == LGNN_KIND_GCN) {
          // up = act'(h_l) * (g W_l)     (gnn/models/layers.py:45-46 backward through lin and the activation)
          GemmEpilogue ep;
          ep.hact = h->fc.hact_p[l - 1]; ep.hact_ld = h->fc.hact_ld[l - 1]; ep.act = h->act; ep.hact_row_mod = N;
          const bool top_level = l == L - 1;
          const bool fuse_here = plan.fuse[l];
          if (top_level && fuse_here) ep.row_active = row_active;  // inactive rows are never read below
          int64_t ping_stride = N * d;
          // res / norm (unfused route only): below the top level the layer above has a res Linear whose gradient u_l (still
          // in `ping` from the previous step) joins in, dh_l = g_l W_l + u_l Wr_l; then the activation mask and the norm's
          // row-local backward turn dh_l into u_{l-1}, the gradient at s_{l-1} -- what res.{l-1} sees and what P^T propagates
          const bool res_term = h->has_res && !top_level;
          if (res_term) {
            GemmEpilogue none;
            LGNN_CALL(launch_gemm(ping, dout, h->Wr[l], d, h->ws.planes_c.as<float>(), d, cc * N, dout, d, none, s));
            ep = none;  // the mask moves behind the sum
          }
          // (32-bit row offsets inside a plane: beyond 2 GiB per plane the generic GEMM takes over)
          if (plan.backgemm[l]) {
            LGNN_REQUIRE(row_active != nullptr, "internal: compacted backward GEMM without its row list");
            ping_stride = (N + 1) * d;  // row N of every plane takes the stores of rows past the end of the list
            BackGemmArgs bg{};
            bg.u_plane_stride = ping_stride;
            bg.G = g; bg.W = h->W[l]; bg.ldw = d; bg.U = ping; bg.N = N; bg.K = dout; bg.Nout = d; bg.planes = cc;
            bg.rows = h->ws.act_list.as<int32_t>(); bg.na_dev = h->ws.act_count.as<int32_t>();
            if (h->act == LGNN_ACT_RELU) { bg.mask_bits = h->fc.mask_bits[l - 1].as<uint32_t>(); bg.mask_words = cdiv(d, 32); }
            else { bg.hact = ep.hact; bg.hact_ld = ep.hact_ld; bg.act = h->act; }
            LGNN_CALL(launch_backgemm(bg, s));
          } else {
            LGNN_CALL(launch_gemm(g, dout, h->W[l], d, ping, d, cc * N, dout, d, ep, s));
          }
          if (res_term || h->norm != LGNN_NORM_NONE)
            LGNN_CALL(launch_resnorm_backward(h, l - 1, ping, d, cc * N, res_term ? h->ws.planes_c.as<float>() : nullptr,
                                              res_term, s));
          if (h->has_res)  // B of res.{l-1}: the Gram of u_{l-1} before the propagation (curvlinops/kfac.py:777-817)
            LGNN_CALL(launch_gram(ping, d, cc * N, d, h->ws.gram_scratch_res[l - 1].as<float>(), s));
          FusedArgs a{};
          a.rowptr = h->PT.rowptr; a.col = h->PT.col; a.val = (top_level && fuse_here) ? val_top : h->PT.val;
          a.nrows = N; a.nplanes = cc;
          a.in = ping; a.in_ld = d; a.in_plane_stride = ping_stride;
          LGNN_REQUIRE(!store || pong != nullptr, "internal: stored planes without a buffer");
          a.store = store ? pong : nullptr; a.store_ld = d; a.store_plane_stride = N * d;
          a.width = d; a.scratch = scratch;
          if (fuse_here) {
            LGNN_CALL(attach_long_rows(h, a, s));
            if (h->timing && dominant) LGNN_CALL(record_event(h, s));
            LGNN_CALL(launch_spmm_gram_ex(a, s));
            if (h->timing && dominant) { LGNN_CALL(record_event(h, s)); h->ev_planes += cc; }
          } else {
            LGNN_REQUIRE(pong != nullptr, "internal: unfused path without its output planes");
            SpmmArgs sa{};
            sa.rowptr = a.rowptr; sa.col = a.col; sa.val = a.val; sa.nrows = N;
            sa.skip_zero = a.val != h->PT.val;  // the per-batch values: zero at the columns of all-zero source rows
            sa.in = ping; sa.in_ld = d; sa.in_plane_stride = N * d;
            sa.out = pong; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
            LGNN_CALL(launch_spmm_ex(sa, cc, s));
            LGNN_CALL(launch_gram(pong, d, cc * N, d, scratch, s));
          }
        } else {
          // dcat = g W_l  [cc*N, 2d];  g_{l-1} = act'(h_l) * (dcat[:, :d] + P^T dcat[:, d:])
          const bool compact = sage_compact && l == L - 1;
          int64_t ping_stride = N * 2 * d;
          if (compact) {
            // rows of dcat other than the batch nodes' are zero: the buffer is zeroed once (per allocation / extent)
            // and the rows written here are cleared again after the fused kernel has consumed them
            ping_stride = (N + 1) * 2 * d;
            const size_t extent = size_t(cc) * ping_stride * 4;
            if (h->ws.planes_a_zero_ptr != h->ws.planes_a.p || h->ws.planes_a_zero_bytes < extent) {
              LGNN_HIP_CHECK(hipMemsetAsync(h->ws.planes_a.p, 0, extent, s));
              h->ws.planes_a_zero_ptr = h->ws.planes_a.p;
              h->ws.planes_a_zero_bytes = extent;
            }
            BackGemmArgs bg{};
            bg.G = g; bg.W = h->W[l]; bg.ldw = 2 * d; bg.U = ping; bg.N = N; bg.K = dout; bg.Nout = 2 * d; bg.planes = cc;
            bg.u_plane_stride = ping_stride;
            bg.rows = h->ws.act_list.as<int32_t>(); bg.na_dev = h->ws.act_count.as<int32_t>();
            LGNN_CALL(launch_backgemm(bg, s));
          } else {
            // second backward level of a deeper model: g (the level above's output) is non-zero only where that level could
            // write, i.e. on the batch nodes and their neighbours (products shape: 20 % of the rows).  Once per batch: the
            // flags and the list of those rows and P^T's values with the other columns zeroed (their rows are not gathered
            // below); the list's length stays on the device.
            const bool second = l == L - 2 && !no_fuse && h->nnz > 0;
            if (second && c0 == cb) {
              LGNN_CALL(h->ws.out_flags.reserve(size_t(N)));
              LGNN_CALL(h->ws.out_list.reserve(size_t(N) * 4));
              LGNN_CALL(h->ws.out_count.reserve(64));
              LGNN_CALL(h->ws.val_act2.reserve(size_t(h->nnz) * 4));
              LGNN_HIP_CHECK(hipMemsetAsync(h->ws.out_flags.p, 0, size_t(N), s));
              hipLaunchKernelGGL(mark_active_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr,
                                 h->P.col, h->ws.out_flags.as<uint8_t>());
              hipLaunchKernelGGL(mark_batch_flags_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, N,
                                 h->ws.out_flags.as<uint8_t>());
              hipLaunchKernelGGL(mask_values_kernel, dim3(unsigned(std::min<int64_t>(cdiv(h->nnz, 256), 4096))), dim3(256), 0, s,
                                 h->PT.col, h->PT.val, h->nnz, h->ws.out_flags.as<uint8_t>(), h->ws.pos.as<int32_t>(),
                                 h->ws.val_act2.as<float>());
              LGNN_HIP_CHECK(hipGetLastError());
              LGNN_CALL(compact_flags(h->ws.out_flags.as<uint8_t>(), N, h->ws.out_list.as<int32_t>(),
                                      h->ws.out_count.as<int32_t>(), h->ws.select_tmp, s));
            }
            GemmEpilogue ep;
            if (second) {
              // g W_l on the listed rows only, straight from and into the planes (the kernel gathers the A rows and scatters
              // the C rows through the list and reads the list's length on the device: no host round trip); the other rows
              // of dcat are zero
              LGNN_HIP_CHECK(hipMemsetAsync(ping, 0, size_t(cc) * N * 2 * d * 4, s));
              LGNN_CALL(launch_gemm_listed(g, dout, h->Wback(l), 2 * d, ping, 2 * d, cc, N, h->ws.out_list.as<int32_t>(),
                                           h->ws.out_count.as<int32_t>(), dout, 2 * d, ep, s));
            } else {
              // (with res: W_l + [Wr_l | 0] -- the Linear's output gradient is also res.{l}'s, base_gnn.py:141-144)
              LGNN_CALL(launch_gemm(g, dout, h->Wback(l), 2 * d, ping, 2 * d, cc * N, dout, 2 * d, ep, s));
            }
            h->ws.planes_a_zero_ptr = nullptr;  // every row written
          }
          FusedArgs a{};
          a.rowptr = h->PT.rowptr; a.col = h->PT.col; a.val = (l == L - 1) ? val_top : h->PT.val;
          if (l == L - 2 && !no_fuse && h->nnz > 0) a.val = h->ws.val_act2.as<float>();
          a.nrows = N; a.nplanes = cc;
          a.in = ping + d; a.in_ld = 2 * d; a.in_plane_stride = ping_stride;
          a.self = ping; a.self_ld = 2 * d; a.self_plane_stride = ping_stride;
          a.hact = h->fc.hact_p[l - 1]; a.hact_ld = h->fc.hact_ld[l - 1]; a.act = h->act;
          if (compact && h->act == LGNN_ACT_RELU && d > 128) {
            // 256-wide kernel, nothing conditional in its row loop: self rows exist for the batch nodes only (flags),
            // the ReLU derivative comes from the bit masks of the forward
            a.self_rows = h->ws.active.as<uint8_t>();
            a.mask_bits = h->fc.mask_bits[l - 1].as<uint32_t>();
            a.mask_words = int(cdiv(d, 32));
            static const bool no_list = getenv("LGNN_SAGE_NO_ROW_LIST") != nullptr;  // dev: A/B of the row list
            if (!store && plan.fuse[l] && !no_list) {
              // g_{l-1} = act' * (dcat_self + P^T dcat_neigh) can be non-zero on the batch nodes and their neighbours only
              // (the columns of the batch nodes' P rows): the fused kernel visits just those rows
              LGNN_CALL(h->ws.out_flags.reserve(size_t(N)));
              LGNN_CALL(h->ws.out_list.reserve(size_t(N) * 4));
              LGNN_CALL(h->ws.out_count.reserve(64));
              LGNN_HIP_CHECK(hipMemsetAsync(h->ws.out_flags.p, 0, size_t(N), s));
              hipLaunchKernelGGL(mark_active_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, h->P.rowptr,
                                 h->P.col, h->ws.out_flags.as<uint8_t>());
              hipLaunchKernelGGL(mark_batch_flags_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, N,
                                 h->ws.out_flags.as<uint8_t>());
              LGNN_HIP_CHECK(hipGetLastError());
              LGNN_CALL(compact_flags(h->ws.out_flags.as<uint8_t>(), N, h->ws.out_list.as<int32_t>(),
                                      h->ws.out_count.as<int32_t>(), h->ws.select_tmp, s));
              a.row_list = h->ws.out_list.as<int32_t>();
              a.row_count = h->ws.out_count.as<int32_t>();
            }
          }
          LGNN_REQUIRE(!store || pong != nullptr, "internal: stored planes without a buffer");
          a.store = store ? pong : nullptr; a.store_ld = d; a.store_plane_stride = N * d;
          a.width = d; a.scratch = scratch;
          if (plan.fuse[l]) {
            LGNN_CALL(attach_long_rows(h, a, s));
            if (h->timing && dominant) LGNN_CALL(record_event(h, s));
            LGNN_CALL(launch_spmm_gram_ex(a, s));
            if (h->timing && dominant) { LGNN_CALL(record_event(h, s)); h->ev_planes += cc; }
            if (compact) {
              hipLaunchKernelGGL(clear_rows_kernel, dim3(2048), dim3(256), 0, s, ping, ping_stride, 2 * d, cc,
                                 h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>());
              LGNN_HIP_CHECK(hipGetLastError());
            }
          } else {
            LGNN_REQUIRE(!compact, "internal: compact GraphSAGE top level without the fused kernel");
            LGNN_REQUIRE(pong != nullptr, "internal: unfused path without its output planes");
            SpmmArgs sa{};
            sa.rowptr = a.rowptr; sa.col = a.col; sa.val = a.val; sa.nrows = N;
            sa.skip_zero = a.val != h->PT.val;  // the per-batch values: zero at the columns of all-zero source rows
            sa.in = a.in; sa.in_ld = a.in_ld; sa.in_plane_stride = a.in_plane_stride;
            sa.self = a.self; sa.self_ld = a.self_ld; sa.self_plane_stride = a.self_plane_stride;
            sa.hact = a.hact; sa.hact_ld = a.hact_ld; sa.act = a.act;
            sa.out = pong; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
            LGNN_CALL(launch_spmm_ex(sa, cc, s));
            if (h->norm != LGNN_NORM_NONE) LGNN_CALL(launch_resnorm_backward(h, l - 1, pong, d, cc * N, nullptr, false, s));
            LGNN_CALL(launch_gram(pong, d, cc * N, d, scratch, s));
          }
        }
        // stream order makes the two buffers reusable: the next GEMM reads g (= pong) and overwrites
        // ping, which the SpMM above has finished reading; the next SpMM then overwrites pong
        g = pong;
      }
    }
  }

  for (int l = 0; l < L; ++l)
    LGNN_CALL(launch_sym_accumulate(h->ws.gram_scratch[l].as<float>(), h->dims[l + 1], b_scale, B_out[l], s));
  // res.{l}: GCN -- the Gram of u_l taken before the propagation; GraphSAGE -- the Linear's output is not propagated, so
  // convs.{l}.lin and res.{l} see the same gradient and share B
  for (int l = 0; h->has_res && l < L - 1; ++l)
    LGNN_CALL(launch_sym_accumulate((h->kind == LGNN_KIND_GCN ? h->ws.gram_scratch_res[l] : h->ws.gram_scratch[l]).as<float>(),
                                    h->dims[l + 1], b_scale, B_out[L + l], s));
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

}  // namespace lgnn

// Host-only: which kernels a KFAC accumulate of this shape would run (no context, no device work) -- lets the path
// decisions be tested per branch without allocating the planes (products-shaped GraphSAGE: 5 GB per plane).
extern "C" int lgnn_kfac_plan(int kind, int num_layers, const int64_t* dims, int64_t num_nodes, int64_t nnz, int activation,
                              uint32_t flags, int64_t workspace_limit, int64_t* out) {
  using namespace lgnn;
  if (!dims || !out) { set_error("null argument"); return 2; }
  LGNN_REQUIRE(num_layers >= 1 && num_layers <= kMaxLayers, "num_layers out of range");
  LGNN_REQUIRE(kind == LGNN_KIND_GCN || kind == LGNN_KIND_SAGE, "unknown graph kind");
  LGNN_REQUIRE(workspace_limit > 0 && num_nodes > 0, "workspace limit and node count must be positive");
  const KfacPlan p = plan_kfac(kind, num_layers, num_nodes, nnz, dims, activation, (flags & LGNN_FLAG_NO_FUSE) != 0,
                               workspace_limit, (flags & LGNN_FLAG_NO_PATHS) != 0);
  out[0] = p.seeds_on_the_fly; out[1] = p.sage_compact; out[2] = p.need_pong; out[3] = p.cc_max;
  for (int l = 0; l < num_layers; ++l)
    out[4 + l] = (p.fuse[l] ? 1 : 0) | (p.backgemm[l] ? 2 : 0) | ((p.paths && l == num_layers - 1) ? 4 : 0);
  return 0;
}

// Empirical / Monte-Carlo Fisher KFAC: one backward pass per call, seeded with resid_scale * d loss(f, y_seed) / d f.
extern "C" int lgnn_kfac_accumulate_fisher(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss,
                                           int64_t M, int64_t n_train, uint32_t flags, float resid_scale, float b_scale,
                                           float* const* A_out, float* const* B_out, float* loss_out, void* stream) {
  using namespace lgnn;
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(y_seed != nullptr, "the Fisher accumulate needs the labels / targets to seed with");
  KfacFisherOpts o{y_seed, resid_scale, b_scale, y_loss != nullptr};
  return kfac_accumulate(h, idx, y_loss, M, n_train, flags, 0, 1, A_out, B_out, loss_out, static_cast<hipStream_t>(stream), &o);
}
