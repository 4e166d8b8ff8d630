// Diagonal GGN and last-layer full GGN of one mini-batch, without materialising Jacobians.
//
// Reference: GGNInterface.diag (laplace/curvature/curvature.py:412-432) builds Js[M, C, P] with
// torch.func.jacrev (:89-130; 3.4 GB for the Cora-shaped config) and contracts
// einsum('bcp,bck,bkp->p', Js, Lambda, Js), Lambda_n = diag(p_n) - p_n p_n^T (:365-372).
// For the model family of the path the per-sample Jacobian is closed form:
//
//   last layer (any L):  J_n wrt W_{L-1} = I_C (x) phi_n, wrt b = s_n I_C
//        diag(W)[k,i] = sum_n Lambda_n[k,k] phi_n[i]^2 ,  diag(b)[k] = sum_n Lambda_n[k,k] s_n^2
//        GCN : phi_n = (P h)[n], s_n = rowsum(P)[n]      GraphSAGE: phi_n = cat[n], s_n = 1
//   first layer of a 2-layer model: with E[v] = [in-features the first Linear multiplies | bias column]
//        GCN : E[v] = [(P X)[v] | rowsum(P)[v]],   GraphSAGE: E[v] = [cat_0[v] | 1]
//        T[n,j,i] = sum_v P[n,v] act'(h_1[v,j]) E[v,i]     S[n,j,i] = act'(h_1[n,j]) E[n,i]  (GraphSAGE self path)
//        diag(W_0)[j,i] = sum_n qbb[n,j] T^2 + 2 qab[n,j] S T + qaa[n,j] S^2
//        q**[n,j] = w*_j^T Lambda_n w*_j with w_j the j-th column of the self / neighbour half of W_1.
// The reference cannot construct models with more than 2 layers (live breakpoint at
// gnn/models/base_gnn.py:109), so L <= 2 covers everything it can run.
//
// Last-layer full GGN (laplace/curvature/curvature.py:132-167, 374-410):
//   H = sum_n Lambda_n (x) phi~ phi~^T = blockdiag_c( sum_n p_nc phi~ phi~^T ) - Z^T Z,  z_n = p_n (x) phi~_n,
// i.e. one large fp32 MFMA Gram (2 M P^2 flops) plus C small weighted Grams.
#include "device_utils.h"
#include "lgnn_internal.h"

namespace lgnn {

namespace {

struct FeatView {  // E[v, i]: i < width -> base[v*ld + i]; i == width -> bias column
  const float* base;
  int64_t ld;
  int64_t width;
  const float* bias_col;  // per-node scale (rowsum) or nullptr for the constant 1
  int64_t nrows;          // node ids outside [0, nrows) (flagged by the batch prologue) read as zero rows
};

__device__ __forceinline__ float feat(const FeatView& f, int64_t v, int64_t i) {
  if (v < 0 || v >= f.nrows) return 0.f;
  if (i < f.width) return f.base[v * f.ld + i];
  return f.bias_col ? f.bias_col[v] : 1.f;
}

// q[m, j] = w_j^T Lambda_m u_j = sum_k p_k w_kj u_kj - (sum_k p_k w_kj)(sum_k p_k u_kj)
// out[0] = (neigh,neigh), out[1] = (self,neigh), out[2] = (self,self); W1 is [C, ldw]
__global__ void q_kernel(const float* __restrict__ probs, int64_t M, int64_t C, const float* __restrict__ W1,
                         int64_t ldw, int64_t d, int64_t off_self, int64_t off_neigh, int has_self,
                         float* __restrict__ q) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= M * d) return;
  const int64_t m = t / d, j = t - m * d;
  float snn = 0.f, sn = 0.f, sss = 0.f, ss = 0.f, ssn = 0.f;
  if (!probs) {  // regression: Lambda = I (laplace/curvature/curvature.py:429-430), q = sum_k w_kj u_kj
    for (int64_t k = 0; k < C; ++k) {
      const float wn = W1[k * ldw + off_neigh + j];
      snn += wn * wn;
      if (has_self) {
        const float ws = W1[k * ldw + off_self + j];
        sss += ws * ws;
        ssn += ws * wn;
      }
    }
    q[t] = snn;
    if (has_self) { q[M * d + t] = ssn; q[2 * M * d + t] = sss; }
    return;
  }
  // centred form: q = sum_k p_k (w_k - wbar)(u_k - ubar), wbar = sum_k p_k w_k -- no cancellation between two large sums
  // (and exactly zero for a single class, where softmax is 1 and Lambda vanishes)
  for (int64_t k = 0; k < C; ++k) {
    const float p = probs[m * C + k];
    sn += p * W1[k * ldw + off_neigh + j];
    if (has_self) ss += p * W1[k * ldw + off_self + j];
  }
  for (int64_t k = 0; k < C; ++k) {
    const float p = probs[m * C + k];
    const float dn = W1[k * ldw + off_neigh + j] - sn;
    snn += p * dn * dn;
    if (has_self) {
      const float dsf = W1[k * ldw + off_self + j] - ss;
      sss += p * dsf * dsf;
      ssn += p * dsf * dn;
    }
  }
  q[t] = snn;
  if (has_self) {
    q[M * d + t] = ssn;
    q[2 * M * d + t] = sss;
  }
}

// Empirical Fisher: the per-sample gradient of the first layer is diag(rho_n) T_n (+ diag(rho_s) S_n), rho[j] = sum_k r_k w_kj,
// so its square has the GGN kernel's form with q(neigh,neigh) = scale rho_n^2, q(self,neigh) = scale rho_s rho_n,
// q(self,self) = scale rho_s^2; the last layer's weights are scale r_k^2.
__global__ void q_ef_kernel(const float* __restrict__ r, int64_t M, int64_t C, const float* __restrict__ W1, int64_t ldw,
                            int64_t d, int64_t off_self, int64_t off_neigh, int has_self, float scale, float* __restrict__ q) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= M * d) return;
  const int64_t m = t / d, j = t - m * d;
  float rn = 0.f, rs = 0.f;
  for (int64_t k = 0; k < C; ++k) {
    const float rk = r[m * C + k];
    rn += rk * W1[k * ldw + off_neigh + j];
    if (has_self) rs += rk * W1[k * ldw + off_self + j];
  }
  q[t] = scale * rn * rn;
  if (has_self) {
    q[M * d + t] = scale * rs * rn;
    q[2 * M * d + t] = scale * rs * rs;
  }
}
__global__ void ef_last_weights_kernel(const float* __restrict__ r, int64_t n, float scale, float* __restrict__ w) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < n) w[t] = scale * r[t] * r[t];
}

// Classification batches of the closed-form diagonal: what batch_prologue + q_kernel did in five launches (mark, clear,
// seed, q, unmark) as one.  One wave per sample: p = softmax(logits[idx[m]]) -> probs, loss += logsumexp - f[y]
// (CrossEntropyLoss(reduction='sum')), the q rows of q_kernel from the probabilities in LDS, and the sticky flags the
// prologue raises (word 1: node id out of range, word 0 = 2: label out of range).  The diagonal needs neither the batch
// positions nor the multiplicities of duplicated ids (every occurrence is its own sample), so nothing is marked.
__global__ __launch_bounds__(256) void diag_prologue_kernel(const float* __restrict__ logits, int64_t C,
                                                            const int64_t* __restrict__ idx, const int64_t* __restrict__ y,
                                                            int64_t M, int64_t N, const float* __restrict__ W1, int64_t ldw,
                                                            int64_t d, int64_t off_self, int64_t off_neigh, int has_self,
                                                            float* __restrict__ probs, float* __restrict__ q,
                                                            float* __restrict__ loss, int* __restrict__ bad) {
  extern __shared__ float sm[];
  __shared__ float loss_part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* p_s = sm + size_t(wave) * C;
  float loss_acc = 0.f;
  for (int64_t m = int64_t(blockIdx.x) * 4 + wave; m < M; m += int64_t(gridDim.x) * 4) {
    const int64_t n = idx[m];
    if (n < 0 || n >= N) {  // no entries in the first-layer kernel, zero weight in the last-layer kernel
      if (lane == 0) bad[1] = 1;
      for (int64_t k = lane; k < C; k += 64) probs[m * C + k] = 0.f;
      for (int64_t j = lane; j < (has_self ? 3 : 1) * d; j += 64) q[(j / d) * M * d + m * d + (j % d)] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int64_t k = lane; k < C; k += 64) { const float v = logits[n * C + k]; p_s[k] = v; mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float fy = 0.f;
    if (lane == 0) {
      const int64_t yy = y[m];
      if (yy < 0 || yy >= C) *bad = 2;
      else fy = logits[n * C + yy];
    }
    float se = 0.f;
    for (int64_t k = lane; k < C; k += 64) { const float e = expf(p_s[k] - mx); p_s[k] = e; se += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const float inv = 1.0f / se;
    for (int64_t k = lane; k < C; k += 64) { const float p = p_s[k] * inv; p_s[k] = p; probs[m * C + k] = p; }
    if (lane == 0) loss_acc += logf(se) + mx - fy;
    // p_s is read across lanes below: same wave, LDS executes a wave's operations in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int64_t j = lane; j < d; j += 64) {  // q_kernel's centred form, same summation order
      float sn = 0.f, ss = 0.f, snn = 0.f, sss = 0.f, ssn = 0.f;
      for (int64_t k = 0; k < C; ++k) {
        sn += p_s[k] * W1[k * ldw + off_neigh + j];
        if (has_self) ss += p_s[k] * W1[k * ldw + off_self + j];
      }
      for (int64_t k = 0; k < C; ++k) {
        const float p = p_s[k];
        const float dn = W1[k * ldw + off_neigh + j] - sn;
        snn += p * dn * dn;
        if (has_self) {
          const float dsf = W1[k * ldw + off_self + j] - ss;
          sss += p * dsf * dsf;
          ssn += p * dsf * dn;
        }
      }
      q[m * d + j] = snn;
      if (has_self) { q[M * d + m * d + j] = ssn; q[2 * M * d + m * d + j] = sss; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the next sample overwrites p_s
  }
  if (lane == 0) loss_part[wave] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (loss_part[0] + loss_part[1]) + (loss_part[2] + loss_part[3]));
}

constexpr int JPT = 16;  // hidden units per thread
constexpr int DCH = 32;  // (sample, neighbour) entries staged per chunk

// grid (i-chunks of 64, j-chunks of 64, sample slabs); block 256 = 64 i-lanes x 4 j-groups of JPT.
//   T[n, j, i] = sum_v P[n, v] act'(h_1[v, j]) E[v, i]     diag(W_0)[j, i] += q[n, j] T^2  (+ the GraphSAGE self terms)
// At the shapes this runs at (Cora: 16 MB in total) the kernel is bound by load LATENCY, not bytes or flops: a thread
// that walks its samples' neighbours one after the other waits for one dependent load chain per neighbour (0.25 ms for
// 1 299 samples).  So the slab's (sample, neighbour) entries are laid out as ONE list -- every sample followed by a
// virtual entry for the node itself that closes the sample (and carries the self row GraphSAGE needs) -- and consumed in
// chunks of DCH: wave 0 resolves the chunk's entries (sample, column, value), all four waves then fetch the entries'
// feature and derivative row slices into LDS with every load of the chunk in flight at once, and the FMAs run from LDS.
template <int HAS_SELF>
__global__ __launch_bounds__(256) void diag_first_layer_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int64_t* __restrict__ idx, int64_t M, int64_t slab, FeatView E, const float* __restrict__ dact,
    int64_t H, const float* __restrict__ q, float* __restrict__ diag_w, float* __restrict__ diag_b) {
  __shared__ float sE[DCH][64], sD[DCH][64], sQ[HAS_SELF ? 3 : 1][DCH][64];
  __shared__ float sa[DCH];
  __shared__ int32_t sv[DCH], sm[DCH];  // column (node) of the entry; sample index if the entry closes its sample, else -1
  __shared__ int32_t soff[65], snode[64];
  const int tid = threadIdx.x, lane = tid & 63, jg = tid >> 6;
  const int64_t i = int64_t(blockIdx.x) * 64 + lane;
  const int64_t j0 = int64_t(blockIdx.y) * 64;
  const int64_t ncols = E.width + 1;
  const bool i_ok = i < ncols;
  const int64_t m_begin = int64_t(blockIdx.z) * slab, m_end = min(M, m_begin + slab);
  const int ns = int(m_end - m_begin);  // <= 64
  // entry offsets of the slab's samples: len(row) + 1 each (invalid node ids: 0, flagged by the batch prologue)
  if (tid < 64) {
    int32_t len = 0, node = -1;
    if (tid < ns) {
      const int64_t n = idx[m_begin + tid];
      if (n >= 0 && n < E.nrows) { node = int32_t(n); len = rowptr[n + 1] - rowptr[n] + 1; }
    }
    snode[tid] = node;
    int32_t incl = len;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    soff[tid + 1] = incl;
    if (tid == 0) soff[0] = 0;
  }
  __syncthreads();
  const int32_t etot = soff[64];
  float acc[JPT], T[JPT];
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) { acc[jj] = 0.f; T[jj] = 0.f; }

  for (int32_t c0 = 0; c0 < etot; c0 += DCH) {
    const int cn = min(DCH, int(etot - c0));
    // (1) wave 0: resolve the chunk's entries
    if (tid < DCH) {
      int32_t v = 0, mk = -1;
      float a = 0.f;
      if (tid < cn) {
        const int32_t g = c0 + tid;
        int lo = 0, hi = ns;  // sample k with soff[k] <= g < soff[k + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (soff[mid] <= g) lo = mid; else hi = mid; }
        const int32_t node = snode[lo], r = g - soff[lo], len = soff[lo + 1] - soff[lo] - 1;
        if (r < len) { const int32_t p = rowptr[node] + r; v = col[p]; a = val[p]; }
        else { v = node; mk = lo; }  // the virtual entry: closes sample lo, weight 0
      }
      sv[tid] = v; sa[tid] = a; sm[tid] = mk;
    }
    __syncthreads();
    // (2) all waves: the entries' row slices, every load of the chunk in flight at once
    for (int e = jg; e < cn; e += 4) {
      const int64_t v = sv[e];
      sE[e][lane] = i_ok ? feat(E, v, i) : 0.f;
      sD[e][lane] = (j0 + lane < H) ? dact[v * H + j0 + lane] : 0.f;
      const int mk = sm[e];
      if (mk >= 0) {
        const int64_t m = m_begin + mk;
        const bool ok = j0 + lane < H;
        sQ[0][e][lane] = ok ? q[m * H + j0 + lane] : 0.f;
        if (HAS_SELF) {
          sQ[1][e][lane] = ok ? q[M * H + m * H + j0 + lane] : 0.f;
          sQ[2][e][lane] = ok ? q[2 * M * H + m * H + j0 + lane] : 0.f;
        }
      }
    }
    __syncthreads();
    // (3) FMAs from LDS; a closing entry folds the finished T into the accumulator
    for (int e = 0; e < cn; ++e) {
      const float ev = sE[e][lane];
      const float4* __restrict__ dr = reinterpret_cast<const float4*>(&sD[e][jg * JPT]);
      const int mk = sm[e];
      if (mk < 0) {
        const float ea = ev * sa[e];
#pragma unroll
        for (int q4 = 0; q4 < JPT / 4; ++q4) {
          const float4 d4 = dr[q4];
          T[4 * q4] = fmaf(d4.x, ea, T[4 * q4]); T[4 * q4 + 1] = fmaf(d4.y, ea, T[4 * q4 + 1]);
          T[4 * q4 + 2] = fmaf(d4.z, ea, T[4 * q4 + 2]); T[4 * q4 + 3] = fmaf(d4.w, ea, T[4 * q4 + 3]);
        }
      } else {
#pragma unroll
        for (int jj = 0; jj < JPT; ++jj) {
          const float qbb = sQ[0][e][jg * JPT + jj];
          if (HAS_SELF) {
            const float sf = sD[e][jg * JPT + jj] * ev;
            acc[jj] += qbb * T[jj] * T[jj] + 2.f * sQ[1][e][jg * JPT + jj] * sf * T[jj] + sQ[2][e][jg * JPT + jj] * sf * sf;
          } else {
            acc[jj] += qbb * T[jj] * T[jj];
          }
          T[jj] = 0.f;
        }
      }
    }
    __syncthreads();
  }
  if (!i_ok) return;
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int64_t j = j0 + jg * JPT + jj;
    if (j < H) {
      if (i < E.width) atomicAdd(&diag_w[j * E.width + i], acc[jj]);
      else atomicAdd(&diag_b[j], acc[jj]);
    }
  }
}

// ---- the same contraction, rows by LDS-DMA, the wave-uniform operand by v_readlane --------------------------------------
// diag_first_layer_kernel above stages a chunk of entries, waits for it, computes, and starts over.  Measured at the Cora
// shape (0.131 ms for a ~10-20 us VALU floor): (1) per chunk one exposed round trip for the entries (row pointer ->
// column / value) and one for their rows; (2) in the FMA loop every thread reads the 16 act' values of its wave with
// four ds_read_b128 that all 64 lanes address identically -- 1 KB of LDS bandwidth per instruction for 64 B of data,
// 3 GB of LDS reads per launch -- and waits for LDS twice per entry (closing mark, then data).  Here
// (a) ALL entries of a slab are resolved up front, in parallel, into LDS (one chain per workgroup, not one per chunk);
// (b) the entries' rows -- E for the lanes' 64 input columns, act' (or, for the entry that closes a GCN sample, q) for
//     the tile's 64 hidden units -- are copied global -> LDS by the DMA path (global_load_lds, one dword per lane: the
//     [entry][64 lanes] layout, no data registers) into a ring of NBUF slots, NBUF - 1 chunks ahead of the FMAs, behind
//     counted s_waitcnt vmcnt -- every wave issues the same number of copies per chunk (padding entries copy a zero
//     word) so that the count is a compile-time constant -- and one raw barrier per chunk;
// (c) both rows are read ONE dword per lane (conflict free, 512 B per entry and wave instead of 4.3 KB); the hidden-unit
//     operand, which is the same for all lanes, is taken out of the row register with v_readlane (an SGPR operand of the
//     FMA): 16 readlanes + 8 packed FMAs per entry, no LDS broadcast, and the next entry's two dwords are loaded while
//     the current one computes.
// (Tried and measured slower: the uniform operand through the scalar cache, s_load_dwordx16 per entry -- 0.111 ms: the
//  scalar data cache has little miss parallelism, and its out-of-order returns force lgkmcnt(0) waits.)
__device__ float g_diag_consts[2] = {0.f, 1.f};

__device__ __forceinline__ void lds_dma4(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                            reinterpret_cast<uintptr_t>(lds_dst)), 4, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// all but the youngest `ahead` chunks' copies (NI instructions each) have landed; ahead <= MAXA (wave uniform)
template <int NI, int MAXA> __device__ __forceinline__ void wait_chunks(int ahead) {
  if constexpr (MAXA == 0) {
    wait_vmcnt<0>();
  } else {
    if (ahead >= MAXA) wait_vmcnt<MAXA * NI>();
    else wait_chunks<NI, MAXA - 1>(ahead);
  }
}
__device__ __forceinline__ float lane_value(float x, int l) {  // x of lane l (wave uniform l) as a scalar operand
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

constexpr int kDiagMaxEntries = 512;  // entries resolved per pass (a slab of 64 Cora samples has ~380)

template <int HAS_SELF, int DCH, int NBUF>
__global__ __launch_bounds__(256) void diag_first_layer_dma_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int64_t* __restrict__ idx, int64_t M, int64_t slab, FeatView E, const float* __restrict__ dact,
    int64_t H, const float* __restrict__ q, float* __restrict__ diag_w, float* __restrict__ diag_b) {
  constexpr int ARR = HAS_SELF ? 5 : 2;  // arrays per slot: E rows, act' rows (GCN closing entry: its q row) [, 3 q rows]
  constexpr int BUF = ARR * DCH * 64;    // floats per ring slot
  constexpr int NI = (DCH / 4) * ARR;    // DMA instructions per wave and chunk
  constexpr int MAXE = kDiagMaxEntries;
  static_assert(NBUF >= 2 && NBUF <= 4 && (NBUF - 2) * NI <= 63 && (DCH & (DCH - 1)) == 0, "ring: vmcnt is a 6-bit counter");
  // ONE LDS object (hipcc drains vmcnt before reads of a second one): ring | entries {column, weight, closing mark} | slab tables
  __shared__ float smem[NBUF * BUF + 4 * (MAXE + 2) + 65 + 64 + 64 + 63];
  float* __restrict__ ring = smem;
  // (read as scalars: a vector-typed LDS read makes hipcc wait for every LDS-DMA copy in flight)
  int32_t* __restrict__ smeta = reinterpret_cast<int32_t*>(smem + NBUF * BUF);
  int32_t* __restrict__ soff = reinterpret_cast<int32_t*>(smem + NBUF * BUF + 4 * (MAXE + 2));
  int32_t* __restrict__ snode = soff + 65;
  int32_t* __restrict__ sbase = snode + 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int jg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t i = int64_t(blockIdx.x) * 64 + lane;
  const int64_t j0 = int64_t(blockIdx.y) * 64;
  const bool i_ok = i <= E.width, j_ok = j0 + lane < H;
  const int64_t m_begin = int64_t(blockIdx.z) * slab, m_end = min(M, m_begin + slab);
  const int ns = int(m_end - m_begin);  // <= 64
  if (tid < 64) {
    int32_t len = 0, node = -1, base = 0;
    if (tid < ns) {
      const int64_t n = idx[m_begin + tid];
      if (n >= 0 && n < E.nrows) { node = int32_t(n); base = rowptr[n]; len = rowptr[n + 1] - base + 1; }
    }
    snode[tid] = node;
    sbase[tid] = base;
    int32_t incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    soff[tid + 1] = incl;
    if (tid == 0) soff[0] = 0;
  }
  __syncthreads();
  const int32_t etot = soff[64];
  float acc[JPT], T[JPT];
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) { acc[jj] = 0.f; T[jj] = 0.f; }
  const float* __restrict__ zero = g_diag_consts;
  // this lane's column of E: a feature column, the bias column (rowsum(P) per node, or the constant 1), or nothing
  const float* __restrict__ ecol = !i_ok ? nullptr : (i < E.width ? E.base + i : E.bias_col);
  const int64_t estride = !i_ok ? 0 : (i < E.width ? E.ld : (E.bias_col ? 1 : 0));
  if (i_ok && i == E.width && !E.bias_col) ecol = g_diag_consts + 1;
  const float* __restrict__ dcol = dact + j0 + lane;               // this lane's hidden unit of the act' rows ...
  const float* __restrict__ qcol = q + m_begin * H + j0 + lane;    // ... and of the slab's q rows
  const int jb = jg * JPT;  // this wave's 16 hidden units sit in lanes jb .. jb + 15 of a row register

  for (int32_t sb0 = 0; sb0 < etot; sb0 += MAXE) {
    const int ne = min(MAXE, int(etot - sb0));
    for (int e = tid; e < ne + 2; e += 256) {  // (a) every entry of this pass: sample by bisection, then column / value
      const int32_t g = sb0 + e;
      struct { int32_t x, y, z; } m = {0, 0, -1};  // past the end: two entries of weight 0 for the look-ahead
      if (e < ne) {
        int lo = 0, hi = ns;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (soff[mid] <= g) lo = mid; else hi = mid; }
        const int32_t r = g - soff[lo], len = soff[lo + 1] - soff[lo] - 1;
        m.x = snode[lo]; m.z = lo;  // the virtual entry: the node itself, closes sample lo, weight 0
        if (r < len) { const int32_t p = sbase[lo] + r; m.x = col[p]; m.y = __builtin_bit_cast(int, val[p]); m.z = -1; }
      }
      smeta[4 * e] = m.x; smeta[4 * e + 1] = m.y; smeta[4 * e + 2] = m.z;
    }
    __syncthreads();
    const int nch = (ne + DCH - 1) / DCH;
    auto issue = [&](int c) {  // (b) chunk c -> ring slot c % NBUF; wave jg copies the rows of entries jg, jg + 4, ...
      float* __restrict__ slot = ring + (c % NBUF) * BUF;
#pragma unroll
      for (int k = 0; k < DCH / 4; ++k) {
        const int el = jg + 4 * k, e = c * DCH + el, em = min(e, MAXE + 1);
        const bool ok = e < ne;
        const int64_t v = smeta[4 * em];
        const int mk = smeta[4 * em + 2];
        const bool nb = ok && mk < 0, cl = ok && mk >= 0;
        lds_dma4(((nb || (HAS_SELF && cl)) && i_ok) ? ecol + v * estride : zero, slot + el * 64);
        if (HAS_SELF) {
          lds_dma4((ok && j_ok) ? dcol + v * H : zero, slot + (DCH + el) * 64);
#pragma unroll
          for (int z = 0; z < 3; ++z)
            lds_dma4((cl && j_ok) ? qcol + int64_t(mk) * H + int64_t(z) * M * H : zero, slot + ((2 + z) * DCH + el) * 64);
        } else {  // the entry that closes a GCN sample reads no act' row: its row is the sample's q row
          lds_dma4(j_ok ? (nb ? dcol + v * H : (cl ? qcol + int64_t(mk) * H : zero)) : zero, slot + (DCH + el) * 64);
        }
      }
    };
    for (int c = 0; c < NBUF - 1 && c < nch; ++c) issue(c);
    // (c) entry g computes while entry g + 1's dwords are on their way from LDS
    float ev0 = 0.f, r0 = 0.f, a0 = 0.f, ev1 = 0.f, r1 = 0.f, a1 = 0.f;
    int mk0 = -1, rmk1 = -1, g = 0;
    auto fetch = [&]() {  // ring step at a chunk's first entry; the next entry's dwords (same chunk) start their way
      const int el = g & (DCH - 1), c = g / DCH;
      const float* __restrict__ slot = ring + (c % NBUF) * BUF;
      if (el == 0) {
        const int ahead = min(NBUF - 2, nch - 1 - c);  // chunks after c whose copies are in flight
        if (NBUF >= 4 && ahead >= 2) wait_vmcnt<(NBUF >= 4 ? 2 : 0) * NI>();
        else if (ahead == 1) wait_vmcnt<NI>();
        else wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");  // chunk c has landed for all waves; chunk c - 1 is consumed
        if (c + NBUF - 1 < nch) issue(c + NBUF - 1);
        ev0 = slot[lane]; r0 = slot[DCH * 64 + lane];  // (a chunk's first rows are read behind its barrier)
        a0 = __builtin_bit_cast(float, smeta[4 * g + 1]);
      }
      if (el + 1 < DCH) {
        ev1 = slot[(el + 1) * 64 + lane]; r1 = slot[(DCH + el + 1) * 64 + lane];
        a1 = __builtin_bit_cast(float, smeta[4 * (g + 1) + 1]);
      }
      rmk1 = smeta[4 * (g + 1) + 2];
      return slot + el * 64 + lane;
    };
    auto rotate = [&]() { ev0 = ev1; r0 = r1; a0 = a1; mk0 = __builtin_amdgcn_readfirstlane(rmk1); ++g; };
    mk0 = __builtin_amdgcn_readfirstlane(smeta[2]);
    while (g < ne) {
      while (mk0 < 0 && g < ne) {  // neighbour entries: T += act'(h_1[v]) (x) a E[v]
        fetch();
        const float ea = ev0 * a0;
#pragma unroll
        for (int jj = 0; jj < JPT; ++jj) T[jj] = fmaf(lane_value(r0, jb + jj), ea, T[jj]);
        rotate();
      }
      if (g < ne) {  // the entry that closes its sample folds the finished T into the accumulator
        const float* __restrict__ mine = fetch();
        if (HAS_SELF) {
          const float q0v = mine[2 * DCH * 64], q1v = mine[3 * DCH * 64], q2v = mine[4 * DCH * 64];
#pragma unroll
          for (int jj = 0; jj < JPT; ++jj) {
            const float sf = lane_value(r0, jb + jj) * ev0;
            acc[jj] += lane_value(q0v, jb + jj) * T[jj] * T[jj] + 2.f * lane_value(q1v, jb + jj) * sf * T[jj] +
                       lane_value(q2v, jb + jj) * sf * sf;
            T[jj] = 0.f;
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < JPT; ++jj) { acc[jj] = fmaf(lane_value(r0, jb + jj) * T[jj], T[jj], acc[jj]); T[jj] = 0.f; }
        }
        rotate();
      }
    }
    __syncthreads();  // the next pass overwrites the entry table and the ring
  }
  if (!i_ok) return;
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int64_t j = j0 + jb + jj;
    if (j < H) {
      if (i < E.width) atomicAdd(&diag_w[j * E.width + i], acc[jj]);
      else atomicAdd(&diag_b[j], acc[jj]);
    }
  }
}

// ---- GCN: the per-sample outer products on the matrix cores ---------------------------------------------------------------
// T_n = sum_v P[n, v] act'(h_1[v]) (x) E[v] is a [64 hidden x 64 input] product with K = the sample's row length: with
// v_mfma_f32_32x32x2_f32 a pair of entries is one K step, lane l supplying A[j = l & 31][k = l >> 5] = act'(h_1[v_k, j]) and
// B[k][i = l & 31] = P[n, v_k] E[v_k, i] -- BOTH operands are one dword per lane straight from the staged rows; no operand
// is wave uniform, so none of the broadcast traffic (LDS b128, scalar cache, v_readlane: see above) exists.  The VALU
// kernel's counters at the Cora shape (profiles/r03_cora_sq_counters_valu.txt): 30 M VALU instructions for 5.6 M packed
// FMAs -- 11 M of them v_readlane -- 54 % VALU busy for 0.12 ms; here the VALU only folds a finished sample,
// acc += q[j] T[j, i]^2 on the accumulator layout, 64 packed instructions per sample.  Measured at the Cora shape
// (1 299 samples, 42 slabs x 6 column blocks = 252 workgroups): 0.067 ms against 0.131 ms for the round-2 kernel; by
// switching phases off (DESIGN.md has the table): prologue 8 us, entry resolution + operand reads 21 us, row copies 7 us,
// MFMAs 14 us, folds 8 us, the 64 float atomics per lane of the flush 15 us -- a latency chain of one wave per SIMD, not
// a throughput bound: shorter slabs overlap more workgroups per CU but multiply the atomics (5.6 us per million).
//   * a wave owns a 64 x 64 tile of (hidden, input) pairs: T and the running sum are 4 + 4 accumulator tiles (128 VGPRs);
//     the 4 waves of a workgroup share the slab's entry stream and the act' rows and take 4 adjacent column blocks;
//   * E = [P X | rowsum(P) | 0] is ONE padded matrix (the bias column sits in the first pad column: context.hip
//     build_px), so a row's 256-column block is one 16-byte-per-lane LDS-DMA copy;
//   * samples are padded to an even number of entries (weight 0) so that a K step never straddles two samples;
//   * entries resolved up front, rows through the NBUF-slot DMA ring with counted vmcnt waits, as in the VALU kernel.
__device__ float g_diag_zero16[4] = {0.f, 0.f, 0.f, 0.f};
using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

__device__ __forceinline__ void lds_dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                            reinterpret_cast<uintptr_t>(lds_dst)), 16, 0, 0);
}

template <int DCH, int NBUF>
__global__ __launch_bounds__(256) void diag_first_layer_mfma_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int64_t* __restrict__ idx, int64_t M, int64_t slab, const float* __restrict__ E, int64_t ldE, int64_t F, int64_t N,
    const float* __restrict__ dact, int64_t H, const float* __restrict__ q, float* __restrict__ diag_w,
    float* __restrict__ diag_b) {
  constexpr int ROW = 256 + 64;        // floats per staged entry: 256 columns of E, 64 hidden units of act'
  constexpr int BUF = DCH * ROW;       // floats per ring slot
  constexpr int NI = DCH / 2;          // DMA instructions per wave and chunk (DCH / 4 entries x 2 rows)
  constexpr int NP = DCH / 2;          // K steps (pairs of entries) per chunk
  constexpr int MAXE = kDiagMaxEntries;
  static_assert(NBUF >= 2 && (NBUF - 2) * NI <= 63 && DCH % 4 == 0 && MAXE % DCH == 0, "ring: vmcnt is a 6-bit counter");
  // ONE LDS object (hipcc drains vmcnt before reads of a second one):
  // ring | q rows of the slab [64][64] | entries {column, weight, pair record} | slab tables
  __shared__ float smem[NBUF * BUF + 64 * 64 + 3 * MAXE + 65 + 3 * 64 + 63];
  float* __restrict__ ring = smem;
  float* __restrict__ sq = smem + NBUF * BUF;
  int32_t* __restrict__ sv = reinterpret_cast<int32_t*>(sq + 64 * 64);
  float* __restrict__ sa = sq + 64 * 64 + MAXE;
  // record of the K step that starts at an even entry: sample | 256 (first step of its sample) | 512 (last); -1 past the end
  int32_t* __restrict__ sflag = reinterpret_cast<int32_t*>(sa + MAXE);
  int32_t* __restrict__ soff = sflag + MAXE;  // padded (even) entry offsets of the samples
  int32_t* __restrict__ snode = soff + 65;
  int32_t* __restrict__ sbase = snode + 64;
  int32_t* __restrict__ slen = sbase + 64;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t cblk = int64_t(blockIdx.x) * 256;   // the workgroup's 256 columns of E; wave w: cblk + 64 w .. + 63
  const int64_t j0 = int64_t(blockIdx.y) * 64;
  const int64_t m_begin = int64_t(blockIdx.z) * slab, m_end = min(M, m_begin + slab);
  const int ns = int(m_end - m_begin);  // <= 64
  if (tid < 64) {
    int32_t len = 0, node = -1, base = 0;
    if (tid < ns) {
      const int64_t n = idx[m_begin + tid];
      if (n >= 0 && n < N) { node = int32_t(n); base = rowptr[n]; len = rowptr[n + 1] - base; }
    }
    snode[tid] = node;
    sbase[tid] = base;
    slen[tid] = len;
    int32_t incl = (len + 1) & ~1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    soff[tid + 1] = incl;
    if (tid == 0) soff[0] = 0;
  }
  for (int f = tid; f < 64 * 64; f += 256) {  // q rows of the slab's samples (zero past the slab / past H)
    const int sidx = f >> 6, j = f & 63;
    sq[f] = (sidx < ns && j0 + j < H) ? q[(m_begin + sidx) * H + j0 + j] : 0.f;
  }
  __syncthreads();
  const int32_t etot = soff[64];
  f32x16 T[2][2], acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { T[a][b][r] = 0.f; acc[a][b][r] = 0.f; }
  // DMA sources of this lane: 4 columns of E (16 bytes), one hidden unit of act'
  const bool e_ok = cblk + 4 * lane < ldE, j_ok = j0 + lane < H;
  const float* __restrict__ esrc = E + cblk + 4 * lane;
  const float* __restrict__ dsrc = dact + j0 + lane;

  for (int32_t sb0 = 0; sb0 < etot; sb0 += MAXE) {
    const int ne = min(MAXE, int(etot - sb0));  // even
    const int nch = (ne + DCH - 1) / DCH;
    for (int e = tid; e < nch * DCH; e += 256) {  // every entry of this pass: sample by bisection, then column / value
      const int32_t g = sb0 + e;
      int32_t v = 0, fl = -1;
      float a = 0.f;
      if (e < ne) {
        int lo = 0, hi = ns;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (soff[mid] <= g) lo = mid; else hi = mid; }
        const int32_t r = g - soff[lo], len = slen[lo];
        v = snode[lo];  // (the padding entry of an odd row: weight 0)
        if (r < len) { const int32_t p = sbase[lo] + r; v = col[p]; a = val[p]; }
        fl = lo | (r == 0 ? 256 : 0) | (r + 2 >= len ? 512 : 0);
      }
      sv[e] = v; sa[e] = a; sflag[e] = fl;
    }
    __syncthreads();
    auto issue = [&](int c) {  // chunk c -> ring slot c % NBUF; wave w copies the rows of entries w, w + 4, ...
      float* __restrict__ slot = ring + (c % NBUF) * BUF;
#pragma unroll
      for (int k = 0; k < DCH / 4; ++k) {
        const int el = wave + 4 * k, e = c * DCH + el;
        const bool ok = e < ne;
        const int64_t v = sv[e];
        lds_dma16((ok && e_ok) ? esrc + v * ldE : g_diag_zero16, slot + el * ROW);
        lds_dma4((ok && j_ok) ? dsrc + v * H : g_diag_zero16, slot + el * ROW + 256);
      }
    };
    for (int c = 0; c < NBUF - 1 && c < nch; ++c) issue(c);
    for (int c = 0; c < nch; ++c) {
      wait_chunks<NI, NBUF - 2>(min(NBUF - 2, nch - 1 - c));  // chunks after c whose copies are in flight
      asm volatile("s_barrier" ::: "memory");  // chunk c has landed for all waves; chunk c - 1 is consumed
      if (c + NBUF - 1 < nch) issue(c + NBUF - 1);
      // the chunk's K steps: all operands on their way before the first product (one LDS round trip per chunk, not per step)
      const float* __restrict__ slot = ring + (c % NBUF) * BUF + lhi * ROW;  // this lane's K index: entry 2 p + lhi
      float A0[NP], A1[NP], B0[NP], B1[NP];
      int fl[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float* __restrict__ row = slot + 2 * p * ROW;
        const float w = sa[c * DCH + 2 * p + lhi];
        A0[p] = row[256 + l31]; A1[p] = row[256 + 32 + l31];
        B0[p] = w * row[wave * 64 + l31]; B1[p] = w * row[wave * 64 + 32 + l31];
        fl[p] = __builtin_amdgcn_readfirstlane(sflag[c * DCH + 2 * p]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (fl[p] < 0) break;  // past the last entry
        if (fl[p] & 256) {     // a sample's first step starts from zero
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          T[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[p], B0[p], z, 0, 0, 0);
          T[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[p], B1[p], z, 0, 0, 0);
          T[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[p], B0[p], z, 0, 0, 0);
          T[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[p], B1[p], z, 0, 0, 0);
        } else {
          T[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[p], B0[p], T[0][0], 0, 0, 0);
          T[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[p], B1[p], T[0][1], 0, 0, 0);
          T[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[p], B0[p], T[1][0], 0, 0, 0);
          T[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[p], B1[p], T[1][1], 0, 0, 0);
        }
        if (fl[p] & 512) {  // the sample is complete: acc += q[j] T[j, i]^2
          // accumulator register r: j = 32 jt + 8 (r >> 2) + 4 lhi + (r & 3)
          const float* __restrict__ qs = sq + (fl[p] & 255) * 64 + 4 * lhi;
#pragma unroll
          for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float qv = qs[32 * jt + 8 * (r >> 2) + (r & 3)];
#pragma unroll
              for (int it = 0; it < 2; ++it) acc[jt][it][r] = fmaf(qv * T[jt][it][r], T[jt][it][r], acc[jt][it][r]);
            }
        }
      }
    }
    __syncthreads();  // the next pass overwrites the entry table and the ring
  }
  // flush: accumulator register r of tile (jt, it) is (hidden j0 + 32 jt + 8 (r >> 2) + 4 lhi + (r & 3), column + 32 it + l31)
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t i = cblk + wave * 64 + it * 32 + l31;
      if (i > F) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t j = j0 + 32 * jt + 8 * (r >> 2) + 4 * lhi + (r & 3);
        if (j < H) atomicAdd(i < F ? &diag_w[j * F + i] : &diag_b[j], acc[jt][it][r]);
      }
    }
}

// ---- GCN, wide inputs (Cora: F = 1 433): the same per-sample outer products, output tiles owned ------------------------------
// Round 3's kernel above cut the samples into 42 slabs and every workgroup flushed its [64 x 256] partial tile with float
// atomics: 35.9 MB through the fabric for a 0.37 MB result (97 x), E re-fetched by whichever XCD a workgroup landed on, and a
// per-workgroup latency chain (32 samples) that left the matrix pipes 12 % busy.  Here
//   * a workgroup (4 waves, one per SIMD) owns 192 columns x 64 hidden units for a LONG slab of samples (Cora: 8 column blocks x
//     32 slabs of 41 samples); wave (jh, ch) holds a 32 x 96 tile of T and of the running sum (3 + 3 accumulator tiles: one
//     A operand read feeds three MFMAs);
//   * blockIdx.x = column block + ncb * slab: with 8 column blocks the workgroups of a column block share an XCD (blocks are
//     dealt round-robin: speed only), so an XCD's L2 holds exactly its 192-column slice of E (2 MB) -- E crosses the fabric once;
//   * ONE 16-byte-per-lane LDS-DMA copy stages an entry: lanes 0 .. 47 its 192 columns of E, lanes 48 .. 63 the 64 act' values;
//   * sqrt(q) rides on the A operand (q = w^T Lambda w >= 0), so folding a finished sample is one FMA per accumulator register,
//     acc += T'^2 -- the fp32 MFMAs and the VALU share the SIMD's ALUs, every saved vector instruction is matrix time;
//   * the slab's partial tile goes to a workspace with PLAIN stores, every element written exactly once;
//     diag_tile_reduce_kernel adds the slabs up in a fixed order: no atomics, deterministic, 16 x 0.37 MB written.
constexpr int kTileCols = 192;
constexpr int kTileDch = 12;    // entries per ring chunk (two copies per wave)
constexpr int kTileNbuf = 4;
constexpr int kTileSlabMax = 96;
constexpr int kTileMaxE = 768;  // entries resolved per pass

__global__ __launch_bounds__(256) void diag_first_layer_tile_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int64_t* __restrict__ idx, int64_t M, int64_t slab, int ncb, const float* __restrict__ E, int64_t ldE, int64_t F,
    int64_t N, const float* __restrict__ dact, int64_t H, const float* __restrict__ q, float* __restrict__ part,
    int64_t part_ld) {
  constexpr int DCH = kTileDch, NBUF = kTileNbuf, ROW = 256, BUF = DCH * ROW, NI = DCH / 4, NP = DCH / 2, MAXE = kTileMaxE;
  constexpr int SM = kTileSlabMax;
  static_assert((NBUF - 2) * NI <= 63 && DCH % 4 == 0 && MAXE % DCH == 0, "ring");
  // ONE LDS object: ring | sqrt(q) rows of the slab [SM][64] | entries {column, weight, pair record} | slab tables
  __shared__ __attribute__((aligned(16))) float smem[NBUF * BUF + SM * 64 + 3 * MAXE + (SM + 1) + 3 * SM];
  float* __restrict__ ring = smem;
  float* __restrict__ sq = smem + NBUF * BUF;
  int32_t* __restrict__ sv = reinterpret_cast<int32_t*>(sq + SM * 64);
  float* __restrict__ sa = sq + SM * 64 + MAXE;
  // record of the K step that starts at an even entry: sample | 256 (first step of its sample) | 512 (last); -1 past the end
  int32_t* __restrict__ sflag = reinterpret_cast<int32_t*>(sa + MAXE);
  int32_t* __restrict__ soff = sflag + MAXE;  // padded (even) entry offsets of the samples
  int32_t* __restrict__ snode = soff + SM + 1;
  int32_t* __restrict__ sbase = snode + SM;
  int32_t* __restrict__ slen = sbase + SM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jh = wave >> 1, ch = wave & 1;  // hidden half, column half (96 columns = three 32-column MFMA tiles)
  const int cb = int(blockIdx.x) % ncb, slab_i = int(blockIdx.x) / ncb;
  const int64_t cblk = int64_t(cb) * kTileCols;
  const int64_t j0 = int64_t(blockIdx.y) * 64;
  const int64_t m_begin = int64_t(slab_i) * slab, m_end = min(M, m_begin + slab);
  const int ns = int(m_end - m_begin);  // <= SM
  if (tid < 64) {  // exclusive prefix sums of the padded row lengths, two samples per lane (SM <= 128)
    int32_t len2[2], incl = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sidx = 2 * tid + u;
      int32_t len = 0, node = -1, base = 0;
      if (sidx < ns) {
        const int64_t n = idx[m_begin + sidx];
        if (n >= 0 && n < N) { node = int32_t(n); base = rowptr[n]; len = rowptr[n + 1] - base; }
      }
      if (sidx < SM) { snode[sidx] = node; sbase[sidx] = base; slen[sidx] = len; }
      len2[u] = (len + 1) & ~1;
      incl += len2[u];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (2 * tid + 1 <= SM) soff[2 * tid + 1] = incl - len2[1];
    if (2 * tid + 2 <= SM) soff[2 * tid + 2] = incl;
    if (tid == 0) soff[0] = 0;
  }
  for (int f = tid; f < SM * 64; f += 256) {  // sqrt(q) rows of the slab's samples (zero past the slab / past H)
    const int sidx = f >> 6, j = f & 63;
    sq[f] = (sidx < ns && j0 + j < H) ? sqrtf(fmaxf(q[(m_begin + sidx) * H + j0 + j], 0.f)) : 0.f;
  }
  __syncthreads();
  const int32_t etot = soff[min(ns, SM)];
  f32x16 T[3], acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { T[b][r] = 0.f; acc[b][r] = 0.f; }
  // DMA source of this lane: lanes 0 .. 47 four columns of E, lanes 48 .. 63 four hidden units of act'
  const bool is_e = lane < 48;
  const int64_t ecol = cblk + 4 * lane, dcol = j0 + 4 * (lane - 48);
  const bool src_ok = is_e ? ecol < ldE : dcol < H;
  const float* __restrict__ src0 = is_e ? E + ecol : dact + dcol;
  const int64_t src_ld = is_e ? ldE : H;

  for (int32_t sb0 = 0; sb0 < etot; sb0 += MAXE) {
    const int ne = min(MAXE, int(etot - sb0));  // even
    const int nch = (ne + DCH - 1) / DCH;
    for (int e = tid; e < nch * DCH; e += 256) {  // every entry of this pass: sample by bisection, then column / value
      const int32_t g = sb0 + e;
      int32_t v = 0, fl = -1;
      float a = 0.f;
      if (e < ne) {
        int lo = 0, hi = ns;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (soff[mid] <= g) lo = mid; else hi = mid; }
        const int32_t r = g - soff[lo], len = slen[lo];
        v = max(snode[lo], 0);  // (the padding entry of an odd row: weight 0)
        if (r < len) { const int32_t p = sbase[lo] + r; v = col[p]; a = val[p]; }
        fl = lo | (r == 0 ? 256 : 0) | (r + 2 >= len ? 512 : 0);
      }
      sv[e] = v; sa[e] = a; sflag[e] = fl;
    }
    __syncthreads();
    auto issue = [&](int c) {  // chunk c -> ring slot c % NBUF; wave w copies the rows of entries w, w + 4, w + 8
      float* __restrict__ slot = ring + (c % NBUF) * BUF;
#pragma unroll
      for (int k = 0; k < DCH / 4; ++k) {
        const int el = wave + 4 * k, e = c * DCH + el;
        const int64_t v = sv[e];
        lds_dma16((e < ne && src_ok) ? src0 + v * src_ld : g_diag_zero16, slot + el * ROW);
      }
    };
    for (int c = 0; c < NBUF - 1 && c < nch; ++c) issue(c);
    for (int c = 0; c < nch; ++c) {
      wait_chunks<NI, NBUF - 2>(min(NBUF - 2, nch - 1 - c));  // chunks after c whose copies are in flight
      asm volatile("s_barrier" ::: "memory");  // chunk c has landed for all waves; chunk c - 1 is consumed
      if (c + NBUF - 1 < nch) issue(c + NBUF - 1);
      const float* __restrict__ slot = ring + (c % NBUF) * BUF + lhi * ROW;  // this lane's K index: entry 2 p + lhi
      float A[NP], B0[NP], B1[NP], B2[NP];
      int fl[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float* __restrict__ row = slot + 2 * p * ROW;
        fl[p] = __builtin_amdgcn_readfirstlane(sflag[c * DCH + 2 * p]);
        const float sqv = sq[(max(fl[p], 0) & 255) * 64 + 32 * jh + l31];
        A[p] = sa[c * DCH + 2 * p + lhi] * sqv * row[kTileCols + 32 * jh + l31];
        B0[p] = row[ch * 96 + l31]; B1[p] = row[ch * 96 + 32 + l31]; B2[p] = row[ch * 96 + 64 + l31];
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        // (steps past the last entry carry weight 0: they add nothing and need no branch)
        T[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[p], B0[p], T[0], 0, 0, 0);
        T[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[p], B1[p], T[1], 0, 0, 0);
        T[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[p], B2[p], T[2], 0, 0, 0);
        if (fl[p] >= 0 && (fl[p] & 512)) {  // the sample is complete: acc += (sqrt(q_j) T[j, i])^2, T = 0 for the next one
          // A REAL branch (the empty statement keeps hipcc from if-converting it): left to itself it folds on EVERY K step
          // behind 32 v_cndmask and picks the first step's zero accumulator with 32 more -- ~100 vector instructions per
          // K step on the ALUs the fp32 MFMAs run on, which is what round 3's kernel spent its 67 us on.
          asm volatile("" ::: "memory");
#pragma unroll
          for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[b][r] = fmaf(T[b][r], T[b][r], acc[b][r]); T[b][r] = 0.f; }
        }
      }
    }
    __syncthreads();  // the next pass overwrites the entry table and the ring
  }
  // the slab's partial tile: accumulator register r of tile b is (hidden j0 + 32 jh + 8 (r >> 2) + 4 lhi + (r & 3), column + 32 b + l31)
  float* __restrict__ pt = part + int64_t(slab_i) * H * part_ld;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int64_t i = cblk + ch * 96 + b * 32 + l31;
    if (i > F) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t j = j0 + 32 * jh + 8 * (r >> 2) + 4 * lhi + (r & 3);
      if (j < H) pt[j * part_ld + i] = acc[b][r];
    }
  }
}
// diag(W_0)[j, i] += sum_slabs part[slab][j][i] (i < F), diag(b_0)[j] += ... (i == F): a fixed summation order.  Thread t of
// linear block `blk` owns four consecutive columns of one hidden unit (part_ld % 4 == 0: 16-byte reads).  Runs as extra
// z-slices of the last-layer launch (TileReduce below): no launch of its own, and it overlaps that latency-bound kernel.
struct TileReduce {
  const float* part; int64_t part_ld; int nslab; int64_t H, F; float* diag_w; float* diag_b;
  int z0;  // first z-slice of the launch that reduces (z0 < 0: nothing to reduce)
};
__device__ __forceinline__ void diag_tile_reduce(const TileReduce& a, int64_t blk) {
  const int64_t q4 = (a.F + 4) / 4;  // groups of four columns covering 0 .. F
  const int64_t t = blk * 256 + threadIdx.x;
  if (t >= a.H * q4) return;
  const int64_t j = t / q4, i = 4 * (t - j * q4);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < a.nslab; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(a.part + (int64_t(s) * a.H + j) * a.part_ld + i);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  const float e[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (i + u < a.F) a.diag_w[j * a.F + i + u] += e[u];
    else if (i + u == a.F) a.diag_b[j] += e[u];
  }
}

// grid (i-chunks of 256, C, slabs): diag(W)[k,i] += sum_n wgt[n,k] phi[n,i]^2 ; bias with s_n.
// wgt = p_k (1 - p_k) from the probabilities (GGN), or the caller's [M, C] array (empirical Fisher: scale * r_k^2)
__global__ void diag_last_layer_kernel(const float* __restrict__ probs, const int64_t* __restrict__ idx, int64_t M,
                                       int64_t C, int64_t slab, FeatView Phi, float* __restrict__ diag_w,
                                       float* __restrict__ diag_b, const float* __restrict__ wgt, TileReduce red) {
  if (red.z0 >= 0 && int(blockIdx.z) >= red.z0) {  // the first layer's partial tiles (diag_first_layer_tile_kernel)
    diag_tile_reduce(red, blockIdx.x + int64_t(gridDim.x) * (blockIdx.y + int64_t(gridDim.y) * (blockIdx.z - red.z0)));
    return;
  }
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t k = blockIdx.y;
  if (i > Phi.width) return;
  const int64_t m_begin = int64_t(blockIdx.z) * slab, m_end = min(M, m_begin + slab);
  float acc = 0.f;
  // four samples per step: their ids first, then the (dependent) feature loads together -- one latency chain per four samples
  for (int64_t m = m_begin; m < m_end; m += 4) {
    int64_t n[4];
    float wt[4], ph[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) n[u] = m + u < m_end ? idx[m + u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t mm = min(m + u, m_end - 1);
      const float p = wgt ? wgt[mm * C + k] : probs[mm * C + k];
      wt[u] = wgt ? p : p * (1.f - p);
      ph[u] = feat(Phi, n[u], i);  // id -1 (past the slab) or out of range: 0
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += wt[u] * ph[u] * ph[u];
  }
  if (i < Phi.width) atomicAdd(&diag_w[k * Phi.width + i], acc);
  else atomicAdd(&diag_b[k], acc);
}

// ---- last-layer full GGN as weighted Grams over class pairs -----------------------------------------------------
// H = sum_n Lambda_n (x) phi~_n phi~_n^T: block (c, c') of H is the weighted Gram Phi~^T diag(Lambda[:, c, c']) Phi~ -- symmetric
// in (c, c') AND inside the block.  Only the pairs c <= c' and the upper half of every block are computed (a quarter
// of the P x P products; the formulation through Z^T Z, Z = [p_c phi~], computed half).
// pair index q <-> (c, c'), c <= c', row major
__device__ __forceinline__ void pair_of(int64_t q, int64_t C, int64_t& c, int64_t& c2) {
  c = 0;
  while (q >= C - c) { q -= C - c; ++c; }
  c2 = c + q;
}
// Phi[m][j] = phi~ of batch row m (j < D features, j == D bias scale), zero padded to ldp
__global__ void ll_build_phi_kernel(const int64_t* __restrict__ idx, int64_t M, FeatView Phi, int64_t ldp,
                                    float* __restrict__ out) {
  const int64_t total = M * ldp;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t m = t / ldp, j = t - m * ldp;
    out[t] = j <= Phi.width ? feat(Phi, idx[m], j) : 0.f;
  }
}
// w = Lambda[m, c, c'] = [c == c'] p_c - p_c p_c'.  rs[q][m] = sqrt|w| (row scale of the Gram), ws[q][m] = w * s_m (signed,
// times the bias scale of the row: the bias column of a block is sum_m w s_m phi~_m), zsign[q] = +1 (c == c') / -1
__global__ void ll_pair_weights_kernel(const float* __restrict__ probs, const int64_t* __restrict__ idx, FeatView Phi,
                                       int64_t M, int64_t C, int64_t q0, float* __restrict__ rs, float* __restrict__ ws,
                                       float* __restrict__ zsign) {
  const int64_t q = blockIdx.y;  // index inside the chunk of pairs that starts at q0
  int64_t c, c2;
  pair_of(q0 + q, C, c, c2);
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; m < M; m += stride) {
    const float pc = probs[m * C + c], pc2 = probs[m * C + c2];
    const float w = (c == c2 ? pc : 0.f) - pc * pc2;
    rs[q * M + m] = sqrtf(fabsf(w));
    ws[q * M + m] = w * feat(Phi, idx[m], Phi.width);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) zsign[q] = c == c2 ? 1.f : -1.f;
}
// S[q] [D x D] (upper 32 x 32 sub-tiles valid) and Sb[q] [D + 1] (bias column) -> upper triangle of H.
// H index of (class c, column a): a < D -> c * D + a, a == D (bias) -> C * D + c.
// One workgroup per (32 x 32 tile (ti <= tj) of the block, pair): the tile goes through LDS so that the block's (ti, tj)
// part and -- for c < c' or an off-diagonal tile -- its mirror image (tj, ti) are both written with coalesced rows.
__global__ __launch_bounds__(256) void ll_place_tiles_kernel(const float* __restrict__ S, int64_t D, int64_t C, int64_t q0,
                                                             float* __restrict__ Hout) {
  __shared__ float t[32][33];
  const int64_t P = C * D + C;
  int64_t c, c2;
  pair_of(q0 + blockIdx.z, C, c, c2);
  const int64_t ti = blockIdx.y, tj = blockIdx.x;
  if (ti > tj) return;
  const float* __restrict__ Sq = S + int64_t(blockIdx.z) * D * D;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = ti * 32 + r, j = tj * 32 + tx;
    t[r][tx] = (i < D && j < D) ? Sq[i * D + j] : 0.f;
  }
  __syncthreads();
  // block element (i, j) -> H[c D + i][c2 D + j]   (c <= c2: always in the upper triangle for c < c2; for c == c2 keep i <= j)
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = ti * 32 + r, j = tj * 32 + tx;
    if (i < D && j < D && (c < c2 || i <= j)) Hout[(c * D + i) * P + c2 * D + j] += t[r][tx];
  }
  if (ti < tj && c < c2) {  // the block is symmetric: element (j, i) of it equals (i, j); rows j of H, coalesced over i
    for (int r = ty; r < 32; r += 8) {
      const int64_t j = tj * 32 + r, i = ti * 32 + tx;
      if (i < D && j < D) Hout[(c * D + j) * P + c2 * D + i] += t[tx][r];
    }
  }
}
// bias column of every block: H[(c, a)][(bias c2)] and H[(c2, a)][(bias c)] for a < D, H[bias c][bias c2]
__global__ void ll_place_bias_kernel(const float* __restrict__ Sb, int64_t D, int64_t C, int64_t q0, int64_t nq,
                                     float* __restrict__ Hout) {
  const int64_t D1 = D + 1, P = C * D + C;
  const int64_t total = nq * D1;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t tq = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; tq < total; tq += stride) {
    const int64_t q = tq / D1, a = tq - q * D1;
    int64_t c, c2;
    pair_of(q0 + q, C, c, c2);
    const float v = Sb[tq];
    if (a == D) { Hout[(C * D + c) * P + C * D + c2] += v; continue; }  // (bias c, bias c2), c <= c2
    Hout[(c * D + a) * P + C * D + c2] += v;                            // ((c, a), bias c2): row < column always
    if (c != c2) Hout[(c2 * D + a) * P + C * D + c] += v;               // ((c2, a), bias c)
  }
}

// Generic depth: diag[p] += sum_m sum_{c,k} J[m,c,p] Lambda_m[c,k] J[m,k,p] from a chunk of explicit Jacobians
// J [mc, C, P] (jacobian.hip); Lambda_m = diag(p_m) - p_m p_m^T (laplace/curvature/curvature.py:365-372, 428), so
// per sample: sum_c p_c t_c^2 - (sum_c p_c t_c)^2 with t_c = J[m,c,p].  regression (H_lik = None, :429-430): sum_c t_c^2.
__global__ __launch_bounds__(256) void diag_from_jac_kernel(const float* __restrict__ J, const float* __restrict__ probs,
                                                            int64_t mc, int64_t C, int64_t P, int regression,
                                                            float* __restrict__ diag) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int64_t m_begin = int64_t(blockIdx.y) * 8, m_end = min(mc, m_begin + 8);
  float acc = 0.f;
  for (int64_t m = m_begin; m < m_end; ++m) {
    const float* __restrict__ Jm = J + m * C * P + p;
    float s2 = 0.f, s1 = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float t = Jm[c * P];
      const float pc = regression ? 1.f : probs[m * C + c];
      s2 += pc * t * t;
      s1 += pc * t;
    }
    acc += regression ? s2 : s2 - s1 * s1;
  }
  atomicAdd(&diag[p], acc);
}

int diag_from_jacobians(lgnn_ctx* h, const int64_t* idx, int64_t M, float* diag_out, hipStream_t s) {
  const int64_t C = h->dims[h->L], P = h->n_params;
  // chunk of samples whose Jacobians [mc, C, P] take at most a quarter of the workspace cap (the planes of
  // jacobian.hip are chunked against the cap by themselves)
  const int64_t mc_max = std::max<int64_t>(1, std::min<int64_t>(M, (h->ws_limit / 4) / std::max<int64_t>(C * P * 4, 1)));
  LGNN_CALL(h->ws.jac.reserve(size_t(mc_max) * C * P * 4));
  float* J = h->ws.jac.as<float>();
  for (int64_t m0 = 0; m0 < M; m0 += mc_max) {
    const int64_t mc = std::min(mc_max, M - m0);
    LGNN_CALL(jacobians(h, idx + m0, mc, J, nullptr, s));
    hipLaunchKernelGGL(diag_from_jac_kernel, dim3(unsigned(cdiv(P, 256)), unsigned(cdiv(mc, 8))), dim3(256), 0, s, J,
                       h->ws.probs.as<float>() + m0 * C, mc, C, P, h->lik == LGNN_LIK_REGRESSION ? 1 : 0, diag_out);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// ---- empirical / Monte-Carlo Fisher from per-sample gradients (EFInterface, laplace/curvature/curvature.py:435-504;
// GGNInterface with stochastic=True, :343-364): G[m, p] = sum_c r[m, c] J[m, c, p] with the functional gradient
// r = resid_scale * (softmax(f) - onehot(y_seed))  resp.  resid_scale * (f - y_seed) for the regression likelihood.
__global__ void ef_resid_kernel(const float* __restrict__ probs, const float* __restrict__ logits,
                                const int64_t* __restrict__ idx, const void* __restrict__ yseed, int64_t mc, int64_t C,
                                int64_t N, int regression, float resid_scale, float* __restrict__ r, int* __restrict__ bad) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= mc * C) return;
  const int64_t m = t / C, c = t - m * C;
  const int64_t n = idx[m];
  float v = 0.f;
  if (n >= 0 && n < N) {
    if (regression) v = logits[n * C + c] - static_cast<const float*>(yseed)[t];
    else {
      const int64_t ys = static_cast<const int64_t*>(yseed)[m];
      if (ys < 0 || ys >= C) *bad = 2;
      v = probs[t] - (c == ys ? 1.f : 0.f);
    }
  }
  r[t] = resid_scale * v;
}
__global__ __launch_bounds__(256) void grads_from_jac_kernel(const float* __restrict__ J, const float* __restrict__ r,
                                                             int64_t mc, int64_t C, int64_t P, float* __restrict__ G) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t m = blockIdx.y;
  if (p >= P) return;
  const float* __restrict__ Jm = J + m * C * P + p;
  float acc = 0.f;
  for (int64_t c = 0; c < C; ++c) acc += r[m * C + c] * Jm[c * P];
  G[m * P + p] = acc;
}
__global__ __launch_bounds__(256) void sumsq_rows_kernel(const float* __restrict__ G, int64_t mc, int64_t P, float scale,
                                                         float* __restrict__ diag) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int64_t m_begin = int64_t(blockIdx.y) * 32, m_end = min(mc, m_begin + 32);
  float acc = 0.f;
  for (int64_t m = m_begin; m < m_end; ++m) { const float g = G[m * P + p]; acc += g * g; }
  atomicAdd(&diag[p], scale * acc);
}

int feat_views(lgnn_ctx* h, int layer, FeatView& f) {
  f.nrows = h->N;
  // what Linear `layer` multiplies, seen from an output node: propagated input (GCN) or cat (GraphSAGE)
  if (h->kind == LGNN_KIND_GCN) {
    f.base = h->fc.prop_in[layer].as<float>();
    f.ld = h->fc.prop_ld[layer];
    f.width = h->dims[layer];
    f.bias_col = h->fc.rowsum.as<float>();
  } else {
    f.base = h->fc.lin_in_p[layer];
    f.ld = h->fc.lin_in_ld[layer];
    f.width = h->in_dim[layer];
    f.bias_col = nullptr;
  }
  return 0;
}

}  // namespace

int diag_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, uint32_t flags, float* diag_out,
                    float* loss_out, hipStream_t s) {
  (void)flags;
  LGNN_REQUIRE(M > 0 && idx && y && diag_out && loss_out, "empty batch or null pointers");
  LGNN_REQUIRE(h->L >= 1, "no model bound");
  const int L = h->L;
  const int64_t C = h->dims[L];
  const bool regression = h->lik == LGNN_LIK_REGRESSION;
  if (L > 2 || h->extras() || getenv("LGNN_JAC_PLANES") != nullptr) {
    // deeper models (the reference builds them once the breakpoint at gnn/models/base_gnn.py:109 is removed): per-sample
    // Jacobians in chunks + the contraction with Lambda; no closed form
    LGNN_CALL(forward_ensure(h, s));
    LGNN_CALL(batch_prologue(h, idx, y, M, false, false, loss_out, s));
    LGNN_CALL(diag_from_jacobians(h, idx, M, diag_out, s));
    LGNN_CALL(batch_epilogue(h, idx, M, s));
    return 0;
  }
  LGNN_CALL(forward_ensure_aux(h, s));
  // classification: probabilities, loss, q and the id / label checks in ONE launch (diag_prologue_kernel); regression
  // (Lambda = I, no probabilities) keeps the shared prologue
  const bool light = !regression;
  if (!light) LGNN_CALL(batch_prologue(h, idx, y, M, false, false, loss_out, s));
  // regression: Lambda = I, i.e. q = sum_k w_kj^2 and unit weights in the last layer (the interface's factor is the caller's)
  const float* probs = regression ? nullptr : h->ws.probs.as<float>();
  // Samples per workgroup slab.  A thread walks its slab's samples and their neighbours one after the other (dependent
  // loads: a latency chain), so small problems want short slabs -- Cora shape: 0.42 -> 0.2 ms with 16 instead of 64 --
  // while every slab costs one atomic per output element: aim for ~2048 workgroups.
  int64_t tiles = 1;
  if (h->L == 2) tiles = cdiv(h->in_dim[0] + 1, 64) * cdiv(h->dims[1], 64);
  // (measured at the Cora shape: 1 024 - 2 048 workgroups are the optimum -- fewer lengthen the per-workgroup chain,
  //  more multiply the float atomics of the flush: 0.13 ms at 2 048, 0.26 ms at 8 192)
  int64_t slab = std::max<int64_t>(8, std::min<int64_t>(64, cdiv(M * tiles, 2048)));
  // First-layer kernel: 2 = matrix cores (GCN), 1 = LDS-DMA rows + v_readlane operand, 0 = the register-staged kernel of
  // round 2.  GraphSAGE takes 0: its self path needs three q rows and the node's own rows per closing entry, and at the
  // arxiv shape the staged kernel is the faster one there (2.02 against 2.21 ms per batch; at the Cora shape 0.131 / 0.122).
  // LGNN_DIAG_STAGED / LGNN_DIAG_VALU force 0 / 1 for either family (A/B runs, tests).
  const int route = getenv("LGNN_DIAG_STAGED") != nullptr ? 0
                    : getenv("LGNN_DIAG_VALU") != nullptr ? 1 : (h->kind == LGNN_KIND_SAGE ? 0 : 2);
  // the LDS-DMA kernel hides the chain: longer slabs, fewer atomics (~768 workgroups, three per CU)
  if (route != 0) slab = std::max<int64_t>(8, std::min<int64_t>(64, cdiv(M * tiles, 768)));
  if (const char* e = getenv("LGNN_DIAG_SLAB")) slab = std::max<int64_t>(1, std::min<int64_t>(64, atoll(e)));
  const unsigned nslab = unsigned(cdiv(M, slab));

  int64_t off = 0;
  TileReduce red{nullptr, 0, 0, 0, 0, nullptr, nullptr, -1};
  {
    const int64_t H = L == 2 ? h->dims[1] : 0, in0 = h->in_dim[0];
    const int has_self = h->kind == LGNN_KIND_SAGE ? 1 : 0;
    LGNN_CALL(h->ws.misc.reserve(size_t(3) * M * H * 4 + size_t(M) * C * 4));
    float* q = h->ws.misc.as<float>();
    if (light) {
      LGNN_CALL(h->ws.probs.reserve(size_t(M) * C * 4));
      probs = h->ws.probs.as<float>();
      LGNN_REQUIRE(4 * C * 4 <= 60 * 1024, "too many classes for the prologue kernel");
      hipLaunchKernelGGL(diag_prologue_kernel, dim3(unsigned(std::min<int64_t>(cdiv(M, 4), 4096))), dim3(256),
                         size_t(4 * C) * 4, s, h->fc.out.as<float>(), C, idx, static_cast<const int64_t*>(y), M, h->N,
                         L == 2 ? h->W[1] : nullptr, L == 2 ? h->in_dim[1] : 0, H, int64_t(0), has_self ? H : int64_t(0),
                         has_self, h->ws.probs.as<float>(), q, loss_out, h->ws.flags.as<int>());
      LGNN_HIP_CHECK(hipGetLastError());
    } else if (L == 2) {
      hipLaunchKernelGGL(q_kernel, dim3(unsigned(cdiv(M * H, 256))), dim3(256), 0, s, probs, M, C, h->W[1],
                         h->in_dim[1], H, int64_t(0), has_self ? H : int64_t(0), has_self, q);
    }
    if (L == 2) {
      FeatView E;
      feat_views(h, 0, E);
      const dim3 grid{unsigned(cdiv(E.width + 1, 64)), unsigned(cdiv(H, 64)), nslab};
      if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel of the diagonal path (bench.py roofline)
      if (route == 0) {  // the register-staged kernel
        if (has_self)
          hipLaunchKernelGGL(diag_first_layer_kernel<1>, grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val, idx, M, slab,
                             E, h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
        else
          hipLaunchKernelGGL(diag_first_layer_kernel<0>, grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val, idx, M, slab,
                             E, h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
      } else if (route == 1 || has_self) {
        if (has_self)
          hipLaunchKernelGGL((diag_first_layer_dma_kernel<1, 16, 3>), grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val,
                             idx, M, slab, E, h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
        else
          hipLaunchKernelGGL((diag_first_layer_dma_kernel<0, 16, 4>), grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val,
                             idx, M, slab, E, h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
      } else {
        // GCN: the matrix-core kernels; E = [P X | rowsum(P) | 0] is one padded matrix (context.hip build_px)
        LGNN_REQUIRE(E.ld >= E.width + 1 && E.ld % 4 == 0, "internal: P X without its bias column");
        const int64_t ncb = cdiv(E.width + 1, kTileCols);
        if (ncb >= 4 && H % 4 == 0 && getenv("LGNN_DIAG_ATOMIC") == nullptr) {
          // wide inputs: owned output tiles, long slabs, plain-store partials + a fixed-order reduction (no atomics)
          const int64_t gyt = cdiv(H, 64);
          int64_t nsl = std::max<int64_t>(1, std::min<int64_t>(cdiv(256, ncb * gyt), cdiv(M, 16)));  // ~ one workgroup per CU
          int64_t sl = std::min<int64_t>(kTileSlabMax, cdiv(M, nsl));
          if (const char* e = getenv("LGNN_DIAG_SLAB")) sl = std::max<int64_t>(1, std::min<int64_t>(kTileSlabMax, atoll(e)));
          nsl = cdiv(M, sl);
          const int64_t part_ld = ncb * kTileCols;
          LGNN_CALL(h->ws.jac.reserve(size_t(nsl) * H * part_ld * 4));
          float* part = h->ws.jac.as<float>();
          hipLaunchKernelGGL(diag_first_layer_tile_kernel, dim3(unsigned(ncb * nsl), unsigned(gyt)), dim3(256), 0, s,
                             h->P.rowptr, h->P.col, h->P.val, idx, M, sl, int(ncb), E.base, E.ld, E.width, h->N,
                             h->fc.dact0.as<float>(), H, q, part, part_ld);
          LGNN_HIP_CHECK(hipGetLastError());
          if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += M; }
          red = TileReduce{part, part_ld, int(nsl), H, E.width, diag_out, diag_out + H * in0, 0};
          off = H * in0 + H;
          goto first_layer_done;
        }
        const unsigned gx = unsigned(cdiv(E.width + 1, 256)), gy = unsigned(cdiv(H, 64));
        int64_t sl = std::max<int64_t>(4, std::min<int64_t>(64, cdiv(M * gx * gy, 512)));
        if (const char* e = getenv("LGNN_DIAG_SLAB")) sl = std::max<int64_t>(1, std::min<int64_t>(64, atoll(e)));
        hipLaunchKernelGGL((diag_first_layer_mfma_kernel<8, 3>), dim3(gx, gy, unsigned(cdiv(M, sl))), dim3(256), 0, s,
                           h->P.rowptr, h->P.col, h->P.val, idx, M, sl, E.base, E.ld, E.width, h->N,
                           h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
      }
      LGNN_HIP_CHECK(hipGetLastError());
      if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += M; }
      off = H * in0 + H;
    }
  }
first_layer_done:
  {
    FeatView Phi;
    feat_views(h, L - 1, Phi);
    const int64_t slab_last = slab;
    dim3 grid{unsigned(cdiv(Phi.width + 1, 256)), unsigned(C), unsigned(cdiv(M, slab_last))};
    if (red.z0 >= 0) {  // the tile kernel's partials are added up by extra z-slices of this launch
      red.z0 = int(grid.z);
      const int64_t blocks = cdiv(red.H * ((red.F + 4) / 4), 256);
      grid.z += unsigned(cdiv(blocks, int64_t(grid.x) * grid.y));
    }
    const float* wgt = nullptr;
    if (regression) {
      const int64_t Hq = L == 2 ? h->dims[1] : 0;
      LGNN_CALL(h->ws.misc.reserve(size_t(3) * M * Hq * 4 + size_t(M) * C * 4));
      float* ones = h->ws.misc.as<float>() + 3 * M * Hq;
      LGNN_CALL(launch_fill_i32(reinterpret_cast<int32_t*>(ones), M * C, 0x3f800000, s));  // 1.0f
      wgt = ones;
    }
    hipLaunchKernelGGL(diag_last_layer_kernel, grid, dim3(256), 0, s, h->ws.probs.as<float>(), idx, M, C, slab_last, Phi,
                       diag_out + off, diag_out + off + C * Phi.width, wgt, red);
    LGNN_HIP_CHECK(hipGetLastError());
  }
  if (!light) LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

int gram_rows_sgemm(const float* G, int64_t rows, int64_t P, float scale, float* out, hipStream_t s);  // jacobian.hip
int ef_grads_closed_form(lgnn_ctx* h, const int64_t* idx, int64_t M, const float* r, float* G, hipStream_t s);  // jacobian.hip

// Full GGN over all weights (GGNInterface.full, laplace/curvature/curvature.py:374-410; the reference's default backend
// applies a matrix-free GGN to the P columns of the identity, curvlinops/ggn.py:44-75 via laplace/curvature/
// curvlinops.py:110-140): H = sum_n J_n^T Lambda_n J_n with Lambda_n = S_n S_n^T, S[k, c] = sqrt(p_c) (d_kc - p_k), so
// H = X^T X for the rows X[(n, c), :] = sum_k S_n[k, c] J[n, k, :] = sqrt(p_c) (J[n, c, :] - sum_k p_k J[n, k, :]) -- one
// in-place row mixing pass over a chunk of device Jacobians, then the fp32 MFMA Gram kernel (upper sub-tiles) straight
// into the caller's H.  Regression (H_lik = None): X = J.
__global__ __launch_bounds__(256) void ggn_mix_rows_kernel(float* __restrict__ J, const float* __restrict__ probs,
                                                           int64_t mc, int64_t C, int64_t P) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t m = blockIdx.y;
  if (p >= P) return;
  float* __restrict__ Jm = J + m * C * P + p;
  const float* __restrict__ pm = probs + m * C;
  float t = 0.f;
  for (int64_t k = 0; k < C; ++k) t += pm[k] * Jm[k * P];
  for (int64_t c = 0; c < C; ++c) Jm[c * P] = sqrtf(pm[c]) * (Jm[c * P] - t);
}

int full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out, float* loss_out,
                    hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && y && H_out && loss_out, "empty batch or null pointers");
  LGNN_REQUIRE(h->L >= 1, "no model bound");
  LGNN_CALL(forward_ensure(h, s));
  const int64_t C = h->dims[h->L], P = h->n_params;
  LGNN_REQUIRE(P * P < (int64_t(1) << 40), "full GGN: P x P does not fit");
  LGNN_CALL(batch_prologue(h, idx, y, M, false, false, loss_out, s));
  const int64_t mc_max = std::max<int64_t>(1, std::min<int64_t>(M, (h->ws_limit / 4) / std::max<int64_t>(C * P * 4, 1)));
  LGNN_CALL(h->ws.jac.reserve(size_t(mc_max) * C * P * 4));
  float* J = h->ws.jac.as<float>();
  for (int64_t m0 = 0; m0 < M; m0 += mc_max) {
    const int64_t mc = std::min(mc_max, M - m0);
    LGNN_CALL(jacobians(h, idx + m0, mc, J, nullptr, s));
    if (h->lik != LGNN_LIK_REGRESSION) {
      hipLaunchKernelGGL(ggn_mix_rows_kernel, dim3(unsigned(cdiv(P, 256)), unsigned(mc)), dim3(256), 0, s, J,
                         h->ws.probs.as<float>() + m0 * C, mc, C, P);
      LGNN_HIP_CHECK(hipGetLastError());
    }
    LGNN_CALL(launch_gram(J, P, mc * C, P, H_out, s));
  }
  LGNN_CALL(launch_symmetrize_upper(H_out, P, s));
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

int ef_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y_seed, const void* y_loss, int64_t M, float resid_scale,
                  float scale, float* diag_out, float* full_out, float* grads_out, float* loss_out, hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && y_seed, "empty batch or null pointers");
  LGNN_REQUIRE(h->L >= 1, "no model bound");
  LGNN_REQUIRE(diag_out || full_out || grads_out, "nothing to compute");
  LGNN_REQUIRE(!y_loss || loss_out, "loss requested without an output");
  LGNN_CALL(forward_ensure(h, s));
  const int64_t C = h->dims[h->L], P = h->n_params, N = h->N;
  LGNN_CALL(batch_prologue(h, idx, y_loss ? y_loss : y_seed, M, false, false, y_loss ? loss_out : nullptr, s));
  if (diag_out && !full_out && !grads_out && h->L <= 2 && !h->extras() && getenv("LGNN_JAC_PLANES") == nullptr) {
    // diagonal only, <= 2 layers: the closed form of the diagonal GGN with other weights -- no Jacobians at all
    LGNN_CALL(forward_ensure_aux(h, s));
    const int L = h->L;
    const int64_t H = L == 2 ? h->dims[1] : 0;
    LGNN_CALL(h->ws.misc.reserve(size_t(3) * M * std::max<int64_t>(H, 1) * 4 + size_t(2) * M * C * 4));
    float* q = h->ws.misc.as<float>();
    float* r = q + 3 * M * std::max<int64_t>(H, 1);
    float* wl = r + M * C;
    hipLaunchKernelGGL(ef_resid_kernel, dim3(unsigned(cdiv(M * C, 256))), dim3(256), 0, s, h->ws.probs.as<float>(),
                       h->fc.out.as<float>(), idx, y_seed, M, C, N, h->lik == LGNN_LIK_REGRESSION ? 1 : 0, resid_scale, r,
                       h->ws.flags.as<int>());
    hipLaunchKernelGGL(ef_last_weights_kernel, dim3(unsigned(cdiv(M * C, 256))), dim3(256), 0, s, r, M * C, scale, wl);
    LGNN_HIP_CHECK(hipGetLastError());
    int64_t tiles = 1;
    if (L == 2) tiles = cdiv(h->in_dim[0] + 1, 64) * cdiv(H, 64);
    const int64_t slab = std::max<int64_t>(8, std::min<int64_t>(64, cdiv(M * tiles, 2048)));
    const unsigned nslab = unsigned(cdiv(M, slab));
    int64_t off = 0;
    if (L == 2) {
      const int64_t in0 = h->in_dim[0];
      const int has_self = h->kind == LGNN_KIND_SAGE ? 1 : 0;
      hipLaunchKernelGGL(q_ef_kernel, dim3(unsigned(cdiv(M * H, 256))), dim3(256), 0, s, r, M, C, h->W[1], h->in_dim[1], H,
                         int64_t(0), has_self ? H : int64_t(0), has_self, scale, q);
      FeatView E;
      feat_views(h, 0, E);
      const dim3 grid{unsigned(cdiv(E.width + 1, 64)), unsigned(cdiv(H, 64)), nslab};
      if (has_self)
        hipLaunchKernelGGL(diag_first_layer_kernel<1>, grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val, idx, M, slab, E,
                           h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
      else
        hipLaunchKernelGGL(diag_first_layer_kernel<0>, grid, dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val, idx, M, slab, E,
                           h->fc.dact0.as<float>(), H, q, diag_out, diag_out + H * in0);
      LGNN_HIP_CHECK(hipGetLastError());
      off = H * in0 + H;
    }
    FeatView Phi;
    feat_views(h, L - 1, Phi);
    hipLaunchKernelGGL(diag_last_layer_kernel, dim3(unsigned(cdiv(Phi.width + 1, 256)), unsigned(C), nslab), dim3(256), 0, s,
                       h->ws.probs.as<float>(), idx, M, C, slab, Phi, diag_out + off, diag_out + off + C * Phi.width, wl,
                       TileReduce{nullptr, 0, 0, 0, 0, nullptr, nullptr, -1});
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(batch_epilogue(h, idx, M, s));
    return 0;
  }
  // <= 2 layers: the gradients come straight from the closed form (M * P floats); deeper models contract Jacobians
  const bool closed = h->L <= 2 && !h->extras() && getenv("LGNN_JAC_PLANES") == nullptr;
  const int64_t jrows = closed ? 0 : C;
  const int64_t mc_max = std::max<int64_t>(1, std::min<int64_t>(M, (h->ws_limit / 4) / std::max<int64_t>((jrows + 1) * P * 4, 1)));
  LGNN_CALL(h->ws.jac.reserve(size_t(mc_max) * (jrows + 1) * P * 4 + size_t(mc_max) * C * 4));
  float* J = h->ws.jac.as<float>();
  float* G = J + mc_max * jrows * P;
  float* r = G + mc_max * P;
  for (int64_t m0 = 0; m0 < M; m0 += mc_max) {
    const int64_t mc = std::min(mc_max, M - m0);
    if (!closed) LGNN_CALL(jacobians(h, idx + m0, mc, J, nullptr, s));
    const void* ys = h->lik == LGNN_LIK_REGRESSION ? static_cast<const void*>(static_cast<const float*>(y_seed) + m0 * C)
                                                   : static_cast<const void*>(static_cast<const int64_t*>(y_seed) + m0);
    hipLaunchKernelGGL(ef_resid_kernel, dim3(unsigned(cdiv(mc * C, 256))), dim3(256), 0, s, h->ws.probs.as<float>() + m0 * C,
                       h->fc.out.as<float>(), idx + m0, ys, mc, C, N, h->lik == LGNN_LIK_REGRESSION ? 1 : 0, resid_scale, r,
                       h->ws.flags.as<int>());
    if (closed) LGNN_CALL(ef_grads_closed_form(h, idx + m0, mc, r, G, s));
    else hipLaunchKernelGGL(grads_from_jac_kernel, dim3(unsigned(cdiv(P, 256)), unsigned(mc)), dim3(256), 0, s, J, r, mc, C, P, G);
    LGNN_HIP_CHECK(hipGetLastError());
    if (grads_out) LGNN_HIP_CHECK(hipMemcpyAsync(grads_out + m0 * P, G, size_t(mc) * P * 4, hipMemcpyDeviceToDevice, s));
    if (diag_out)
      hipLaunchKernelGGL(sumsq_rows_kernel, dim3(unsigned(cdiv(P, 256)), unsigned(cdiv(mc, 32))), dim3(256), 0, s, G, mc, P,
                         scale, diag_out);
    if (full_out) LGNN_CALL(gram_rows_sgemm(G, mc, P, scale, full_out, s));
    LGNN_HIP_CHECK(hipGetLastError());
  }
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

// phi~ of the batch rows: out [M, D + 1] = [last layer's input features at the node | bias scale]
// (last_layer_jacobians, laplace/curvature/curvature.py:132-167: J_n = [I_C (x) phi_n^T | s_n I_C])
int lastlayer_features(lgnn_ctx* h, const int64_t* idx, int64_t M, float* out, float* f_out, hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && out, "empty batch or null pointers");
  LGNN_CALL(forward_ensure_aux(h, s));
  FeatView Phi;
  feat_views(h, h->L - 1, Phi);
  const int64_t D1 = Phi.width + 1;
  hipLaunchKernelGGL(ll_build_phi_kernel, dim3(unsigned(std::min<int64_t>(cdiv(M * D1, 256), 8192))), dim3(256), 0, s, idx, M,
                     Phi, D1, out);
  LGNN_HIP_CHECK(hipGetLastError());
  if (f_out)
    LGNN_CALL(launch_gather_rows(h->fc.out.as<float>(), h->dims[h->L], h->N, idx, M, h->dims[h->L], f_out,
                                 h->ws.flags.as<int>() + 2, s));
  return 0;
}

// Pair-major accumulation: S [Q][D][D] (upper sub-tiles of each block) and Sb [Q][D + 1] are the CALLER's buffers and are
// added to -- a fit accumulates all its batches there and places them into H once (lastlayer_pairs_place), instead of one
// placement (1.2 GB of read-modify-write) + mirror pass per batch; a data-parallel caller all-reduces the pair buffers,
// half the bytes of H.
int lastlayer_pairs_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* S, float* Sb,
                               float* loss_out, hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && y && S && Sb && loss_out, "empty batch or null pointers");
  LGNN_REQUIRE(h->lik == LGNN_LIK_CLASSIFICATION, "last-layer full GGN kernels: classification likelihood");
  LGNN_CALL(forward_ensure_aux(h, s));
  const int L = h->L;
  const int64_t C = h->dims[L];
  LGNN_CALL(batch_prologue(h, idx, y, M, false, false, loss_out, s));
  const float* probs = h->ws.probs.as<float>();
  FeatView Phi;
  feat_views(h, L - 1, Phi);
  const int64_t D = Phi.width, D1 = D + 1;
  const int64_t Q = C * (C + 1) / 2, ldp = cdiv(D1, 4) * 4;
  // chunks of pairs: the row-scale / weight vectors [qc][M] live in the workspace
  const int64_t per_pair = (2 * M + 1) * 4;
  const int64_t qc_max = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(Q, 32768), h->ws_limit / per_pair));
  LGNN_CALL(h->ws.planes_a.reserve(size_t(M) * ldp * 4 + size_t(qc_max) * (2 * M + 1) * 4));
  h->ws.planes_a_zero_ptr = nullptr;
  float* PhiM = h->ws.planes_a.as<float>();
  float* rs = PhiM + M * ldp;
  float* wsg = rs + qc_max * M;
  float* zsign = wsg + qc_max * M;
  hipLaunchKernelGGL(ll_build_phi_kernel, dim3(unsigned(std::min<int64_t>(cdiv(M * ldp, 256), 8192))), dim3(256), 0, s,
                     idx, M, Phi, ldp, PhiM);
  LGNN_HIP_CHECK(hipGetLastError());
  for (int64_t q0 = 0; q0 < Q; q0 += qc_max) {
    const int64_t qc = std::min(qc_max, Q - q0);
    hipLaunchKernelGGL(ll_pair_weights_kernel, dim3(unsigned(std::min<int64_t>(cdiv(M, 256), 64)), unsigned(qc)),
                       dim3(256), 0, s, probs, idx, Phi, M, C, q0, rs, wsg, zsign);
    LGNN_HIP_CHECK(hipGetLastError());
    // S[q] += sign_q * (diag(rs_q) Phi)^T (diag(rs_q) Phi) over the D feature columns
    if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel of the last-layer path (bench.py roofline)
    LGNN_CALL(launch_gram_batched(PhiM, ldp, M, D, S + q0 * D * D, D * D, qc, rs, zsign, 1.0f, s));
    if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += qc; }
    // bias column: Sb[q][j] += sum_m w_qm s_m phi~[m][j], one library GEMM [qc x M] * [M x D1]
    LGNN_CALL(ll_bias_gemm(wsg, PhiM, Sb + q0 * D1, qc, M, D1, ldp, s, 1.0f));
  }
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

// H_out [P, P] += the blocks of the pair-major accumulators (upper triangle), then the mirror pass
int lastlayer_pairs_place(lgnn_ctx* h, const float* S, const float* Sb, float* H_out, hipStream_t s) {
  LGNN_REQUIRE(S && Sb && H_out && h->L > 0, "null pointers / no model bound");
  const int64_t C = h->dims[h->L], D = h->in_dim[h->L - 1], P = C * D + C;
  const int64_t Q = C * (C + 1) / 2;
  const unsigned nt = unsigned(cdiv(D, 32));
  for (int64_t q0 = 0; q0 < Q; q0 += 32768) {
    const int64_t qc = std::min<int64_t>(32768, Q - q0);
    hipLaunchKernelGGL(ll_place_tiles_kernel, dim3(nt, nt, unsigned(qc)), dim3(256), 0, s, S + q0 * D * D, D, C, q0, H_out);
  }
  hipLaunchKernelGGL(ll_place_bias_kernel, dim3(unsigned(std::min<int64_t>(cdiv(Q * (D + 1), 256), 4096))), dim3(256), 0, s,
                     Sb, D, C, int64_t(0), Q, H_out);
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_CALL(launch_symmetrize_upper(H_out, P, s));
  return 0;
}

// One batch straight into H (the entry point of the first round): pair buffers from the workspace, zeroed, accumulated,
// placed and mirrored in this call.
int lastlayer_full_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, float* H_out,
                              float* loss_out, hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && y && H_out && loss_out, "empty batch or null pointers");
  LGNN_REQUIRE(h->L > 0, "no model bound");
  const int64_t C = h->dims[h->L], D = h->in_dim[h->L - 1], D1 = D + 1, Q = C * (C + 1) / 2;
  LGNN_CALL(h->ws.planes_b.reserve(size_t(Q) * (D * D + D1) * 4));
  float* S = h->ws.planes_b.as<float>();
  float* Sb = S + Q * D * D;
  LGNN_HIP_CHECK(hipMemsetAsync(S, 0, size_t(Q) * (D * D + D1) * 4, s));
  LGNN_CALL(lastlayer_pairs_accumulate(h, idx, y, M, S, Sb, loss_out, s));
  return lastlayer_pairs_place(h, S, Sb, H_out, s);
}

}  // namespace lgnn
