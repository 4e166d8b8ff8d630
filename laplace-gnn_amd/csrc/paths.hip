// First-layer gradient covariance B_0 of a 2-layer GCN WITHOUT class planes: the headline KFAC path.
//
// Reference: KFACLinearOperator._compute_loss_and_backward / _accumulate_gradient_covariance (curvlinops/kfac.py:607-661,
// 777-817) run one dense backward pass per class column c of the loss-Hessian square root through the model
// (gnn/models/base_gnn.py:141-156, gnn/models/layers.py:45-46) and add g^T g at the first Linear's output.
//
// The route this file replaces materialised the C right-hand sides as class-major planes
//     U[c][v][:] = [h_1[v] > 0] * (g1_c[v] W_1)          (40 planes x 173 MB at the arxiv shape, written by backgemm.hip)
// and gathered them back through P^T inside the fused SpMM^T -> Gram kernel (fused256.hip): 55 GB per batch through the
// Infinity Cache, 7.1x the algorithmic bytes, the gather and not the matrix pipes setting the time.
//
// The seed block is diagonal + rank 2 (kfac.hip, seed_spmm_gram_kernel):  V_m[k, c] = alpha_c d_kc - beta_c u_k - gamma_c p_k,
// hence  V_m[:, c]^T W_1 = alpha_c^m W_1[c, :] - beta_c^m b_m - gamma_c^m g_m  with the per-sample H-vectors b_m = u_m^T W_1,
// g_m = p_m^T W_1, and for a destination node n the C x H block of all its class rows is a sum over the batch's 2-hop paths
// j = (n <- v_j <- m_j), weight w_j = P^T[n, v_j] P^T[v_j, m_j]:
//     Y[n] = W_1 (.) (A_alpha Mk) + A_beta (B_k (.) Mk) + A_gamma (G_k (.) Mk)
//       A_*[c, j] = w_j * (alpha, -beta, -gamma)_c^{m_j}   (C x K),   Mk[j, :] = ReLU mask bits of v_j  (K x H, 0 / 1),
//       B_k[j, :] = b_{m_j},  G_k[j, :] = g_{m_j}                    (K x H, rows of a 20 MB per-batch table: L2 / MALL resident)
//     B_0 += Y[n]^T Y[n]
// (oracle: kfac_first_layer_B_by_paths, pinned to the reference's goldens).  ~13 paths per node at the arxiv shape, ~2.8 KB
// gathered per path instead of 40 KB per edge; three C x K x H products on the matrix pipes (+ ~1/3 of the Gram's MFMA work)
// buy the removal of the planes' gather.
//
// Kernels (per mini-batch):
//   path_tables_kernel   per-sample coefficient rows (alpha, -beta, -gamma) and the rows (u, p) whose product with W_1 (one
//                        small GEMM, kernels.hip) gives b_m, g_m
//   path_count / fill    R = P^T[:, batch] as CSR over v (counting sort: count, rocPRIM scan, fill): the batch neighbours of v
//   path_list_kernel     the batch's 2-hop paths per destination node as CSR over n (count, rocPRIM scan, fill; one wave per node)
//   ybuild_pipe_kernel   one persistent workgroup per CU, one wave per (32 classes x 64 columns): per node, windows of 16 paths
//                        whose operands (rows b_m, g_m, coefficient rows: 1 KiB each) arrive by LDS-DMA into a double buffer
//                        one node ahead of the products; three products per window on v_mfma_f32_32x32x2_f32 (A = weighted
//                        coefficient rows, B = mask bits / masked table rows, all from LDS), W_1 folded in from registers,
//                        Y[n] (R x H floats, contiguous) streamed to HBM                      -- fp32 MFMA / HBM write bound
//   ybuild_kernel        the same products with the paths enumerated on the fly (taken only when the path list does not fit)
//   gram256_stream_kernel  S += Y^T Y over N*R rows of 1 KiB: one persistent 512-thread workgroup per CU, ALL EIGHT waves on
//                        the matrix pipes (36 upper 32 x 32 sub-tiles dealt 5 + 4 to the two waves of a SIMD), row blocks of
//                        32 rows arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, issued by the MFMA
//                        waves themselves) into a 3-slot ring: one raw s_barrier and one counted vmcnt wait per block
//                                                                                                           -- fp32 MFMA bound
#include "device_utils.h"
#include "gram256.h"
#include "lgnn_internal.h"

namespace lgnn {

namespace {

constexpr int kCoefStride = 64;  // classes per coefficient row (zero padded): two 32-row MFMA tiles
constexpr int kCoefRow = 256;    // floats per sample in the coefficient table: (alpha | -beta | -gamma | pad) = 1 KiB, one LDS-DMA piece
constexpr int kPathWindow = 128; // paths staged in LDS per accumulation window

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per batch sample (first occurrences only; a node listed t times carries t in its R weights).
// mode: 0 upstream seeds, 1 fork exact, 2 regression (V = sqrt(2) I).
__global__ __launch_bounds__(256) void path_tables_kernel(const float* __restrict__ probs, const float* __restrict__ logits,
                                                          const int64_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                                          int64_t M, int64_t N, int C, int mode, float* __restrict__ coef,
                                                          float* __restrict__ up) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t n = idx[m];
  float* __restrict__ cm = coef + m * kCoefRow;
  float* __restrict__ um = up + m * C;
  float* __restrict__ pm = up + (M + m) * C;
  const bool own = n >= 0 && n < N && pos[n] == int32_t(m);
  float pk = 0.f, fk = 0.f;
  if (own && lane < C && mode != 2) { pk = probs[m * C + lane]; fk = logits[n * C + lane]; }
  const float mb = wsum(pk * fk);  // same summation order as seed_kernel / seed_spmm_gram_kernel
  const float sp = sqrtf(pk), t = fk - mb;
  float al = 0.f, be = 0.f, ga = 0.f, u = 0.f;
  if (own && lane < C) {
    if (mode == 2) al = 1.41421356237309515f;
    else if (mode == 1) { al = sp * (1.f + 0.5f * t); be = sp; ga = 0.5f * sp * t; u = pk * (1.f + t); }
    else { al = sp; be = sp; u = pk; }
  }
  cm[lane] = al;                       // lanes >= C write the zero padding
  cm[kCoefStride + lane] = -be;
  cm[2 * kCoefStride + lane] = -ga;
  cm[3 * kCoefStride + lane] = 0.f;
  if (lane < C) { um[lane] = u; pm[lane] = own ? pk : 0.f; }
}

// R = P^T[:, batch]: for every distinct batch node u (its first position m) and every entry (v, val) of row u of P.
template <bool FILL>
__global__ __launch_bounds__(256) void path_r_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N,
                                                     const int32_t* __restrict__ pos, const int32_t* __restrict__ mult,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const float* __restrict__ val, int32_t* __restrict__ cnt,
                                                     const int32_t* __restrict__ rptr, int32_t* __restrict__ r_m,
                                                     float* __restrict__ r_w) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int64_t u = idx[m];
  if (u < 0 || u >= N || pos[u] != int32_t(m)) return;  // invalid ids are flagged by mark_batch_kernel
  const float tm = FILL ? float(mult[m]) : 0.f;
  const int32_t e = rowptr[u + 1];
  for (int32_t p = rowptr[u] + lane; p < e; p += 64) {
    const int32_t v = col[p];
    const int32_t k = atomicAdd(&cnt[v], 1);
    if constexpr (FILL) {
      const int32_t slot = rptr[v] + k;
      r_m[slot] = int32_t(m);
      r_w[slot] = val[p] * tm;
    }
  }
}

// The batch's 2-hop paths per destination node, materialised once per batch (count, rocPRIM scan, fill): the irregular
// three-level walk (row of P^T -> R pointers -> R entries) runs here with one wave per node and tens of thousands of waves
// in flight, so that ybuild_kernel's own chain is just  pointer -> entries -> table rows.
// One wave per node n:  cnt[n] = sum over the entries (v, pv) of row n of P^T of |R[v]|.
template <bool FILL>
__global__ __launch_bounds__(256) void path_list_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const float* __restrict__ val, int64_t N,
                                                        const int32_t* __restrict__ rptr, const int32_t* __restrict__ r_m,
                                                        const float* __restrict__ r_w, int32_t* __restrict__ pcnt,
                                                        const int32_t* __restrict__ pptr, int64_t cap,
                                                        int32_t* __restrict__ pm, int32_t* __restrict__ pv,
                                                        float* __restrict__ pw) {
  const int lane = threadIdx.x & 63;
  const int64_t n = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  if constexpr (FILL) {
    if (int64_t(pptr[N]) > cap) return;  // the list does not fit its buffer: ybuild enumerates on the fly (same results)
  }
  const int32_t s = rowptr[n], e = rowptr[n + 1];
  int32_t run = FILL ? pptr[n] : 0;
  for (int32_t base = s; base < e; base += 64) {
    int32_t v = 0, r0 = 0, cnt = 0;
    float pval = 0.f;
    if (base + lane < e) {
      v = col[base + lane];
      pval = val[base + lane];
      r0 = rptr[v];
      cnt = rptr[v + 1] - r0;
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if constexpr (FILL) {
      const int32_t off = run + incl - cnt;
      for (int k = 0; k < cnt; ++k) {
        pm[off + k] = r_m[r0 + k];
        pv[off + k] = v;
        pw[off + k] = pval * r_w[r0 + k];
      }
    }
    run += __shfl(incl, 63);
  }
  if (!FILL && lane == 0) pcnt[n] = run;
}

struct YArgs {
  const int32_t* rowptr; const int32_t* col; const float* val;  // P^T
  const int32_t* rptr; const int32_t* r_m; const float* r_w;    // R = P^T[:, batch]
  const int32_t* pptr; const int32_t* pm; const int32_t* pv; const float* pw;  // the paths per node (when they fit `cap`)
  int64_t cap;
  const float* coef;        // [M][256]: (alpha | -beta | -gamma | 0) x 64 classes
  const float* zeros;       // >= 1 KiB of zeros
  const float* bg;          // [2 M][H]: rows b_m, then rows g_m
  const uint32_t* mask;     // [N][mask_words] ReLU bits of h_1
  int mask_words;
  const float* W1;          // [C][w1_ld]: the H columns Y is multiplied with (GraphSAGE: the neighbour half of W_1)
  int w1_ld;
  float* Y;                 // [N][R][H]
  int64_t N, M;             // all nodes (pptr has N + 1 entries); rows of a table half
  int64_t n0, n1;           // the destination nodes this launch visits: [n0, n1)
  const int32_t* list;      // optional: the nodes of that range that have a path (relative to n0), ...
  const int32_t* n_list;    // ... and how many (device side: no host round trip); null: every node of the range
  int H, c0, R;
  int no_bg;                // regression / nothing but the diagonal term: the beta / gamma products vanish
};

// Paths staged per window.  26.6 % of the arxiv-shaped nodes have more than 16 paths, 12 % more than 20 (mean 13.7): every
// further window of a node is restaged in place, two barriers and an exposed copy.  20 x (2 x 1 KiB table rows + 768 B
// coefficients + 32 B mask) = 55.6 KiB per window; two of them and a 40-row Y tile fill the 160 KiB of a CU.
constexpr int kWin = 20;
constexpr int kCoefLds = 3 * kCoefStride;  // floats of a coefficient row that are staged (the table's rows are 1 KiB apart)

struct YWin {
  float bg[kWin][2][256];       // rows b_m, g_m as they sit in the table (the mask is applied when they are read)
  float coef[kWin][kCoefLds];   // (alpha | -beta | -gamma) of the path's sample (the path weight is applied when read)
  uint32_t mask[kWin][8];       // ReLU bits of the path's middle node v
};
struct YMeta {                  // the window's triples
  int32_t m[kWin], v[kWin];
  float w[kWin];
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void lds_dma16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                            reinterpret_cast<uintptr_t>(lds_dst)), 16, 0, 0);
}

// Start the LDS-DMA copies of a window of kw <= kWin paths (triples in `mt`): three 1 KiB pieces per path (row b_m, row g_m,
// the coefficient row), one wave instruction each, no data registers.  Asynchronous: the consumer waits on vmcnt + a barrier.
__device__ __forceinline__ void stage_dma(const YArgs& a, YWin& win, const YMeta& mt, int kw, int wave, int nwaves, int lane) {
  const int kw2 = (kw + 1) & ~1;  // the last MFMA step reads an even number of paths: the odd one out is staged as zeros
  const bool lane_ok = 4 * lane < a.H;
  const int npieces = kw2 * 3;
  for (int q = wave; q < npieces; q += nwaves) {   // q, j, kind are wave uniform
    const int j = q / 3, kind = q - 3 * j;
    const float* src = a.zeros;
    float* dst = kind < 2 ? &win.bg[j][kind][0] : &win.coef[j][0];
    if (j < kw) {
      const int64_t mj = __builtin_amdgcn_readfirstlane(mt.m[j]);
      if (kind == 2) src = a.coef + mj * kCoefRow + 4 * lane;
      else if (lane_ok && !a.no_bg) src = a.bg + ((kind ? a.M : 0) + mj) * a.H + 4 * lane;
    }
    // (a coefficient row is 768 bytes in LDS: 48 lanes copy, the others would land in the next path's row)
    if (kind < 2 || lane < kCoefLds / 4) lds_dma16(src, dst);
  }
}
// the mask word (j = tid >> 3, word = tid & 7) of the window's paths, for threads tid < 8 * kw2
__device__ __forceinline__ uint32_t load_mask_word(const YArgs& a, const YMeta& mt, int kw, int tid) {
  const int j = tid >> 3, wd = tid & 7;
  return (j < kw && wd < a.mask_words) ? a.mask[int64_t(mt.v[j]) * a.mask_words + wd] : 0u;
}

// The three products of one staged window: wave (rt, cg), lane l: A row i = l & 31 (class), B column = l & 31, k = l >> 5.
// The LDS operands of step ks + 1 are read before the six MFMAs of step ks are issued (their latency hides behind 384 cycles
// of matrix work instead of stalling every step).
struct YOps { float aa, ab, ag, mf[2], bb[2], gg[2]; };
__device__ __forceinline__ void y_load_ops(const YWin& win, const YMeta& mt, int kw, int ks, int half, int cls,
                                           const int (&colv)[2], const bool (&col_ok)[2], YOps& o) {
  const int j = min(2 * ks + half, kWin - 1);   // (past the window's end: a valid row, weight 0)
  const float wj = 2 * ks + half < kw ? mt.w[j] : 0.f;
  o.aa = wj * win.coef[j][cls]; o.ab = wj * win.coef[j][kCoefStride + cls]; o.ag = wj * win.coef[j][2 * kCoefStride + cls];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const uint32_t word = win.mask[j][colv[ct] >> 5];
    o.mf[ct] = (col_ok[ct] && 2 * ks + half < kw && ((word >> (colv[ct] & 31)) & 1u)) ? 1.f : 0.f;
    o.bb[ct] = o.mf[ct] * win.bg[j][0][colv[ct]];
    o.gg[ct] = o.mf[ct] * win.bg[j][1][colv[ct]];
  }
}
__device__ __forceinline__ void y_mfma_ops(const YOps& o, bool no_bg, f32x16 (&t1)[2], f32x16 (&y2)[2]) {
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    t1[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.aa, o.mf[ct], t1[ct], 0, 0, 0);
    if (!no_bg) {
      y2[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ab, o.bb[ct], y2[ct], 0, 0, 0);
      y2[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ag, o.gg[ct], y2[ct], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void mfma_window(const YWin& win, const YMeta& mt, int kw, int cls, const int (&colv)[2],
                                            const bool (&col_ok)[2], int half, bool no_bg, f32x16 (&t1)[2], f32x16 (&y2)[2]) {
  const int nks = (kw + 1) >> 1;
  if (nks == 0) return;
  YOps oa, ob;
  y_load_ops(win, mt, kw, 0, half, cls, colv, col_ok, oa);
  for (int ks = 0; ks < nks; ks += 2) {
    y_load_ops(win, mt, kw, ks + 1, half, cls, colv, col_ok, ob);  // (a step past the end multiplies zeros)
    __builtin_amdgcn_sched_barrier(0);
    y_mfma_ops(oa, no_bg, t1, y2);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < nks) {
      y_load_ops(win, mt, kw, ks + 2, half, cls, colv, col_ok, oa);
      __builtin_amdgcn_sched_barrier(0);
      y_mfma_ops(ob, no_bg, t1, y2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The second row tile (classes 32 .. 47 of the chunk) on v_mfma_f32_16x16x4_f32: a 16-class tile has no padding rows to multiply
// (C = 40: a 32-row tile would spend three quarters of its work on rows 40 .. 63) and takes half the matrix-pipe time per path.
// Lane l: A row i = l & 15 (class), B column = l & 15 (four 16-column tiles per wave), k = l >> 4 (four paths per step);
// C / D: col = l & 15, row = 4 (l >> 4) + r.  Dependent MFMAs on one accumulator are four instructions apart (latency 40 > issue 32).
using f32x4v = __attribute__((ext_vector_type(4))) float;
struct YOps16 { float aa, ab, ag, mf[4], bb[4], gg[4]; };
__device__ __forceinline__ void y_load_ops16(const YWin& win, const YMeta& mt, int kw, int ks, int lane, int cls16, int cg, int H,
                                             YOps16& o) {
  const int jr = 4 * ks + (lane >> 4);
  const bool valid = jr < kw;
  const int j = valid ? jr : 0;  // (a staged row: what sits past the window's end may not be finite)
  const float wj = valid ? mt.w[j] : 0.f;
  o.aa = wj * win.coef[j][cls16]; o.ab = wj * win.coef[j][kCoefStride + cls16]; o.ag = wj * win.coef[j][2 * kCoefStride + cls16];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int col = 64 * cg + 16 * ct + (lane & 15);
    const int colc = col < H ? col : 0;
    const uint32_t word = win.mask[j][colc >> 5];
    o.mf[ct] = (valid && col < H && ((word >> (colc & 31)) & 1u)) ? 1.f : 0.f;
    o.bb[ct] = o.mf[ct] * win.bg[j][0][colc];
    o.gg[ct] = o.mf[ct] * win.bg[j][1][colc];
  }
}
__device__ __forceinline__ void y_mfma_ops16(const YOps16& o, bool no_bg, f32x4v (&t1)[4], f32x4v (&y2)[4]) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) t1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.aa, o.mf[ct], t1[ct], 0, 0, 0);
  if (!no_bg) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) y2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ab, o.bb[ct], y2[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) y2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ag, o.gg[ct], y2[ct], 0, 0, 0);
  }
}
__device__ __forceinline__ void mfma_window16(const YWin& win, const YMeta& mt, int kw, int lane, int cls16, int cg, int H,
                                              bool no_bg, f32x4v (&t1)[4], f32x4v (&y2)[4]) {
  const int nks = (kw + 3) >> 2;
  if (nks == 0) return;
  YOps16 oa, ob;
  y_load_ops16(win, mt, kw, 0, lane, cls16, cg, H, oa);
  for (int ks = 0; ks < nks; ks += 2) {
    y_load_ops16(win, mt, kw, ks + 1, lane, cls16, cg, H, ob);  // (past the end: zero weights on a staged row)
    __builtin_amdgcn_sched_barrier(0);
    y_mfma_ops16(oa, no_bg, t1, y2);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < nks) {
      y_load_ops16(win, mt, kw, ks + 2, lane, cls16, cg, H, oa);
      __builtin_amdgcn_sched_barrier(0);
      y_mfma_ops16(ob, no_bg, t1, y2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Role of a wave: (rt, cg) owns the 32-class row tile rt (classes c0 + 32 rt ...) and the columns [64 cg, 64 cg + 64) of
// Y[n]: two 32 x 32 accumulator tiles for the alpha product and two for the beta / gamma products; waves w and w + 4 (the
// two row tiles of one column group) share a SIMD.
struct YRole {
  int lane, li, half, wave, nwaves, cg, rt, cls;
  int colv[2];
  bool col_ok[2];
};
__device__ __forceinline__ YRole y_role(const YArgs& a, bool fixed8 = false) {
  YRole r;
  const int tid = threadIdx.x;
  r.lane = tid & 63; r.li = r.lane & 31; r.half = r.lane >> 5;
  r.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  r.nwaves = blockDim.x >> 6;
  const int ncg = fixed8 ? 4 : (a.H + 63) >> 6;  // fused kernel: always 4 x 2 roles (waves w and w + 4 share a SIMD)
  r.cg = r.wave % ncg; r.rt = r.wave / ncg;
  // class of this lane's A-operand row (i = lane & 31), clamped into the zero-padded coefficient row; rows past the class
  // range are computed on whatever sits there and never stored
  r.cls = min(a.c0 + 32 * r.rt + r.li, kCoefStride - 1);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    r.colv[ct] = 64 * r.cg + 32 * ct + r.li;
    r.col_ok[ct] = r.colv[ct] < a.H;
    if (!r.col_ok[ct]) r.colv[ct] = 0;
  }
  return r;
}

// One node's products into the LDS tile `ytile` [rows][256]: the first window is in win[b] (meta slot ms), further windows
// (hubs) are restaged in place -- every wave of the workgroup takes part in the barriers of that loop, whatever its role.
// RT1 = false: the wave owns classes [0, 32) x 64 columns (32 x 32 x 2 MFMA); RT1 = true: classes [32, 48) x 64 columns
// (16 x 16 x 4 MFMA).  W_1's slice of the wave stays in registers for the whole launch.
#ifdef LGNN_DEV  // make DEV=1: per-wave cycle counts of the fused kernel's phases (s_memtime), printed by paths_phase_report()
__device__ unsigned long long g_phase[8][8];
struct Ph { unsigned long long t, acc[8]; };
#define PH_ARG , Ph& ph
#define PH_PASS , ph
#define PH_MARK(k) do { const unsigned long long ph_n = __builtin_amdgcn_s_memtime(); ph.acc[k] += ph_n - ph.t; ph.t = ph_n; } while (0)
#else
#define PH_ARG
#define PH_PASS
#define PH_MARK(k)
#endif

template <bool RT1>
struct YRegs {
  float w1r[RT1 ? 4 : 2][RT1 ? 4 : 16];
};
template <bool RT1>
__device__ __forceinline__ void y_load_w1(const YArgs& a, const YRole& ro, bool path_wave, YRegs<RT1>& g) {
  const int H = a.H;
  if constexpr (!RT1) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int colc = 64 * ro.cg + 32 * ct + ro.li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 4 * ro.half + (r & 3) + 8 * (r >> 2);
        g.w1r[ct][r] = (path_wave && row < a.R && colc < H) ? a.W1[int64_t(a.c0 + row) * a.w1_ld + colc] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int colc = 64 * ro.cg + 16 * ct + (ro.lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 + 4 * (ro.lane >> 4) + r;
        g.w1r[ct][r] = (path_wave && row < a.R && colc < H) ? a.W1[int64_t(a.c0 + row) * a.w1_ld + colc] : 0.f;
      }
    }
  }
}
template <bool RT1>
__device__ __forceinline__ void y_node_products(const YArgs& a, YWin (&win)[2], YMeta (&meta)[4], float (*ytile)[256],
                                                const YRole& ro, bool path_wave, const YRegs<RT1>& g, int b, int ms, int kwc,
                                                int32_t p0c, int32_t p1c PH_ARG) {
  const int tid = threadIdx.x, H = a.H, lane = ro.lane;
  const bool no_bg = a.no_bg != 0;
  const int cls16 = min(a.c0 + 32 + (lane & 15), kCoefStride - 1);
  f32x16 t1[2], y2[2];
  f32x4v u1[4], u2[4];
  if constexpr (!RT1) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { t1[ct][r] = 0.f; y2[ct][r] = 0.f; }
    if (path_wave) mfma_window(win[b], meta[ms], kwc, ro.cls, ro.colv, ro.col_ok, ro.half, no_bg, t1, y2);
  } else {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { u1[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; u2[ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    if (path_wave) mfma_window16(win[b], meta[ms], kwc, lane, cls16, ro.cg, H, no_bg, u1, u2);
  }
  PH_MARK(2);  // first window's products
  for (int32_t wb = p0c + kWin; wb < p1c; wb += kWin) {  // hubs: further windows, restaged in place (not overlapped)
    const int kw = min(kWin, p1c - wb);
    lds_barrier();
    if (tid < kw) { meta[3].m[tid] = a.pm[wb + tid]; meta[3].v[tid] = a.pv[wb + tid]; meta[3].w[tid] = a.pw[wb + tid]; }
    lds_barrier();
    stage_dma(a, win[b], meta[3], kw, ro.wave, 8, lane);
    if (tid < 8 * kWin) win[b].mask[tid >> 3][tid & 7] = load_mask_word(a, meta[3], kw, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (path_wave) {
      if constexpr (!RT1) mfma_window(win[b], meta[3], kw, ro.cls, ro.colv, ro.col_ok, ro.half, no_bg, t1, y2);
      else mfma_window16(win[b], meta[3], kw, lane, cls16, ro.cg, H, no_bg, u1, u2);
    }
  }
  PH_MARK(3);  // further windows (restaged)
  // Y[n] into the LDS tile (readers of the previous node's tile passed this node's first barrier)
  if (path_wave) {
    if constexpr (!RT1) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int colc = 64 * ro.cg + 32 * ct + ro.li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 4 * ro.half + (r & 3) + 8 * (r >> 2);
          if (colc < H && row < a.R) ytile[row][colc] = g.w1r[ct][r] * t1[ct][r] + y2[ct][r];
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int colc = 64 * ro.cg + 16 * ct + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 + 4 * (lane >> 4) + r;
          if (colc < H && row < a.R) ytile[row][colc] = g.w1r[ct][r] * u1[ct][r] + u2[ct][r];
        }
      }
    }
  }
}

// The part of the node loop both persistent kernels share: prologue (node 0's window in flight, node 1's triples in
// registers) and, per node, the top of the iteration (wait, publish the next triples, barrier, start the next window).
struct YPipe {
  int32_t p0c, p1c, p0n, p1n, p0f, p1f, trm, trv;  // path ranges of the current node, the next one, and the one after (fetched early)
  float trw;
  int kwc, kwn;
  uint32_t mwn;
};
template <bool LIST>
__device__ __forceinline__ void y_range(const YArgs& a, int64_t cnt, int64_t i, int32_t& p0, int32_t& p1) {
  p0 = p1 = 0;
  if (i < cnt) {
    const int64_t k = blockIdx.x + i * int64_t(gridDim.x);
    const int64_t n = a.n0 + (LIST ? int64_t(a.list[k]) : k);
    p0 = a.pptr[n]; p1 = a.pptr[n + 1];
  }
}
template <bool LIST>
__device__ __forceinline__ void y_pipe_prologue(const YArgs& a, YWin (&win)[2], YMeta (&meta)[4], const YRole& ro, int64_t cnt,
                                                YPipe& pp) {
  const int tid = threadIdx.x;
  y_range<LIST>(a, cnt, 0, pp.p0c, pp.p1c);
  pp.kwc = min(kWin, pp.p1c - pp.p0c);
  if (tid < pp.kwc) { meta[0].m[tid] = a.pm[pp.p0c + tid]; meta[0].v[tid] = a.pv[pp.p0c + tid]; meta[0].w[tid] = a.pw[pp.p0c + tid]; }
  __syncthreads();
  stage_dma(a, win[0], meta[0], pp.kwc, ro.wave, 8, ro.lane);
  if (tid < 8 * kWin) win[0].mask[tid >> 3][tid & 7] = load_mask_word(a, meta[0], pp.kwc, tid);
  y_range<LIST>(a, cnt, 1, pp.p0n, pp.p1n);
  pp.kwn = min(kWin, pp.p1n - pp.p0n);
  pp.trm = 0; pp.trv = 0; pp.trw = 0.f;
  if (tid < pp.kwn) { pp.trm = a.pm[pp.p0n + tid]; pp.trv = a.pv[pp.p0n + tid]; pp.trw = a.pw[pp.p0n + tid]; }
  y_range<LIST>(a, cnt, 2, pp.p0f, pp.p1f);
}

// The fallback when the batch's path list does not fit its buffer (very large batches on hub-heavy graphs): a grid-stride loop
// over nodes, the paths enumerated here -- block scan over the neighbours' R lists, up to kPathWindow triples at a time in
// LDS, staged kWin at a time.  Same arithmetic, no overlap; its launch returns at once when the list did fit.
__global__ __launch_bounds__(512, 4) void ybuild_kernel(YArgs a) {
  __shared__ struct {
    YWin win;
    YMeta meta;
    int32_t fm[kPathWindow], fv[kPathWindow];
    float fw[kPathWindow];
    int32_t scan[8];
  } sh;
  if (int64_t(a.pptr[a.N]) <= a.cap) return;
  const YRole ro = y_role(a);
  const int tid = threadIdx.x, lane = ro.lane, wave = ro.wave, H = a.H;
  const int nthreads = blockDim.x, nwaves = ro.nwaves;
  const bool no_bg = a.no_bg != 0;
  for (int64_t n = a.n0 + blockIdx.x; n < a.n1; n += gridDim.x) {
    f32x16 t1[2], y2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { t1[ct][r] = 0.f; y2[ct][r] = 0.f; }
    const int32_t rs = a.rowptr[n], re = a.rowptr[n + 1];
    for (int32_t base = rs; base < re; base += nthreads) {
      // ---- this thread's neighbour v and the extent of its batch list R[v]
      int32_t v = 0, r0 = 0, cnt = 0;
      float pv = 0.f;
      if (base + tid < re) {
        v = a.col[base + tid];
        pv = a.val[base + tid];
        r0 = a.rptr[v];
        cnt = a.rptr[v + 1] - r0;
      }
      // ---- block-wide exclusive scan of cnt
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      __syncthreads();  // sh.scan / the triples of the previous chunk are still being read
      if (lane == 63) sh.scan[wave] = incl;
      __syncthreads();
      int woff = 0, total = 0;
      for (int w = 0; w < nwaves; ++w) {
        const int sw = sh.scan[w];
        if (w < wave) woff += sw;
        total += sw;
      }
      const int off = woff + incl - cnt;
      for (int wb = 0; wb < total; wb += kPathWindow) {
        __syncthreads();
        const int lo = max(off, wb), hi = min(off + cnt, wb + kPathWindow);
        for (int j = lo; j < hi; ++j) {
          const int k = j - off;
          sh.fm[j - wb] = a.r_m[r0 + k];
          sh.fw[j - wb] = pv * a.r_w[r0 + k];
          sh.fv[j - wb] = v;
        }
        __syncthreads();
        const int kall = min(kPathWindow, total - wb);
        for (int sb = 0; sb < kall; sb += kWin) {
          const int kw = min(kWin, kall - sb);
          if (sb > 0) __syncthreads();
          if (tid < kw) { sh.meta.m[tid] = sh.fm[sb + tid]; sh.meta.v[tid] = sh.fv[sb + tid]; sh.meta.w[tid] = sh.fw[sb + tid]; }
          __syncthreads();
          stage_dma(a, sh.win, sh.meta, kw, wave, nwaves, lane);
          if (tid < 8 * kWin) sh.win.mask[tid >> 3][tid & 7] = load_mask_word(a, sh.meta, kw, tid);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          mfma_window(sh.win, sh.meta, kw, ro.cls, ro.colv, ro.col_ok, ro.half, no_bg, t1, y2);
        }
      }
    }
    // ---- Y[n][c - c0][col] = W_1[c][col] * T1 + Y2
    // (32-bit offsets from two uniform bases; `late` ties the address arithmetic to this point of the program -- hipcc
    //  otherwise computes all 64 addresses at the top of the kernel and spills them around the products)
    int late = 0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(late));
    float* __restrict__ yn = a.Y + n * int64_t(a.R) * H;
    const float* __restrict__ w1p = a.W1 + int64_t(a.c0) * H;
    const int row0 = 32 * ro.rt + 4 * ro.half + late;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int colc = 64 * ro.cg + 32 * ct + ro.li;
      const bool cok = colc < H;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        const int o = row * H + colc;
        if (cok && row < a.R) yn[o] = w1p[o] * t1[ct][r] + y2[ct][r];
      }
    }
    __syncthreads();  // the next node restages the shared buffers
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Streaming Gram: S += Y^T Y for rows of `width` floats (129 .. 256), all eight waves on the matrix pipes.
// Tile split: the 4 groups of gram256.h (9 sub-tiles each, <= 6 of the 8 column blocks) go to the two waves that share a SIMD
// (hardware waves g and g + 4): 5 + 4 sub-tiles.
template <int W, int LO, int HI> __device__ __forceinline__ constexpr bool part_uses(int b) {
  for (int s = LO; s < HI; ++s)
    if (Tiles256<W>::si[s] == b || Tiles256<W>::sj[s] == b) return true;
  return false;
}
template <int W, int LO, int HI>
__device__ __forceinline__ void part_load(const float* __restrict__ p, float (&x)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) x[b] = part_uses<W, LO, HI>(b) ? p[b * 32] : 0.f;
}
template <int W, int LO, int HI>
__device__ __forceinline__ void part_mfma(const float (&x)[8], f32x16 (&acc)[HI - LO]) {
#pragma unroll
  for (int s = LO; s < HI; ++s)
    acc[s - LO] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[Tiles256<W>::si[s]], x[Tiles256<W>::sj[s]], acc[s - LO], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The headline route, everything of a node on one CU: ONE persistent 512-thread workgroup per CU walks its nodes; per node
//   (1) the window of <= 16 paths (rows b_m, g_m, coefficient rows: 1 KiB LDS-DMA pieces) is already in LDS -- it was put in
//       flight one node earlier, its triples loaded one node earlier still -- and the three path products run on the matrix
//       pipes (wave (rt, cg): 32 classes x 64 columns, v_mfma_f32_32x32x2_f32);
//   (2) Y[n] = W_1 (.) T1 + Y2 goes to an LDS tile [rows][256] (never to HBM);
//   (3) all eight waves contract it into the register-resident upper-triangular 256 x 256 accumulators (36 sub-tiles, 5 + 4 per
//       SIMD pair), S += Y[n]^T Y[n].
// Two barriers per node, no exposed memory latency; nodes with more than 16 paths (hubs) restage further windows in place.
constexpr int kYRows = 40;  // classes per launch (LDS: 2 x 55.6 KiB windows + 40 KiB tile = 152 KiB)

struct FusedShared {
  YWin win[2];
  YMeta meta[4];
  float y[kYRows][256];
};

template <int W, int LO, int HI, bool LIST>
__device__ __forceinline__ void fused_wave(const YArgs& a, FusedShared& sh, float* __restrict__ scratch) {
  constexpr int NT = HI - LO;
  constexpr bool RT1 = LO != 0;  // hardware waves 4 .. 7: the second class tile, the second half of the SIMD pair's sub-tiles
  const YRole ro = y_role(a, true);
  const int tid = threadIdx.x, H = a.H, lane = ro.lane;
  const int rtiles = a.R > 32 ? 2 : 1, ncg = (H + 63) >> 6;
  const bool path_wave = ro.rt < rtiles && ro.cg < ncg;  // (H <= 192 or R <= 32: some waves only stage and contract)
  const int r2 = (a.R + 1) & ~1;                          // rows the Gram reads (an odd class count: one zero row)
  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  YRegs<RT1> g;
  y_load_w1<RT1>(a, ro, path_wave, g);
  // the tile is zero where nobody writes: columns >= H, the odd row out
  for (int q = tid; q < kYRows * 256; q += 512) (&sh.y[0][0])[q] = 0.f;
  const int64_t stride = gridDim.x;
  // nodes of this launch: entry blockIdx.x + i * stride of the list of nodes with paths (or of the whole range)
  const int64_t nn = LIST ? int64_t(__builtin_amdgcn_readfirstlane(*a.n_list)) : a.n1 - a.n0;
  const int64_t cnt = nn > int64_t(blockIdx.x) ? (nn - blockIdx.x + stride - 1) / stride : 0;
  YPipe pp;
  y_pipe_prologue<LIST>(a, sh.win, sh.meta, ro, cnt, pp);
#ifdef LGNN_DEV
  Ph ph;
  ph.t = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 8; ++k) ph.acc[k] = 0;
#endif
  for (int64_t i = 0; i < cnt; ++i) {
    const int b = int(i & 1), ms = int(i % 3), msn = int((i + 1) % 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // node i's window has landed; node i + 1's triples are in registers
    PH_MARK(7);  // waiting for the window's copies
    if (tid < pp.kwn) { sh.meta[msn].m[tid] = pp.trm; sh.meta[msn].v[tid] = pp.trv; sh.meta[msn].w[tid] = pp.trw; }
    lds_barrier();  // window i visible to all waves; everybody is done with node i - 1 (its Gram, the other window buffer;
                    // meta slot msn's previous tenant is three nodes back)
    PH_MARK(0);  // first barrier
    // ---- asynchronous, behind this node's work: node i + 1's window, its mask words, node i + 2's triples
    stage_dma(a, sh.win[b ^ 1], sh.meta[msn], pp.kwn, ro.wave, 8, lane);
    pp.mwn = 0;
    if (tid < 8 * kWin) pp.mwn = load_mask_word(a, sh.meta[msn], pp.kwn, tid);
    // (node i + 2's range was fetched one node ago: its triples' addresses do not wait for a pointer load -- measured with
    //  the DEV build's phase counters: 10.6 % of the kernel sat in that dependent load; node i + 3's range starts now)
    const int32_t p0nn = pp.p0f, p1nn = pp.p1f;
    y_range<LIST>(a, cnt, i + 3, pp.p0f, pp.p1f);
    const int kwnn = min(kWin, p1nn - p0nn);
    pp.trm = 0; pp.trv = 0; pp.trw = 0.f;
    if (tid < kwnn) { pp.trm = a.pm[p0nn + tid]; pp.trv = a.pv[p0nn + tid]; pp.trw = a.pw[p0nn + tid]; }
    // ---- (1) + (2): the path products of node i, Y[n] into the LDS tile.  A node without paths (GraphSAGE: neither in the
    // batch nor next to it; a short last batch) has Y[n] = 0: no products, no Gram, only its share of the pipeline
    const bool has_paths = pp.p1c > pp.p0c;  // (workgroup uniform)
    PH_MARK(1);  // staging issue, prefetches
    if (has_paths) y_node_products<RT1>(a, sh.win, sh.meta, sh.y, ro, path_wave, g, b, ms, pp.kwc, pp.p0c, pp.p1c PH_PASS);
    PH_MARK(4);  // Y tile write
    if (tid < 8 * kWin) sh.win[b ^ 1].mask[tid >> 3][tid & 7] = pp.mwn;  // (readers of that buffer passed this node's barrier)
    lds_barrier();  // raw: a __syncthreads() here would drain the copies in flight for node i + 1
    PH_MARK(5);  // second barrier
    // ---- (3) S += Y[n]^T Y[n], rows two at a time (operands of step k + 1 read before the MFMAs of step k)
    if (has_paths) {
      const float* __restrict__ base = &sh.y[0][0] + (lane >> 5) * 256 + (lane & 31);
      const int nk = r2 >> 1;
      float xa[8], xb[8];
      part_load<W, LO, HI>(base, xa);
      for (int kk = 0; kk < nk; kk += 2) {
        if (kk + 1 < nk) part_load<W, LO, HI>(base + (kk + 1) * 512, xb);
        __builtin_amdgcn_sched_barrier(0);
        part_mfma<W, LO, HI>(xa, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < nk) {
          if (kk + 2 < nk) part_load<W, LO, HI>(base + (kk + 2) * 512, xa);
          __builtin_amdgcn_sched_barrier(0);
          part_mfma<W, LO, HI>(xb, acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    PH_MARK(6);  // Gram
    pp.p0c = pp.p0n; pp.p1c = pp.p1n; pp.kwc = pp.kwn;
    pp.p0n = p0nn; pp.p1n = p1nn; pp.kwn = kwnn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LGNN_DEV
  if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_phase[ro.wave][k], ph.acc[k]);
#endif
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t D = H;
#pragma unroll
  for (int s = LO; s < HI; ++s) {
    const int64_t j = Tiles256<W>::sj[s] * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t ii = Tiles256<W>::si[s] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (ii < D && j < D) atomicAdd(&scratch[ii * D + j], acc[s - LO][r]);
    }
  }
}

// LIST: the node loop runs over a device-side list of the nodes that have a path (a separate instance: the indirection
// costs the full-batch GCN launch, where every node has paths, 0.14 ms)
template <bool LIST>
__global__ __launch_bounds__(512, 2) void paths_fused_kernel(YArgs a, float* __restrict__ scratch) {
  __shared__ FusedShared sh;  // ONE LDS object (a second one makes hipcc drain vmcnt before its reads)
  if (int64_t(a.pptr[a.N]) > a.cap) return;  // the path list overflowed its buffer: the enumerating route takes over
  const int hw = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  switch (hw) {  // hardware waves g and g + 4 share a SIMD: 5 + 4 of the group's 9 sub-tiles
    case 0: fused_wave<0, 0, 5, LIST>(a, sh, scratch); break;
    case 4: fused_wave<0, 5, 9, LIST>(a, sh, scratch); break;
    case 1: fused_wave<1, 0, 5, LIST>(a, sh, scratch); break;
    case 5: fused_wave<1, 5, 9, LIST>(a, sh, scratch); break;
    case 2: fused_wave<2, 0, 5, LIST>(a, sh, scratch); break;
    case 6: fused_wave<2, 5, 9, LIST>(a, sh, scratch); break;
    case 3: fused_wave<3, 0, 5, LIST>(a, sh, scratch); break;
    default: fused_wave<3, 5, 9, LIST>(a, sh, scratch); break;
  }
}

constexpr int kSlots = 3;
constexpr int kBlockRows = 32;

__device__ float g_stream_zeros[256];  // (zero initialised) the source of copies past a row's end / past the last row

struct GramStreamArgs {
  const float* Y;       // [rows][ld], `width` floats used per row
  int64_t rows;
  int64_t ld;
  int width;
  const float* zeros;   // >= 16 bytes of zeros: the source of lanes past the row's end and of rows past the last
  float* scratch;       // [width][width], upper sub-tiles, float atomics
  const int32_t* gate;  // optional: run only if *gate > gate_cap (the overflow route of the path kernels)
  int64_t gate_cap;
};

// the 4 LDS-DMA row copies of block `blk` that this wave issues (rows 4 hw .. 4 hw + 3 of the block) into slot `slot`
__device__ __forceinline__ void issue_block(const GramStreamArgs& a, float* tiles, int64_t blk, int slot, int hw, int lane,
                                            bool lane_ok) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * hw + q;
    const int64_t row = blk * kBlockRows + r;
    const float* src = (lane_ok && row < a.rows) ? a.Y + row * a.ld + 4 * lane : a.zeros;
    float* dst = tiles + (slot * kBlockRows + r) * 256;  // wave-uniform LDS base; lane l lands at + 4 l floats
    __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                              reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
  }
}

template <int W, int LO, int HI>
__device__ __forceinline__ void stream_wave(const GramStreamArgs& a, float* tiles, int64_t nb, int hw, int lane) {
  constexpr int NT = HI - LO;
  f32x16 acc[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  const bool lane_ok = 4 * lane < a.width;  // width % 4 == 0 (launcher)
  const int64_t stride = gridDim.x;
  // prologue: blocks 0 and 1 of this workgroup are in flight before the loop
  issue_block(a, tiles, blockIdx.x, 0, hw, lane, lane_ok);
  issue_block(a, tiles, blockIdx.x + stride, 1, hw, lane, lane_ok);
  for (int64_t i = 0; i < nb; ++i) {
    // all but this wave's 4 youngest copies (block i + 1) have landed => its rows of block i are in LDS; the barrier then
    // says so for every wave's rows, and that everybody is done reading block i - 1, whose slot block i + 2 reuses
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_block(a, tiles, blockIdx.x + (i + 2) * stride, int((i + 2) % kSlots), hw, lane, lane_ok);
    const float* __restrict__ base = tiles + int(i % kSlots) * kBlockRows * 256 + (lane >> 5) * 256 + (lane & 31);
    float xa[8], xb[8];
    part_load<W, LO, HI>(base, xa);
#pragma unroll 2
    for (int kk = 0; kk < kBlockRows / 2; kk += 2) {
      part_load<W, LO, HI>(base + (kk + 1) * 512, xb);
      __builtin_amdgcn_sched_barrier(0);
      part_mfma<W, LO, HI>(xa, acc);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 2 < kBlockRows / 2) part_load<W, LO, HI>(base + (kk + 2) * 512, xa);
      __builtin_amdgcn_sched_barrier(0);
      part_mfma<W, LO, HI>(xb, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two look-ahead blocks past the end (zeros) before the LDS dies
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t D = a.width;
#pragma unroll
  for (int s = LO; s < HI; ++s) {
    const int64_t j = Tiles256<W>::sj[s] * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = Tiles256<W>::si[s] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (i < D && j < D) atomicAdd(&a.scratch[i * D + j], acc[s - LO][r]);
    }
  }
}

__global__ __launch_bounds__(512, 2) void gram256_stream_kernel(GramStreamArgs a) {
  __shared__ float tiles[kSlots * kBlockRows * 256];  // 96 KiB: ONE LDS object (a second one makes hipcc drain vmcnt)
  if (a.gate != nullptr && int64_t(*a.gate) <= a.gate_cap) return;  // (the overflow route of the path kernels)
  const int lane = threadIdx.x & 63;
  const int hw = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int64_t nblocks = (a.rows + kBlockRows - 1) / kBlockRows;
  const int64_t nb = nblocks > int64_t(blockIdx.x) ? (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  // hardware waves g and g + 4 share a SIMD (waves are dealt round-robin to the CU's four SIMDs)
  switch (hw) {
    case 0: stream_wave<0, 0, 5>(a, tiles, nb, hw, lane); break;
    case 4: stream_wave<0, 5, 9>(a, tiles, nb, hw, lane); break;
    case 1: stream_wave<1, 0, 5>(a, tiles, nb, hw, lane); break;
    case 5: stream_wave<1, 5, 9>(a, tiles, nb, hw, lane); break;
    case 2: stream_wave<2, 0, 5>(a, tiles, nb, hw, lane); break;
    case 6: stream_wave<2, 5, 9>(a, tiles, nb, hw, lane); break;
    case 3: stream_wave<3, 0, 5>(a, tiles, nb, hw, lane); break;
    default: stream_wave<3, 5, 9>(a, tiles, nb, hw, lane); break;
  }
}

}  // namespace

namespace {

// ---- GraphSAGE: the same fused kernel over ONE-hop paths ----------------------------------------------------------------
// cat_1 = [h_1 | P h_1], out = cat_1 W_1^T + b_1 (gnn/models/layers.py:26-29), so the first-layer gradient rows of node n are
//     G_c[n] = mask_n (.) ( S_c[n] W_1s + sum_m P[m, n] S_c[m] W_1n ),   S_c[m] = V_m[:, c]^T (the sample's seed column),
// W_1 = [W_1s | W_1n].  With V = diag(alpha) - u beta^T - p gamma^T this is the path sum of the GCN route with the mask at
// the DESTINATION (every path of n names n as its mask row), the neighbour half W_1n as the kernel's W_1 operand, and the
// self terms as pseudo paths:
//   sample index 2 m + 1 : neighbour path m -> n, weight P[m, n] mult_m, coefficients (alpha, -beta, -gamma)_m, rows u_m^T W_1n, p_m^T W_1n
//   sample index 2 m     : n's own beta / gamma terms: coefficients (0, -beta, -gamma)_m, rows u_m^T W_1s, p_m^T W_1s
//   sample index 2 M + c': n's own alpha term alpha_{m,c'} W_1s[c', :] as a "beta" product with a one-hot coefficient row
//                          (+1 at class c'), table row W_1s[c', :] and path weight mult_m alpha_{m,c'}
// (u^T [W_1s | W_1n] is one GEMM whose [M][2H] output IS the [2M][H] table in that order).  paths_fused_kernel runs unchanged.
__global__ __launch_bounds__(256) void sage_path_tables_kernel(const float* __restrict__ probs, const float* __restrict__ logits,
                                                               const int64_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                                               int64_t M, int64_t N, int C, int mode, float* __restrict__ coef,
                                                               float* __restrict__ up) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M + C) return;
  if (m >= M) {  // the one-hot rows
    float* __restrict__ cm = coef + (2 * M + (m - M)) * kCoefRow;
    cm[lane] = 0.f; cm[kCoefStride + lane] = lane == int(m - M) ? 1.f : 0.f; cm[2 * kCoefStride + lane] = 0.f; cm[3 * kCoefStride + lane] = 0.f;
    return;
  }
  const int64_t n = idx[m];
  const bool own = n >= 0 && n < N && pos[n] == int32_t(m);
  float pk = 0.f, fk = 0.f;
  if (own && lane < C && mode != 2) { pk = probs[m * C + lane]; fk = logits[n * C + lane]; }
  const float mb = wsum(pk * fk);  // same summation order as seed_kernel
  const float sp = sqrtf(pk), t = fk - mb;
  float al = 0.f, be = 0.f, ga = 0.f, u = 0.f;
  if (own && lane < C) {
    if (mode == 2) al = 1.41421356237309515f;
    else if (mode == 1) { al = sp * (1.f + 0.5f * t); be = sp; ga = 0.5f * sp * t; u = pk * (1.f + t); }
    else { al = sp; be = sp; u = pk; }
  }
  float* __restrict__ cs = coef + (2 * m) * kCoefRow;      // the node's own beta / gamma terms
  float* __restrict__ cn = coef + (2 * m + 1) * kCoefRow;  // a neighbour path from sample m
  cs[lane] = 0.f; cs[kCoefStride + lane] = -be; cs[2 * kCoefStride + lane] = -ga; cs[3 * kCoefStride + lane] = 0.f;
  cn[lane] = al;  cn[kCoefStride + lane] = -be; cn[2 * kCoefStride + lane] = -ga; cn[3 * kCoefStride + lane] = 0.f;
  if (lane < C) { up[m * C + lane] = u; up[(M + m) * C + lane] = own ? pk : 0.f; }
}

// One wave per node n: its paths (see above).  Row n of P^T lists the samples' nodes v with P[v, n] != 0.
template <bool FILL>
__global__ __launch_bounds__(256) void sage_path_list_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const float* __restrict__ val, int64_t N, int64_t M, int C,
                                                             const int32_t* __restrict__ pos, const int32_t* __restrict__ mult,
                                                             const float* __restrict__ coef, int32_t* __restrict__ pcnt,
                                                             const int32_t* __restrict__ pptr, int32_t* __restrict__ pm,
                                                             int32_t* __restrict__ pv, float* __restrict__ pw) {
  const int lane = threadIdx.x & 63;
  const int64_t n = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int32_t s = rowptr[n], e = rowptr[n + 1];
  int32_t run = FILL ? pptr[n] : 0;
  for (int32_t base = s; base < e; base += 64) {
    int32_t mv = INT32_MAX;
    float w = 0.f;
    if (base + lane < e) { mv = pos[col[base + lane]]; w = val[base + lane]; }
    const bool in = mv != INT32_MAX;
    const uint64_t bal = __ballot(in);
    if constexpr (FILL) {
      if (in) {
        const int32_t o = run + __popcll(bal & ((uint64_t(1) << lane) - 1));
        pm[o] = 2 * mv + 1; pv[o] = int32_t(n); pw[o] = w * float(mult[mv]);
      }
    }
    run += __popcll(bal);
  }
  const int32_t ms = pos[n];
  if (ms != INT32_MAX) {  // n is a batch node: its own beta / gamma path and the C one-hot alpha paths
    if constexpr (FILL) {
      const float tm = float(mult[ms]);
      if (lane == 0) { pm[run] = 2 * ms; pv[run] = int32_t(n); pw[run] = tm; }
      if (lane < C) {
        pm[run + 1 + lane] = int32_t(2 * M) + lane; pv[run + 1 + lane] = int32_t(n);
        pw[run + 1 + lane] = tm * coef[(2 * int64_t(ms) + 1) * kCoefRow + lane];
      }
    }
    run += 1 + C;
  }
  if (!FILL && lane == 0) pcnt[n] = run;
}

__global__ void path_flag_kernel(const int32_t* __restrict__ pptr, int64_t n0, int64_t n, uint8_t* __restrict__ flags) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k < n) flags[k] = pptr[n0 + k + 1] > pptr[n0 + k] ? 1 : 0;
}

// S2 = sum_v (entries of row v of P) * (entries of row v of P^T): the number of 2-hop paths n <- v <- m of the whole graph
__global__ void two_hop_count_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ rpt, int64_t N,
                                     unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < N; v += stride)
    acc += (unsigned long long)(rp[v + 1] - rp[v]) * (unsigned long long)(rpt[v + 1] - rpt[v]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
}  // namespace

// The path route's cost grows with the batch's number of 2-hop paths (about 1 ns each at C = 40 on top of the per-node Gram),
// the plane route's with the graph's entries: on hub-heavy graphs (sum of squared degrees) the planes win -- arxiv sizes with
// power-law degrees: 183.7 ms per fit on paths against 107.2 ms on planes; uniform degrees: 84 against 95.  The expected
// paths per destination node, S2 / N * M / N, decides; S2 is counted once per graph (one stream synchronisation, like the long-row
// list).
int two_hop_ensure(lgnn_ctx* h, hipStream_t s) {
  if (h->two_hop >= 0) return 0;
  h->two_hop = 0;
  if (h->nnz <= 0) return 0;
  DevBuf acc;
  LGNN_CALL(acc.reserve(64));
  unsigned long long host = 0;
  int rc = 0;
  if (hipMemsetAsync(acc.p, 0, 8, s) != hipSuccess) rc = 1;
  if (!rc) {
    hipLaunchKernelGGL(two_hop_count_kernel, dim3(unsigned(std::min<int64_t>(cdiv(h->N, 256), 1024))), dim3(256), 0, s,
                       h->P.rowptr, h->PT.rowptr, h->N, acc.as<unsigned long long>());
    if (hipMemcpyAsync(&host, acc.p, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = 1;
  }
  acc.release();
  if (rc) { set_error("two-hop path count failed"); return 1; }
  h->two_hop = double(host);
  return 0;
}
bool paths_pay(const lgnn_ctx* h, int64_t M) {
  static const double limit = getenv("LGNN_PATHS_PER_NODE") ? atof(getenv("LGNN_PATHS_PER_NODE")) : 24.0;  // dev: move the switch
  const double N = double(h->N);
  return h->two_hop >= 0 && h->two_hop / N * double(M) / N <= limit;
}

bool paths_supported(int kind, int L, const int64_t* dims, int act, int64_t nnz) {
  const int64_t C = dims[L], H = L >= 2 ? dims[L - 1] : 0;
  return (kind == LGNN_KIND_GCN || kind == LGNN_KIND_SAGE) && L == 2 && act == LGNN_ACT_RELU && nnz > 0 && C <= kCoefStride &&
         H > 128 && H <= 256 && H % 4 == 0;
}

int launch_gram256_stream(const float* Y, int64_t ld, int64_t rows, int64_t width, float* scratch, hipStream_t s,
                          const int32_t* gate, int64_t gate_cap) {
  LGNN_REQUIRE(width > 128 && width <= 256 && width % 4 == 0 && ld % 4 == 0 && ld >= width, "internal: streaming Gram width");
  if (rows <= 0) return 0;
  static const float* zeros = nullptr;  // address of the device-side zero block (per process; one device per process)
  if (zeros == nullptr) {
    void* p = nullptr;
    LGNN_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_stream_zeros)));
    zeros = static_cast<const float*>(p);
  }
  GramStreamArgs g{Y, rows, ld, int(width), zeros, scratch, gate, gate_cap};
  const int64_t nblocks = cdiv(rows, kBlockRows);
  hipLaunchKernelGGL(gram256_stream_kernel, dim3(unsigned(std::min<int64_t>(nblocks, 256))), dim3(512), 0, s, g);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

#ifdef LGNN_DEV
// make DEV=1: print and clear the phase counters (called from lgnn_destroy when LGNN_PHASE_REPORT is set)
void paths_phase_report() {
  unsigned long long host[8][8];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase), sizeof(host)) != hipSuccess) return;
  static const char* names[8] = {"barrier 1", "stage issue", "products w1", "products w2+", "Y write", "barrier 2", "Gram", "window wait"};
  fprintf(stderr, "paths_fused_kernel phase cycles (s_memtime ticks, summed over workgroups and launches)\n");
  for (int k = 0; k < 8; ++k) {
    fprintf(stderr, "  %-14s", names[k]);
    for (int w = 0; w < 8; ++w) fprintf(stderr, " %12llu", host[w][k]);
    fprintf(stderr, "\n");
  }
  unsigned long long zero[8][8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero));
}
#endif

// The nodes of [nb, ne) that have a path, as a device-side list (GraphSAGE: 65 % of the nodes at the arxiv shape; a short last
// batch of a GCN: 70 %): the fused kernel's node loop, its barriers and its staging pipeline then only see those.
static int path_node_list(lgnn_ctx* h, int64_t nb, int64_t ne, hipStream_t s) {
  Workspace& ws = h->ws;
  const int64_t n = ne - nb;
  LGNN_CALL(ws.path_flags.reserve(size_t(h->N)));
  LGNN_CALL(ws.path_nodes.reserve(size_t(h->N) * 4));
  LGNN_CALL(ws.path_nnodes.reserve(64));
  hipLaunchKernelGGL(path_flag_kernel, dim3(unsigned(cdiv(n, 256))), dim3(256), 0, s, ws.path_pptr.as<int32_t>(), nb, n,
                     ws.path_flags.as<uint8_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  return compact_flags(ws.path_flags.as<uint8_t>(), n, ws.path_nodes.as<int32_t>(), ws.path_nnodes.as<int32_t>(), ws.select_tmp, s);
}

// persistent workgroups of paths_fused_kernel: one per CU (149 KB of LDS each); LGNN_FUSED_WGS (dev) leaves CUs to other streams
static int64_t fused_workgroups() {
  static const int64_t n = getenv("LGNN_FUSED_WGS") ? std::max<int64_t>(1, atoll(getenv("LGNN_FUSED_WGS"))) : 256;
  return n;
}

// B_0 scratch += sum over the class columns [cb, ce) of this batch (see the file header).  Needs batch_prologue's
// probabilities / multiplicities / positions and the cached forward (logits, mask bits).
int kfac_paths_first_layer(lgnn_ctx* h, const int64_t* idx, int64_t M, int seed_mode, int64_t cb, int64_t ce, float* scratch,
                           hipStream_t s, int64_t nb, int64_t ne) {
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1];
  if (ne < 0) ne = N;
  LGNN_REQUIRE(nb >= 0 && nb <= ne && ne <= N, "internal: node range");
  if (nb == ne) return 0;
  Workspace& ws = h->ws;
  LGNN_REQUIRE(paths_supported(h->kind, h->L, h->dims, h->act, h->nnz), "internal: path route on an unsupported model");
  // ---- per-sample tables and b_m / g_m
  LGNN_CALL(ws.path_coef.reserve(size_t(M) * kCoefRow * 4));
  LGNN_CALL(ws.path_up.reserve(size_t(2 * M) * C * 4));
  LGNN_CALL(ws.path_bg.reserve(size_t(2 * M) * H * 4));
  hipLaunchKernelGGL(path_tables_kernel, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, ws.probs.as<float>(),
                     h->fc.out.as<float>(), idx, ws.pos.as<int32_t>(), M, N, int(C), seed_mode, ws.path_coef.as<float>(),
                     ws.path_up.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  const bool no_bg = seed_mode == 2;
  if (!no_bg) {
    GemmEpilogue none;
    LGNN_CALL(launch_gemm(ws.path_up.as<float>(), C, h->W[1], H, ws.path_bg.as<float>(), H, 2 * M, C, H, none, s));
  }
  // ---- R = P^T[:, batch]
  LGNN_CALL(ws.path_cnt.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_rptr.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_rm.reserve(size_t(std::max<int64_t>(h->nnz, 1)) * 4));
  LGNN_CALL(ws.path_rw.reserve(size_t(std::max<int64_t>(h->nnz, 1)) * 4));
  LGNN_CALL(ws.path_zeros.reserve(1024));
  if (!ws.path_zeros_set) {
    LGNN_HIP_CHECK(hipMemsetAsync(ws.path_zeros.p, 0, 1024, s));
    ws.path_zeros_set = true;
  }
  LGNN_HIP_CHECK(hipMemsetAsync(ws.path_cnt.p, 0, size_t(N + 1) * 4, s));
  hipLaunchKernelGGL(path_r_kernel<false>, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, ws.pos.as<int32_t>(),
                     ws.mult.as<int32_t>(), h->P.rowptr, h->P.col, h->P.val, ws.path_cnt.as<int32_t>(),
                     static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr), static_cast<float*>(nullptr));
  LGNN_CALL(exclusive_scan_i32(ws.path_cnt.as<int32_t>(), ws.path_rptr.as<int32_t>(), N + 1, ws.select_tmp, s));
  LGNN_HIP_CHECK(hipMemsetAsync(ws.path_cnt.p, 0, size_t(N + 1) * 4, s));
  hipLaunchKernelGGL(path_r_kernel<true>, dim3(unsigned(cdiv(M, 4))), dim3(256), 0, s, idx, M, N, ws.pos.as<int32_t>(),
                     ws.mult.as<int32_t>(), h->P.rowptr, h->P.col, h->P.val, ws.path_cnt.as<int32_t>(),
                     ws.path_rptr.as<int32_t>(), ws.path_rm.as<int32_t>(), ws.path_rw.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  // ---- the paths of every destination node: count (one wave per node), scan, fill -- when they fit the buffer
  // (arxiv shape: 2.2 M paths per batch of 10 000.  LGNN_PATH_LIST_CAP, read per call: tests force the enumerating route)
  int64_t cap = std::max<int64_t>(4 * h->nnz, int64_t(1) << 22);
  if (const char* e = getenv("LGNN_PATH_LIST_CAP")) cap = std::max<int64_t>(1, std::min<int64_t>(cap, atoll(e)));
  LGNN_CALL(ws.path_pcnt.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_pptr.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_pm.reserve(size_t(cap) * 4));
  LGNN_CALL(ws.path_pv.reserve(size_t(cap) * 4));
  LGNN_CALL(ws.path_pw.reserve(size_t(cap) * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(ws.path_pcnt.as<int32_t>() + N, 0, 4, s));
  const dim3 pgrid{unsigned(cdiv(N, 4))};
  hipLaunchKernelGGL(path_list_kernel<false>, pgrid, dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N,
                     ws.path_rptr.as<int32_t>(), ws.path_rm.as<int32_t>(), ws.path_rw.as<float>(), ws.path_pcnt.as<int32_t>(),
                     static_cast<const int32_t*>(nullptr), cap, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                     static_cast<float*>(nullptr));
  LGNN_CALL(exclusive_scan_i32(ws.path_pcnt.as<int32_t>(), ws.path_pptr.as<int32_t>(), N + 1, ws.select_tmp, s));
  hipLaunchKernelGGL(path_list_kernel<true>, pgrid, dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N,
                     ws.path_rptr.as<int32_t>(), ws.path_rm.as<int32_t>(), ws.path_rw.as<float>(), ws.path_pcnt.as<int32_t>(),
                     ws.path_pptr.as<int32_t>(), cap, ws.path_pm.as<int32_t>(), ws.path_pv.as<int32_t>(), ws.path_pw.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  // ---- class chunks of <= kYRows: everything of a node on one CU (paths_fused_kernel).  Only if the path list overflowed its
  // buffer do the two launches behind it run: the enumerating Y builder (planes in HBM, under the workspace cap) and the
  // streaming Gram over them; otherwise they return at once and no plane is ever allocated.
  LGNN_REQUIRE(N < (int64_t(1) << 31), "too many nodes for one launch");
  // the list of nodes with paths pays when a good share of the nodes has none: with p expected paths per node that share is
  // about exp(-p) (a full arxiv-shaped batch: p = 13.7, every node has paths -- the list would be pure overhead, measured
  // +0.17 ms per launch; its last batch of 941 samples: p = 1.3, 28 % of the nodes without)
  const double ppn = h->two_hop >= 0 ? h->two_hop / double(N) * double(M) / double(N) : 1e9;
  const bool use_list = ppn < 2.5;
  if (use_list) LGNN_CALL(path_node_list(h, nb, ne, s));
  for (int64_t c0 = cb; c0 < ce; c0 += kYRows) {
    const int64_t R = std::min<int64_t>(kYRows, ce - c0);
    YArgs y{};
    if (use_list) { y.list = ws.path_nodes.as<int32_t>(); y.n_list = ws.path_nnodes.as<int32_t>(); }
    y.rowptr = h->PT.rowptr; y.col = h->PT.col; y.val = h->PT.val;
    y.rptr = ws.path_rptr.as<int32_t>(); y.r_m = ws.path_rm.as<int32_t>(); y.r_w = ws.path_rw.as<float>();
    y.pptr = ws.path_pptr.as<int32_t>(); y.pm = ws.path_pm.as<int32_t>(); y.pv = ws.path_pv.as<int32_t>();
    y.pw = ws.path_pw.as<float>(); y.cap = cap;
    y.coef = ws.path_coef.as<float>(); y.bg = ws.path_bg.as<float>(); y.zeros = ws.path_zeros.as<float>();
    y.mask = h->fc.mask_bits[0].as<uint32_t>(); y.mask_words = int(cdiv(H, 32));
    y.W1 = h->W[1]; y.w1_ld = int(H); y.Y = nullptr; y.N = N; y.n0 = nb; y.n1 = ne; y.M = M; y.H = int(H); y.c0 = int(c0); y.R = int(R);
    y.no_bg = no_bg ? 1 : 0;
    if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel(s) of the KFAC path (bench.py roofline)
    if (y.list) hipLaunchKernelGGL(paths_fused_kernel<true>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, scratch);
    else hipLaunchKernelGGL(paths_fused_kernel<false>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, scratch);
    LGNN_HIP_CHECK(hipGetLastError());
    if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += R; }
  }
  // the overflow route: gated on the device (the host cannot know whether a batch needs it without a synchronisation), so its
  // planes -- sized under the workspace cap -- exist from the first call on
  const int64_t per_class = N * H * 4;
  const int64_t cc_max = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ce - cb, 64),
                                                                h->ws_limit / std::max<int64_t>(per_class, 1)));
  LGNN_CALL(ws.planes_a.reserve(size_t(cc_max) * N * H * 4));
  ws.planes_a_zero_ptr = nullptr;
  for (int64_t c0 = cb; c0 < ce; c0 += cc_max) {
    const int64_t R = std::min(cc_max, ce - c0);
    YArgs y{};
    y.rowptr = h->PT.rowptr; y.col = h->PT.col; y.val = h->PT.val;
    y.rptr = ws.path_rptr.as<int32_t>(); y.r_m = ws.path_rm.as<int32_t>(); y.r_w = ws.path_rw.as<float>();
    y.pptr = ws.path_pptr.as<int32_t>(); y.cap = cap;
    y.coef = ws.path_coef.as<float>(); y.bg = ws.path_bg.as<float>(); y.zeros = ws.path_zeros.as<float>();
    y.mask = h->fc.mask_bits[0].as<uint32_t>(); y.mask_words = int(cdiv(H, 32));
    y.W1 = h->W[1]; y.w1_ld = int(H); y.Y = ws.planes_a.as<float>(); y.N = N; y.n0 = nb; y.n1 = ne; y.M = M; y.H = int(H); y.c0 = int(c0); y.R = int(R);
    y.no_bg = no_bg ? 1 : 0;
    const unsigned threads = unsigned(64 * cdiv(H, 64) * cdiv(R, 32));  // (column groups) x (32-class row tiles) waves
    hipLaunchKernelGGL(ybuild_kernel, dim3(unsigned(std::min<int64_t>(ne - nb, 1024))), dim3(threads), 0, s, y);
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(launch_gram256_stream(y.Y + nb * R * H, H, (ne - nb) * R, H, scratch, s, ws.path_pptr.as<int32_t>() + N, cap));
  }
  return 0;
}

// GraphSAGE: B_0 scratch += the class columns [cb, ce) of this batch from the one-hop paths (see sage_path_tables_kernel).
// Needs batch_prologue's probabilities / multiplicities / positions and the cached forward (logits, mask bits).
int kfac_paths_first_layer_sage(lgnn_ctx* h, const int64_t* idx, int64_t M, int seed_mode, int64_t cb, int64_t ce, float* scratch,
                                hipStream_t s, int64_t nb, int64_t ne) {
  const int64_t N = h->N, C = h->dims[2], H = h->dims[1];
  if (ne < 0) ne = N;
  LGNN_REQUIRE(nb >= 0 && nb <= ne && ne <= N, "internal: node range");
  if (nb == ne) return 0;
  Workspace& ws = h->ws;
  LGNN_REQUIRE(h->kind == LGNN_KIND_SAGE && paths_supported(h->kind, h->L, h->dims, h->act, h->nnz),
               "internal: path route on an unsupported model");
  const int64_t T = 2 * M + C;  // sample indices: 2 m (own beta / gamma), 2 m + 1 (neighbour path), 2 M + c' (one-hot alpha)
  LGNN_CALL(ws.path_coef.reserve(size_t(T) * kCoefRow * 4));
  LGNN_CALL(ws.path_up.reserve(size_t(2 * M) * C * 4));
  LGNN_CALL(ws.path_bg.reserve(size_t(2 * T) * H * 4));
  hipLaunchKernelGGL(sage_path_tables_kernel, dim3(unsigned(cdiv(M + C, 4))), dim3(256), 0, s, ws.probs.as<float>(),
                     h->fc.out.as<float>(), idx, ws.pos.as<int32_t>(), M, N, int(C), seed_mode, ws.path_coef.as<float>(),
                     ws.path_up.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  // b rows [0, T): u^T [W_1s | W_1n] as rows (2 m, 2 m + 1), then W_1s[c', :]; g rows [T, 2 T): p^T [W_1s | W_1n], then zeros
  float* bg = ws.path_bg.as<float>();
  GemmEpilogue none;
  LGNN_CALL(launch_gemm(ws.path_up.as<float>(), C, h->W[1], 2 * H, bg, 2 * H, M, C, 2 * H, none, s));
  LGNN_CALL(launch_gemm(ws.path_up.as<float>() + M * C, C, h->W[1], 2 * H, bg + T * H, 2 * H, M, C, 2 * H, none, s));
  LGNN_HIP_CHECK(hipMemcpy2DAsync(bg + 2 * M * H, size_t(H) * 4, h->W[1], size_t(2 * H) * 4, size_t(H) * 4, size_t(C),
                                  hipMemcpyDeviceToDevice, s));
  LGNN_HIP_CHECK(hipMemsetAsync(bg + (T + 2 * M) * H, 0, size_t(C) * H * 4, s));
  LGNN_CALL(ws.path_zeros.reserve(1024));
  if (!ws.path_zeros_set) {
    LGNN_HIP_CHECK(hipMemsetAsync(ws.path_zeros.p, 0, 1024, s));
    ws.path_zeros_set = true;
  }
  // ---- the paths of every node: count (one wave per node), scan, fill.  At most nnz + (C + 1) M of them: the list always fits
  const int64_t cap = std::max<int64_t>(h->nnz, 1) + (C + 1) * M + 64;
  LGNN_REQUIRE(cap < (int64_t(1) << 31), "too many paths for one launch");
  LGNN_CALL(ws.path_pcnt.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_pptr.reserve(size_t(N + 1) * 4));
  LGNN_CALL(ws.path_pm.reserve(size_t(cap) * 4));
  LGNN_CALL(ws.path_pv.reserve(size_t(cap) * 4));
  LGNN_CALL(ws.path_pw.reserve(size_t(cap) * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(ws.path_pcnt.as<int32_t>() + N, 0, 4, s));
  const dim3 pgrid{unsigned(cdiv(N, 4))};
  hipLaunchKernelGGL(sage_path_list_kernel<false>, pgrid, dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N, M, int(C),
                     ws.pos.as<int32_t>(), ws.mult.as<int32_t>(), ws.path_coef.as<float>(), ws.path_pcnt.as<int32_t>(),
                     static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                     static_cast<float*>(nullptr));
  LGNN_CALL(exclusive_scan_i32(ws.path_pcnt.as<int32_t>(), ws.path_pptr.as<int32_t>(), N + 1, ws.select_tmp, s));
  hipLaunchKernelGGL(sage_path_list_kernel<true>, pgrid, dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N, M, int(C),
                     ws.pos.as<int32_t>(), ws.mult.as<int32_t>(), ws.path_coef.as<float>(), ws.path_pcnt.as<int32_t>(),
                     ws.path_pptr.as<int32_t>(), ws.path_pm.as<int32_t>(), ws.path_pv.as<int32_t>(), ws.path_pw.as<float>());
  LGNN_HIP_CHECK(hipGetLastError());
  LGNN_REQUIRE(N < (int64_t(1) << 31), "too many nodes for one launch");
  LGNN_CALL(path_node_list(h, nb, ne, s));
  for (int64_t c0 = cb; c0 < ce; c0 += kYRows) {
    const int64_t R = std::min<int64_t>(kYRows, ce - c0);
    YArgs y{};
    y.list = ws.path_nodes.as<int32_t>(); y.n_list = ws.path_nnodes.as<int32_t>();
    y.rowptr = h->PT.rowptr; y.col = h->PT.col; y.val = h->PT.val;
    y.pptr = ws.path_pptr.as<int32_t>(); y.pm = ws.path_pm.as<int32_t>(); y.pv = ws.path_pv.as<int32_t>();
    y.pw = ws.path_pw.as<float>(); y.cap = cap;
    y.coef = ws.path_coef.as<float>(); y.bg = bg; y.zeros = ws.path_zeros.as<float>();
    y.mask = h->fc.mask_bits[0].as<uint32_t>(); y.mask_words = int(cdiv(H, 32));
    y.W1 = h->W[1] + H; y.w1_ld = int(2 * H);  // the neighbour half: the alpha term of the neighbour paths
    y.Y = nullptr; y.N = N; y.n0 = nb; y.n1 = ne; y.M = T; y.H = int(H); y.c0 = int(c0); y.R = int(R);
    y.no_bg = 0;  // (the one-hot alpha paths go through the beta product: never skipped)
    if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel of the KFAC path (bench.py roofline)
    if (y.list) hipLaunchKernelGGL(paths_fused_kernel<true>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, scratch);
    else hipLaunchKernelGGL(paths_fused_kernel<false>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, scratch);
    LGNN_HIP_CHECK(hipGetLastError());
    if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += R; }
  }
  return 0;
}

}  // namespace lgnn
