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

#ifndef LGNN_PABL
#define LGNN_PABL 0
#endif

namespace lgnn {

namespace {

constexpr int kCoefStride = 64;  // classes per coefficient row (zero padded): two 32-row MFMA tiles
constexpr int kCoefRow = 256;    // floats per sample in the coefficient table: (alpha | -beta | -gamma | pad) = 1 KiB, one LDS-DMA piece
constexpr int kPathWindow = 128; // paths staged in LDS per accumulation window
// Where class c of a call's class range [cb, cb + 64) sits inside the 64 slots of one coefficient kind: the four 16-class MFMA
// tiles of a slot i side by side, so that a product-wave lane fetches its A operands of all tiles with one 16-byte load.
__device__ __forceinline__ int coef_slot(int rel) { return ((rel & 15) << 2) | (rel >> 4); }
__device__ __forceinline__ int slot_class(int slot) { return ((slot & 3) << 4) | (slot >> 2); }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per batch sample (first occurrences only; a node listed t times carries t in its R weights).
// mode: 0 upstream seeds, 1 fork exact, 2 regression (V = sqrt(2) I).  The coefficient row holds the classes [cb, cb + 64) of the
// call in slot order (coef_slot); slots of classes >= ce are zero.
__global__ __launch_bounds__(256) void path_tables_kernel(const float* __restrict__ probs, const float* __restrict__ logits,
                                                          const int64_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                                          int64_t M, int64_t N, int C, int cb, int ce, int mode,
                                                          float* __restrict__ coef, float* __restrict__ up) {
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
  // slot `lane` of each kind holds class cb + slot_class(lane): fetched from the lane that computed it.  Classes outside the
  // call's range [cb, ce) get zero coefficients: their rows of Y come out as exact zeros, the fused kernel stores them unasked
  const int c = cb + slot_class(lane);
  const bool have = c < ce;
  const int src = have ? c : 0;
  const float sal = __shfl(al, src), sbe = __shfl(be, src), sga = __shfl(ga, src);
  cm[lane] = have ? sal : 0.f;
  cm[kCoefStride + lane] = have ? -sbe : 0.f;
  cm[2 * kCoefStride + lane] = have ? -sga : 0.f;
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
  int cb;                   // first class of the coefficient table's slots (the call's class range starts there)
  int64_t n_coef;           // rows of the coefficient table
  int debug;                // LGNN_FUSED_DEBUG (timing experiments only, results are wrong with bits 0 / 1): 1 no Gram, 2 no
                            // products, 4 product waves at priority 3, 8 Gram waves at priority 3, 16 all operand loads from sample 0 / node 0
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

// Role of a wave: (rt, cg) owns the 32-class row tile rt (classes c0 + 32 rt ...) and the columns [64 cg, 64 cg + 64) of
// Y[n]: two 32 x 32 accumulator tiles for the alpha product and two for the beta / gamma products; waves w and w + 4 (the
// two row tiles of one column group) share a SIMD.
struct YRole {
  int lane, li, half, wave, nwaves, cg, rt, cls;
  int colv[2];
  bool col_ok[2];
};
__device__ __forceinline__ YRole y_role(const YArgs& a) {
  YRole r;
  const int tid = threadIdx.x;
  r.lane = tid & 63; r.li = r.lane & 31; r.half = r.lane >> 5;
  r.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  r.nwaves = blockDim.x >> 6;
  const int ncg = (a.H + 63) >> 6;
  r.cg = r.wave % ncg; r.rt = r.wave / ncg;
  // class of this lane's A-operand row (i = lane & 31), clamped into the zero-padded coefficient row; rows past the class
  // range are computed on whatever sits there and never stored
  r.cls = coef_slot(min(a.c0 - a.cb + 32 * r.rt + r.li, kCoefStride - 1));
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    r.colv[ct] = 64 * r.cg + 32 * ct + r.li;
    r.col_ok[ct] = r.colv[ct] < a.H;
    if (!r.col_ok[ct]) r.colv[ct] = 0;
  }
  return r;
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
// The headline route (round 4): everything of a node on one CU, ONE persistent 512-thread workgroup per CU whose eight waves
// have two ROLES.  The hardware deals a workgroup's waves round-robin to the CU's four SIMDs, so hardware waves g and g + 4
// share a SIMD -- and its one matrix pipe:
//   waves 4 .. 7, the PRODUCT waves (one per SIMD): wave 4 + cg owns the columns [64 cg, 64 cg + 64) of Y[n] for all <= 48
//       classes of the launch.  Per step of FOUR paths: three 16-byte loads of the paths' coefficient rows (A operands: lane
//       (i, k) = class slot i of path k, the four class tiles of a slot side by side in the table row), two 16-byte loads of
//       the table rows b_m, g_m (B operands: lane (i, k) = columns 64 cg + 4 i .. + 3 of path k -- the four 16-column MFMA
//       tiles of the wave interleave the columns, so one load feeds all four), one mask word; 36 v_mfma_f32_16x16x4_f32
//       (3 class tiles x 4 column tiles x (alpha, beta, gamma)).  No LDS staging, no window, no restaging of hubs: the
//       operands of step s + 1 are in flight while the MFMAs of step s issue, across node boundaries (a node's triples
//       (m, v, w) arrive with ONE coalesced load a node ahead and are handed to the lanes with ds_bpermute).
//       Y[n] = W_1 (.) T1 + Y2 goes to one of TWO LDS tiles [40][256] with 16-byte stores.
//   waves 0 .. 3, the GRAM waves: S += Y[n - 1]^T Y[n - 1] from the other tile into register-resident upper-triangular
//       accumulators (the 36 sub-tiles of gram256.h, 9 per wave, 144 accumulator registers), nothing else.
// ONE barrier per node.  The product wave of a SIMD needs the matrix pipe for ~4 400 of a node's ~16 000 cycles and sleeps on
// memory the rest of the time; the Gram wave is a dense MFMA stream that takes every slot the product wave leaves: the two
// phases that round 3 ran back to back in every wave (7.7 ms per arxiv batch, matrix pipes 57 % busy) now overlap.
constexpr int kYRows = 48;  // classes per launch: three 16-class MFMA tiles (LDS: two 48 KiB tiles + W_1's 48 rows)

using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int kYStride = 272;  // floats per tile row: 256 + 16, so that the four rows of a Gram operand read (lanes 16 k ..
                               // 16 k + 15 read row k0 + k) fall on disjoint banks
struct alignas(16) FusedShared {
  float y[2][kYRows][kYStride];  // the node tiles (double buffered)
  float w1[kYRows][256];    // W_1's rows of the launch (zero past R / H)
  // hand-off counters (one writer each): ready[p] = nodes whose tile columns product wave p has published, done[g] = nodes
  // Gram wave g has contracted
  int ready[4], done[4];
};

// min over the four counters of a hand-off array (one 16-byte LDS read; wave uniform)
__device__ __forceinline__ int lds_min4(const int* c) {
  int4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(uint32_t(reinterpret_cast<uintptr_t>(c))) : "memory");
  return __builtin_amdgcn_readfirstlane(min(min(v.x, v.y), min(v.z, v.w)));
}
// publish a counter after this wave's earlier LDS traffic has completed
__device__ __forceinline__ void lds_publish(int* c, int value, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) *reinterpret_cast<volatile int*>(c) = value;
}

#ifdef LGNN_DEV  // make DEV=1: per-wave cycle counts of the fused kernel's phases (s_memtime), printed by paths_phase_report()
__device__ unsigned long long g_phase[8][8];
struct Ph { unsigned long long t, acc[8]; };
#define PH_DECL Ph ph; ph.t = __builtin_amdgcn_s_memtime(); for (int k_ = 0; k_ < 8; ++k_) ph.acc[k_] = 0
#define PH_MARK(k) do { const unsigned long long ph_n = __builtin_amdgcn_s_memtime(); ph.acc[k] += ph_n - ph.t; ph.t = ph_n; } while (0)
#define PH_FLUSH(wave, lane) do { if ((lane) == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_phase[wave][k_], ph.acc[k_]); } while (0)
#else
#define PH_DECL
#define PH_MARK(k)
#define PH_FLUSH(wave, lane)
#endif

// path range [p0, p1) of the i-th node of this workgroup (entry blockIdx.x + i * gridDim.x of the list of nodes with paths,
// or of the whole range).  Wave uniform.  `pptr` / `list` are kernel parameters of their own (__restrict__): read through the
// argument struct hipcc cannot prove them read-only and loads them with VECTOR loads followed by vmcnt(0) -- a drain of
// every load in flight once per node; as restrict parameters they are scalar loads.
template <bool LIST>
__device__ __forceinline__ void node_range(const int32_t* __restrict__ pptr, const int32_t* __restrict__ list, int64_t n0,
                                           int64_t cnt, int64_t i, int32_t& p0, int32_t& p1) {
  p0 = p1 = 0;
  if (i < cnt) {
    const int64_t k = blockIdx.x + i * int64_t(gridDim.x);
    const int64_t n = n0 + (LIST ? int64_t(list[k]) : k);
    p0 = pptr[n]; p1 = pptr[n + 1];
  }
}

// ---- loads of the product waves: inline assembly with hand-placed wait counts.  hipcc's own counts are exact only along one
// path; at the loop headers of this kernel it merges the paths pessimistically (measured: the wait for a step's operands also
// waited for half of the NEXT step's, i.e. one step of prefetch distance instead of two), and any load it tracks itself would
// make it wait for vmcnt(0) -- it does not see the assembly loads queued behind.  So every vector load of the role's loop is
// issued here and waited for with a counted s_waitcnt whose "+v" operands tie the loaded registers to the wait (uses cannot
// move above it).  vmcnt counts in order: waiting until at most n operations are outstanding retires everything older than
// the n youngest.
// BUFFER loads (descriptor in SGPRs + one 32-bit byte offset per lane): on this chip the fp32 MFMAs run on the SIMD's vector
// ALUs, so every VALU instruction of either wave of a SIMD is matrix-pipe time lost, not work hidden behind the MFMAs
// (measured: the product wave's MFMA time and the time of its other instructions add up, with or without the Gram wave) --
// 64-bit address arithmetic per load was a quarter of this wave's instructions.  An offset past the table's end reads zeros.
using i32x4 = int __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint64_t bytes) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  return i32x4{int(uint32_t(u)), int(uint32_t(u >> 32) & 0xffffu), int(uint32_t(bytes > 0xffffffffull ? 0xffffffffull : bytes)),
               0x00020000};
}
template <int OFF>
__device__ __forceinline__ void bload4(f32x4v& d, uint32_t voff, const i32x4& rsrc) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
__device__ __forceinline__ void bload1(uint32_t& d, uint32_t voff, const i32x4& rsrc) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(rsrc) : "memory");
}
struct PTables {  // descriptors of what the role reads
  i32x4 coef, b, g, mask, pm, pv, pw;
};

// The paths of a chunk (up to 64): lane l holds path p0 + l as the BYTE OFFSETS of its sample's rows in the coefficient table
// (oc) and the b / g tables (ob), of its middle node's mask words (om), and its weight.  Unconditional loads from a clamped
// index; lanes past the range get offsets 0 and weight 0 in meta_finish: every load formed from them is valid, every product
// with them is zero.
struct PMeta { uint32_t oc, ob, om, w; };  // (w: the bits of a float)
__device__ __forceinline__ void meta_issue(const PTables& tb, int32_t p0, int32_t p1, int lane, PMeta& t) {
  const uint32_t q = uint32_t(max(min(p0 + lane, p1 - 1), 0)) * 4u;
  bload1(t.oc, q, tb.pm);  // (sample m, node v: turned into offsets in meta_finish)
  bload1(t.om, q, tb.pv);
  bload1(t.w, q, tb.pw);
}
template <int YOUNGER>  // vector-memory operations issued after the triples' loads that may still be in flight
__device__ __forceinline__ void meta_finish(int32_t p0, int32_t p1, int lane, uint32_t row_bytes, uint32_t mask_bytes, PMeta& t) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(t.oc), "+v"(t.om), "+v"(t.w) : "n"(YOUNGER) : "memory");
  const bool in = p0 + lane < p1;
  const uint32_t m = in ? t.oc : 0u, v = in ? t.om : 0u;
  t.oc = m * uint32_t(kCoefRow * 4); t.ob = m * row_bytes; t.om = v * mask_bytes;
  t.w = in ? t.w : 0u;
}

struct PLane {        // what a product-wave lane is: class slot / column slot i, path k of a step
  int kq;
  uint32_t oc, ob, om;  // the lane's byte offsets inside a coefficient row (16 i), a table row (4 (64 cg + 4 i)) and a node's
                        // mask words; past H: the row's start resp. an offset outside the mask (reads zero: no bit set)
  int mshift;           // bit of the lane's first column inside its mask word
};

struct POps {        // the loaded operands of one step (22 registers)
  f32x4v ca[3];      // coefficient rows (alpha | -beta | -gamma), class slots (i, t = 0 .. 3)
  f32x4v b4, g4;     // rows b_m, g_m at the lane's four columns
  uint32_t mw;       // mask word of the path's middle node
  float w;           // path weight (0 past the chunk's last path)
};

// Issue the loads of step s (paths 4 s .. 4 s + 3 of the chunk `mt`): kStepLoads<NOBG> instructions, nothing conditional.
template <bool NOBG> constexpr int kStepLoads = NOBG ? 2 : 6;
template <bool NOBG>
__device__ __forceinline__ void p_load(const PTables& tb, const PMeta& mt, int s, const PLane& pl, POps& o) {
  const int src = 4 * s + pl.kq;  // the lane that holds this lane's path (s < 16)
  const uint32_t oc = uint32_t(__shfl(int(mt.oc), src)) + pl.oc;
  const uint32_t om = uint32_t(__shfl(int(mt.om), src)) + pl.om;
  o.w = __uint_as_float(uint32_t(__shfl(int(mt.w), src)));
#if LGNN_PABL == 2  // (timing experiment: no operand loads)
  asm volatile("" :: "v"(oc), "v"(om));
  return;
#endif
  bload4<0>(o.ca[0], oc, tb.coef);
  if constexpr (!NOBG) {
    const uint32_t ob = uint32_t(__shfl(int(mt.ob), src)) + pl.ob;
    bload4<kCoefStride * 4>(o.ca[1], oc, tb.coef);
    bload4<2 * kCoefStride * 4>(o.ca[2], oc, tb.coef);
    bload4<0>(o.b4, ob, tb.b);
    bload4<0>(o.g4, ob, tb.g);
  }
  bload1(o.mw, om, tb.mask);
}
// the step's loads have landed once at most YOUNGER younger vector-memory operations are outstanding
template <bool NOBG, int YOUNGER>
__device__ __forceinline__ void p_wait(POps& o) {
  if constexpr (NOBG)
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(o.ca[0]), "+v"(o.mw) : "n"(YOUNGER) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(o.ca[0]), "+v"(o.ca[1]), "+v"(o.ca[2]), "+v"(o.b4), "+v"(o.g4), "+v"(o.mw)
                 : "n"(YOUNGER) : "memory");
}

// A step's MFMA operands: lane (i, k): A[row i][k] = weighted coefficient of class 16 t + i, B[k][col i] = mask bit / masked
// table value of the lane's column ct.  HI (a second launch of a call with more than 48 classes): the launch's only class
// tile is the fourth of the slot.  26 vector instructions (each costs the SIMD's matrix pipe its issue cycles, see above):
// bit ct of the mask word as 0 / -1 with one v_bfe_i32, ANDed with 1.0f.
struct PCur { float a0[3], a1[3], a2[3], mf[4], bb[4], gg[4]; };
template <bool NOBG, bool HI>
__device__ __forceinline__ void p_xform(const POps& o, const PLane& pl, PCur& c) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tt = HI ? 3 : t;
    c.a0[t] = (HI && t > 0) ? 0.f : o.w * o.ca[0][tt];
    if constexpr (!NOBG) {
      c.a1[t] = (HI && t > 0) ? 0.f : o.w * o.ca[1][tt];
      c.a2[t] = (HI && t > 0) ? 0.f : o.w * o.ca[2][tt];
    }
  }
  const int bits = int(o.mw >> pl.mshift);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int on = __builtin_amdgcn_sbfe(bits, ct, 1);  // 0 or -1
    c.mf[ct] = __int_as_float(on & 0x3f800000);
    if constexpr (!NOBG) { c.bb[ct] = c.mf[ct] * o.b4[ct]; c.gg[ct] = c.mf[ct] * o.g4[ct]; }
  }
}
// an empty statement that reads every register of `c`: keeps the set alive (and out of the other set's registers) up to here
template <bool NOBG>
__device__ __forceinline__ void p_keep(const PCur& c) {
  asm volatile("" :: "v"(c.a0[0]), "v"(c.a0[1]), "v"(c.a0[2]), "v"(c.mf[0]), "v"(c.mf[1]), "v"(c.mf[2]), "v"(c.mf[3]));
  if constexpr (!NOBG) {
    asm volatile("" :: "v"(c.a1[0]), "v"(c.a1[1]), "v"(c.a1[2]), "v"(c.a2[0]), "v"(c.a2[1]), "v"(c.a2[2]));
    asm volatile("" :: "v"(c.bb[0]), "v"(c.bb[1]), "v"(c.bb[2]), "v"(c.bb[3]), "v"(c.gg[0]), "v"(c.gg[1]), "v"(c.gg[2]), "v"(c.gg[3]));
  }
}
// the 36 (NOBG: 12) MFMAs of one step;  D: col = i, row = 4 k + r
template <bool NOBG>
__device__ __forceinline__ void p_mfma(const PCur& c, f32x4v (&t1)[3][4], f32x4v (&y2)[3][4]) {
#if LGNN_PABL == 1  // (timing experiment: the step's MFMAs are not issued)
  p_keep<NOBG>(c);
  return;
#endif
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) t1[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a0[t], c.mf[ct], t1[t][ct], 0, 0, 0);
  if constexpr (!NOBG) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) y2[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a1[t], c.bb[ct], y2[t][ct], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) y2[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a2[t], c.gg[ct], y2[t][ct], 0, 0, 0);
  }
}

// The product wave's work as a stream of CHUNKS: at most 64 paths of one node (one register of triples); a node is one chunk
// (a hub: several), a node without paths one empty chunk.  Wave-uniform scalar state; the range of the node after the one
// being cut is fetched a node ahead.
template <bool LIST>
struct ChunkGen {
  int64_t gi;            // node being cut into chunks (index into this workgroup's nodes; nodes >= cnt are empty)
  int32_t gp, gend;      // its remaining paths
  int32_t pa0, pa1;      // the path range of node gi + 1
  const int32_t* __restrict__ pptr;
  const int32_t* __restrict__ list;
  int64_t n0;
  __device__ __forceinline__ void init(const int32_t* __restrict__ pptr_, const int32_t* __restrict__ list_, int64_t n0_, int64_t cnt) {
    pptr = pptr_; list = list_; n0 = n0_;
    gi = 0;
    node_range<LIST>(pptr, list, n0, cnt, 0, gp, gend);
    node_range<LIST>(pptr, list, n0, cnt, 1, pa0, pa1);
  }
  // the next chunk [q0, q1) and whether it is its node's last
  __device__ __forceinline__ void next(int64_t cnt, int32_t& q0, int32_t& q1, bool& last) {
    q0 = gp; q1 = min(gp + 64, gend);
    last = q1 >= gend;
    if (last) {
      ++gi;
      gp = pa0; gend = pa1;
      node_range<LIST>(pptr, list, n0, cnt, gi + 1, pa0, pa1);
    } else {
      gp = q1;
    }
  }
};

// The product wave of SIMD cg.  Steps come in PAIRS (8 paths): buffer A holds the loaded operands of the pair's first step, B
// of its second; each is refilled for the NEXT pair -- this chunk's, or the next chunk's first (usually the next node's) --
// right after its values were turned into MFMA operands, so two steps' loads (12 instructions) are in flight behind the 36
// MFMAs being issued.  The loads are unconditional and in one fixed order, the hand-counted waits rely on it; A / B are written
// nowhere else inside the loop.  Hence the chunk stream: hubs and empty nodes take the same path as everything else; a step past
// the chunk's last path multiplies zero weights (meta_finish), an empty chunk is one such pair.  The loop body is straight-line
// code on purpose: skipping the MFMAs of a pair's empty second step with a branch gave wrong tiles now and then (the MFMA ->
// VALU wait states hipcc inserts are counted along straight-line code; the tile write behind the loop read accumulators an
// MFMA inside the branch had not finished writing).
template <bool LIST, bool NOBG, bool HI>
__device__ __forceinline__ void product_role(const YArgs& a, const int32_t* __restrict__ pptr, const int32_t* __restrict__ list,
                                             FusedShared& sh, int64_t cnt, int cg) {
  const int lane = threadIdx.x & 63;
  const int H = a.H;
  const bool path_wave = 64 * cg < H;  // (H <= 192: the last product wave has no columns)
  PH_DECL;
  if (a.debug & 4) __builtin_amdgcn_s_setprio(3);
  if (!path_wave) return;  // (its ready counter was set to "everything" at the kernel's top)
  if (a.debug & 2) {       // (timing experiment: no products)
    lds_publish(&sh.ready[cg], INT32_MAX, lane);
    return;
  }
  PLane pl;
  const int li = lane & 15;
  pl.kq = lane >> 4;
  const int col = 64 * cg + 4 * li;
  const bool col_ok = col < H;  // (H % 4 == 0: the lane's four columns are in or out together)
  pl.oc = 16u * uint32_t(li);
  pl.ob = col_ok ? 4u * uint32_t(col) : 0u;
  pl.om = col_ok ? 4u * uint32_t(col >> 5) : 0x7ffffff0u;  // (past H: outside the mask, the load returns zero bits)
  pl.mshift = col & 31;
  const uint32_t row_bytes = uint32_t(H) * 4u, mask_bytes = uint32_t(a.mask_words) * 4u;
  PTables tb;
  tb.coef = make_rsrc(a.coef, uint64_t(a.n_coef) * kCoefRow * 4);
  tb.b = make_rsrc(a.bg, uint64_t(a.M) * row_bytes);
  tb.g = make_rsrc(a.bg + a.M * int64_t(H), uint64_t(a.M) * row_bytes);
  tb.mask = make_rsrc(a.mask, uint64_t(a.N) * mask_bytes);
  tb.pm = make_rsrc(a.pm, uint64_t(a.cap) * 4);
  tb.pv = make_rsrc(a.pv, uint64_t(a.cap) * 4);
  tb.pw = make_rsrc(a.pw, uint64_t(a.cap) * 4);
  constexpr int NL = kStepLoads<NOBG>;
  ChunkGen<LIST> gen;
  gen.init(pptr, list, a.n0, cnt);
  int32_t q0c, q1c, q0n, q1n;
  bool lastc, lastn;
  gen.next(cnt, q0c, q1c, lastc);
  gen.next(cnt, q0n, q1n, lastn);
  PMeta mc, mn;
  meta_issue(tb, q0c, q1c, lane, mc);
  meta_issue(tb, q0n, q1n, lane, mn);
  meta_finish<0>(q0c, q1c, lane, row_bytes, mask_bytes, mc);
  meta_finish<0>(q0n, q1n, lane, row_bytes, mask_bytes, mn);
  POps A, B;
  p_load<NOBG>(tb, mc, 0, pl, A);
  p_load<NOBG>(tb, mc, 1, pl, B);
  f32x4v t1[3][4], y2[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { t1[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; y2[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
  // Two sets of prepared MFMA operands, alternating (p_keep pins them to registers of their own): the next step is prepared
  // while the MFMAs of the previous one may still be reading theirs.
  PCur cA = {}, cB = {};
  bool node_has = false;  // the node being built has a path so far
  for (int64_t i = 0; i < cnt;) {  // node i's tile is built while the Gram waves contract node i - 1's (or i - 2's)
    // the chunk after the next: its range now, its paths a whole chunk before they are used.  In flight from here
    // (oldest first): A, B (issued by the previous chunk's last pair), these three loads
    int32_t q0f, q1f;
    bool lastf;
    gen.next(cnt, q0f, q1f, lastf);
    PMeta mf2;
    meta_issue(tb, q0f, q1f, lane, mf2);
    const int kch = q1c - q0c, np = max((kch + 7) >> 3, 1);
    node_has = node_has || kch > 0;
    for (int j = 0; j < np; ++j) {
      // the pair after this one: this chunk's, else the next chunk's first
      const bool more = j + 1 < np;
      PMeta mx;
      mx.oc = more ? mc.oc : mn.oc; mx.ob = more ? mc.ob : mn.ob; mx.om = more ? mc.om : mn.om; mx.w = more ? mc.w : mn.w;
      const int sx = more ? 2 * (j + 1) : 0;
      // A's loads: the NL youngest outstanding may be B's (the first pair of a chunk: B's and the three path loads -- there
      // the count also waits for B's first half, issued a whole pair earlier)
      p_wait<NOBG, NL>(A);
      PH_MARK(4);
      p_xform<NOBG, HI>(A, pl, cA);
      p_keep<NOBG>(cB);  // (cB's MFMAs may still be queued: cA must not be prepared into their operand registers)
      PH_MARK(0);
      p_load<NOBG>(tb, mx, sx, pl, A);
      p_mfma<NOBG>(cA, t1, y2);
      PH_MARK(1);
      p_wait<NOBG, NL>(B);  // (younger: A's refill)
      PH_MARK(4);
      p_xform<NOBG, HI>(B, pl, cB);
      p_keep<NOBG>(cA);  // (likewise)
      PH_MARK(0);
      p_load<NOBG>(tb, mx, sx + 1, pl, B);
      p_mfma<NOBG>(cB, t1, y2);  // (unconditional: see the role's header)
      PH_MARK(1);
    }
    if (lastc) {
      // tile i & 1 was last read by the Gram of node i - 2: every Gram wave must have counted i - 1 nodes
      if (i >= 2)
        while (lds_min4(sh.done) < int(i) - 1) __builtin_amdgcn_s_sleep(2);
      PH_MARK(3);
#if LGNN_PABL == 3  // (timing experiment: no tile write)
      if (false) {
#else
      if (node_has && col_ok) {
#endif
        // Y[n] = W_1 (.) T1 + Y2.  Rows past the launch's classes come out as the zeros they already are (their coefficients
        // and their rows of W_1 are zero): one branch around twelve unconditional 16-byte stores.
        float (*ytile)[kYStride] = sh.y[i & 1];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * pl.kq + r;
            const f32x4v w1 = *reinterpret_cast<const f32x4v*>(&sh.w1[row][col]);
            f32x4v o;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[ct] = w1[ct] * t1[t][ct][r] + y2[t][ct][r];
            *reinterpret_cast<f32x4v*>(&ytile[row][col]) = o;
          }
      }
      ++i;
      lds_publish(&sh.ready[cg], int(i), lane);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { t1[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; y2[t][ct] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
      node_has = false;
      PH_MARK(2);
    }
    // rotate the chunk stream (the paths issued at the top are older than the 2 NL loads of the last pair's refills)
    meta_finish<2 * NL>(q0f, q1f, lane, row_bytes, mask_bytes, mf2);
    q0c = q0n; q1c = q1n; lastc = lastn; mc = mn;
    q0n = q0f; q1n = q1f; lastn = lastf; mn = mf2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the refills past the last chunk
  PH_FLUSH(cg, lane);
}

// The Gram wave W: its 9 upper 32 x 32 sub-tiles of gram256.h, each as 2 x 2 tiles of v_mfma_f32_16x16x4_f32 (the lower tile
// of a diagonal sub-tile is never read by the symmetrising pass and is skipped: 34 MFMAs per four tile rows).  The SAME
// instruction shape as the product wave's on purpose: the two waves of a SIMD take turns on its matrix pipe instruction by
// instruction, so with 64-cycle 32x32x2 instructions here every one of the product wave's 32-cycle instructions waited 64
// cycles and that wave -- a third of the pipe's time for a third of the work plus its serial sections -- was the critical path
// (measured: its node time = time alone + 155 x 64 cycles; Gram waves idle 30 %).
// Operand of a k step (4 tile rows) for the 16 columns 32 b + 16 h: lane l holds Y[k0 + (l >> 4)][32 b + 16 h + (l & 15)] -- as
// A operand (row l & 15, k = l >> 4) and as B operand (k = l >> 4, column l & 15) alike.
template <int W>
__device__ __forceinline__ void gram16_load(const float* __restrict__ p, float (&x)[8][2]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    x[b][0] = tiles256_uses<W>(b) ? p[b * 32] : 0.f;
    x[b][1] = tiles256_uses<W>(b) ? p[b * 32 + 16] : 0.f;
  }
}
template <int W>
__device__ __forceinline__ void gram16_mfma(const float (&x)[8][2], f32x4v (&acc)[9][2][2]) {
#pragma unroll
  for (int s = 0; s < 9; ++s)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int hj = 0; hj < 2; ++hj) {
        if (Tiles256<W>::si[s] == Tiles256<W>::sj[s] && hi > hj) continue;  // (below the diagonal)
        acc[s][hi][hj] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[Tiles256<W>::si[s]][hi], x[Tiles256<W>::sj[s]][hj],
                                                              acc[s][hi][hj], 0, 0, 0);
      }
}
template <int W, bool LIST>
__device__ __forceinline__ void gram_role(const YArgs& a, const int32_t* __restrict__ pptr, const int32_t* __restrict__ list,
                                          FusedShared& sh, int64_t cnt, float* __restrict__ scratch) {
  const int lane = threadIdx.x & 63;
  f32x4v acc[9][2][2];
#pragma unroll
  for (int s = 0; s < 9; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[s][q >> 1][q & 1] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int nk = (a.R + 3) >> 2;  // tile rows four at a time (rows past R are zero)
  int32_t p0, p1;
  node_range<LIST>(pptr, list, a.n0, cnt, 0, p0, p1);
  PH_DECL;
  if (a.debug & 8) __builtin_amdgcn_s_setprio(3);
  const bool gwork = !(a.debug & 1);
  for (int64_t i = 0; i < cnt; ++i) {
    int32_t q0, q1;
    node_range<LIST>(pptr, list, a.n0, cnt, i + 1, q0, q1);
    // node i's tile: every product wave must have published i + 1 nodes
    while (lds_min4(sh.ready) < int(i) + 1) __builtin_amdgcn_s_sleep(2);
    PH_MARK(6);
    if (p1 > p0 && gwork) {
      const float* __restrict__ base = &sh.y[i & 1][0][0] + (lane >> 4) * kYStride + (lane & 15);
      float xa[8][2], xb[8][2];
      gram16_load<W>(base, xa);
      for (int kk = 0; kk < nk; kk += 2) {
        if (kk + 1 < nk) gram16_load<W>(base + (kk + 1) * 4 * kYStride, xb);
        __builtin_amdgcn_sched_barrier(0);
        gram16_mfma<W>(xa, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < nk) {
          if (kk + 2 < nk) gram16_load<W>(base + (kk + 2) * 4 * kYStride, xa);
          __builtin_amdgcn_sched_barrier(0);
          gram16_mfma<W>(xb, acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    lds_publish(&sh.done[W], int(i) + 1, lane);  // (the tile's reads have returned: the MFMAs above consumed them)
    PH_MARK(5);
    p0 = q0; p1 = q1;
  }
  PH_FLUSH(4 + W, lane);
  // accumulator layout of 16x16x4: column l & 15, rows 4 (l >> 4) + r
  const int64_t D = a.H;
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int s = 0; s < 9; ++s)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int hj = 0; hj < 2; ++hj) {
        if (Tiles256<W>::si[s] == Tiles256<W>::sj[s] && hi > hj) continue;
        const int64_t jj = Tiles256<W>::sj[s] * 32 + 16 * hj + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t ii = Tiles256<W>::si[s] * 32 + 16 * hi + 4 * lq + r;
          if (ii < D && jj < D) atomicAdd(&scratch[ii * D + jj], acc[s][hi][hj][r]);
        }
      }
}

// LIST: the node loop runs over a device-side list of the nodes that have a path
template <bool LIST>
__global__ __launch_bounds__(512, 2) void paths_fused_kernel(YArgs a, const int32_t* __restrict__ pptr,
                                                             const int32_t* __restrict__ list, float* __restrict__ scratch) {
  __shared__ FusedShared sh;  // ONE LDS object
  if (int64_t(pptr[a.N]) > a.cap) return;  // the path list overflowed its buffer: the enumerating route takes over
  // the tiles are zero where nobody writes: columns >= H, the odd row out
  for (int q = threadIdx.x; q < 2 * kYRows * kYStride; q += 512) (&sh.y[0][0][0])[q] = 0.f;
  for (int q = threadIdx.x; q < kYRows * 256; q += 512) {
    const int row = q >> 8, colq = q & 255;
    sh.w1[row][colq] = (row < a.R && colq < a.H) ? a.W1[int64_t(a.c0 + row) * a.w1_ld + colq] : 0.f;
  }
  if (threadIdx.x < 4) {
    sh.ready[threadIdx.x] = 64 * int(threadIdx.x) < a.H ? 0 : INT32_MAX;  // (H <= 192: the last product wave has no columns)
    sh.done[threadIdx.x] = 0;
  }
  __syncthreads();
  const int hw = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int64_t stride = gridDim.x;
  const int64_t nn = LIST ? int64_t(__builtin_amdgcn_readfirstlane(*a.n_list)) : a.n1 - a.n0;
  const int64_t cnt = nn > int64_t(blockIdx.x) ? (nn - blockIdx.x + stride - 1) / stride : 0;
  // (which half is which matters: see the kernel's header -- LGNN_FUSED_DEBUG bit 5 swaps them for the A/B run)
  const int role = (a.debug & 32) ? (hw ^ 4) : hw;
  switch (role) {
    case 4: gram_role<0, LIST>(a, pptr, list, sh, cnt, scratch); break;
    case 5: gram_role<1, LIST>(a, pptr, list, sh, cnt, scratch); break;
    case 6: gram_role<2, LIST>(a, pptr, list, sh, cnt, scratch); break;
    case 7: gram_role<3, LIST>(a, pptr, list, sh, cnt, scratch); break;
    default:
      if (a.c0 != a.cb) {  // classes cb + 48 ..: the fourth tile of the coefficient slots
        if (a.no_bg) product_role<LIST, true, true>(a, pptr, list, sh, cnt, role);
        else product_role<LIST, false, true>(a, pptr, list, sh, cnt, role);
      } else {
        if (a.no_bg) product_role<LIST, true, false>(a, pptr, list, sh, cnt, role);
        else product_role<LIST, false, false>(a, pptr, list, sh, cnt, role);
      }
      break;
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
                                                               int64_t M, int64_t N, int C, int cb, int ce, int mode,
                                                               float* __restrict__ coef, float* __restrict__ up,
                                                               float* __restrict__ alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M + C) return;
  const int c = cb + slot_class(lane);  // the class of slot `lane` (coef_slot order, see path_tables_kernel)
  const bool have = c < ce;             // (classes outside the call's range: zero coefficients)
  if (m >= M) {  // the one-hot rows
    float* __restrict__ cm = coef + (2 * M + (m - M)) * kCoefRow;
    cm[lane] = 0.f; cm[kCoefStride + lane] = (have && c == int(m - M)) ? 1.f : 0.f; cm[2 * kCoefStride + lane] = 0.f; cm[3 * kCoefStride + lane] = 0.f;
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
  const int src = have ? c : 0;
  float sal = __shfl(al, src), sbe = __shfl(be, src), sga = __shfl(ga, src);  // (unconditional: every lane takes part)
  if (!have) { sal = 0.f; sbe = 0.f; sga = 0.f; }
  float* __restrict__ cs = coef + (2 * m) * kCoefRow;      // the node's own beta / gamma terms
  float* __restrict__ cn = coef + (2 * m + 1) * kCoefRow;  // a neighbour path from sample m
  cs[lane] = 0.f; cs[kCoefStride + lane] = -sbe; cs[2 * kCoefStride + lane] = -sga; cs[3 * kCoefStride + lane] = 0.f;
  cn[lane] = sal; cn[kCoefStride + lane] = -sbe; cn[2 * kCoefStride + lane] = -sga; cn[3 * kCoefStride + lane] = 0.f;
  alpha[m * kCoefStride + lane] = al;  // class major: the weights of the one-hot alpha paths (sage_path_list_kernel)
  if (lane < C) { up[m * C + lane] = u; up[(M + m) * C + lane] = own ? pk : 0.f; }
}

// One wave per node n: its paths (see above).  Row n of P^T lists the samples' nodes v with P[v, n] != 0.
template <bool FILL>
__global__ __launch_bounds__(256) void sage_path_list_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const float* __restrict__ val, int64_t N, int64_t M, int C,
                                                             const int32_t* __restrict__ pos, const int32_t* __restrict__ mult,
                                                             const float* __restrict__ alpha, int32_t* __restrict__ pcnt,
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
        pw[run + 1 + lane] = tm * alpha[int64_t(ms) * kCoefStride + lane];
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

// max over m of the paths n <- k <- m that START at m: sum over the entries k of row m of P of the entries of row k of P.
// M times this bounds a batch's path count from the host (duplicated batch nodes share their paths).
__global__ void two_hop_max_kernel(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, int64_t N,
                                   unsigned long long* __restrict__ out) {
  unsigned long long best = 0;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t m = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; m < N; m += stride) {
    unsigned long long acc = 0;
    for (int32_t p = rp[m]; p < rp[m + 1]; ++p) { const int32_t k = col[p]; acc += (unsigned long long)(rp[k + 1] - rp[k]); }
    best = acc > best ? acc : best;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = t > best ? t : best; }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
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
  unsigned long long host[2] = {0, 0};
  int rc = 0;
  if (hipMemsetAsync(acc.p, 0, 16, s) != hipSuccess) rc = 1;
  if (!rc) {
    hipLaunchKernelGGL(two_hop_count_kernel, dim3(unsigned(std::min<int64_t>(cdiv(h->N, 256), 1024))), dim3(256), 0, s,
                       h->P.rowptr, h->PT.rowptr, h->N, acc.as<unsigned long long>());
    hipLaunchKernelGGL(two_hop_max_kernel, dim3(unsigned(std::min<int64_t>(cdiv(h->N, 256), 1024))), dim3(256), 0, s,
                       h->P.rowptr, h->P.col, h->N, acc.as<unsigned long long>() + 1);
    if (hipMemcpyAsync(host, acc.p, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = 1;
  }
  acc.release();
  if (rc) { set_error("two-hop path count failed"); return 1; }
  h->two_hop = double(host[0]);
  h->two_hop_max = double(host[1]);
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
  static const char* names[8] = {"P: operand prep", "P: loads + MFMA issue", "P: Y write + publish", "P: drain + wait for tile buffer", "P: operand wait", "G: Gram + publish", "G: wait for tile", "-"};
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

static int fused_debug() {  // timing experiments (see YArgs::debug); read per call
  const char* e = getenv("LGNN_FUSED_DEBUG");
  return e ? atoi(e) : 0;
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
                     h->fc.out.as<float>(), idx, ws.pos.as<int32_t>(), M, N, int(C), int(cb), int(ce), seed_mode,
                     ws.path_coef.as<float>(), ws.path_up.as<float>());
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
    y.cb = int(cb); y.n_coef = M; y.no_bg = no_bg ? 1 : 0; y.debug = fused_debug();
    if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel(s) of the KFAC path (bench.py roofline)
    if (y.list) hipLaunchKernelGGL(paths_fused_kernel<true>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, y.pptr, y.list, scratch);
    else hipLaunchKernelGGL(paths_fused_kernel<false>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, y.pptr, y.list, scratch);
    LGNN_HIP_CHECK(hipGetLastError());
    if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += R; }
  }
  // the overflow route: gated on the device (the host cannot know a batch's path count without a synchronisation).  What the
  // host does know is a bound: M times the largest number of paths that start at one node (counted once per graph beside the
  // graph's total).  If that fits the list, no batch can overflow: no planes, no launches (arxiv shape: 10 000 x 726 paths
  // against a cap of 10 M entries -- the 6.9 GB of planes are never reserved).
  if (h->two_hop_max >= 0 && double(M) * h->two_hop_max <= double(cap)) return 0;
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
    y.cb = int(cb); y.n_coef = M; y.no_bg = no_bg ? 1 : 0;
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
  LGNN_CALL(ws.path_alpha.reserve(size_t(M) * kCoefStride * 4));
  hipLaunchKernelGGL(sage_path_tables_kernel, dim3(unsigned(cdiv(M + C, 4))), dim3(256), 0, s, ws.probs.as<float>(),
                     h->fc.out.as<float>(), idx, ws.pos.as<int32_t>(), M, N, int(C), int(cb), int(ce), seed_mode,
                     ws.path_coef.as<float>(), ws.path_up.as<float>(), ws.path_alpha.as<float>());
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
                     ws.pos.as<int32_t>(), ws.mult.as<int32_t>(), ws.path_alpha.as<float>(), ws.path_pcnt.as<int32_t>(),
                     static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                     static_cast<float*>(nullptr));
  LGNN_CALL(exclusive_scan_i32(ws.path_pcnt.as<int32_t>(), ws.path_pptr.as<int32_t>(), N + 1, ws.select_tmp, s));
  hipLaunchKernelGGL(sage_path_list_kernel<true>, pgrid, dim3(256), 0, s, h->PT.rowptr, h->PT.col, h->PT.val, N, M, int(C),
                     ws.pos.as<int32_t>(), ws.mult.as<int32_t>(), ws.path_alpha.as<float>(), ws.path_pcnt.as<int32_t>(),
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
    y.cb = int(cb); y.n_coef = T; y.no_bg = 0;  // (the one-hot alpha paths go through the beta product: never skipped)
    y.debug = fused_debug();
    if (h->timing) LGNN_CALL(record_event(h, s));  // dominant kernel of the KFAC path (bench.py roofline)
    if (y.list) hipLaunchKernelGGL(paths_fused_kernel<true>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, y.pptr, y.list, scratch);
    else hipLaunchKernelGGL(paths_fused_kernel<false>, dim3(unsigned(std::min<int64_t>(ne - nb, fused_workgroups()))), dim3(512), 0, s, y, y.pptr, y.list, scratch);
    LGNN_HIP_CHECK(hipGetLastError());
    if (h->timing) { LGNN_CALL(record_event(h, s)); h->ev_planes += R; }
  }
  return 0;
}

}  // namespace lgnn
