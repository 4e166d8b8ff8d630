// Fused SpMM^T -> Gram for plane widths in (128, 256]: the dominant kernel of the KFAC path.
//
//   for every class plane p and every 32-row block of nodes:
//       y[r, :] = epi( self[r, :] + sum_j val[j] * in_p[col[j], :] )      r in block   (gather, 16 B / lane)
//       S      += y^T y                                                     (fp32 MFMA, registers)
//
// Wave-specialised, one 512-thread workgroup per CU: 4 waves gather (two on each of SIMD 0/1), 4 waves run
// the MFMAs (two on each of SIMD 2/3).  The row block is double buffered in LDS
// (2 x 32 KiB), one barrier per block: while the MFMA waves contract block i-1 the gather waves build
// block i, so the kernel runs at max(gather, MFMA) per block and the gather waves keep all their
// registers for loads in flight (16 neighbour rows = 16 KiB per wave).
//
// Gather: a row's (col, val) pairs come in with ONE coalesced vector load pair (64 entries) and are
// broadcast with v_readlane into scalar registers, so each neighbour row is a scalar-base +
// lane-offset global_load_dwordx4 (1 KiB per wave instruction).  Entries whose value is zero (source
// rows known to be all zero for this batch, see kfac.hip) issue no load.  The finished block lives in
// LDS only; y goes to HBM only when a lower layer needs it.
// MFMA: gram256.h (36 upper-triangular 32x32 tiles, 9 per wave, <= 6 operand reads per k-step).
#include "device_utils.h"
#include "gram256.h"
#include "lgnn_internal.h"

namespace lgnn {

namespace {

// ablation switches (tools/sweep_fused.py) exist in `make DEV=1` builds only: the production kernels test nothing in their loops
#ifdef LGNN_DEV
#define LGNN_DBG(a) ((a).debug)
#else
#define LGNN_DBG(a) 0
#endif

constexpr int KT256 = 32;  // rows per block
constexpr int UNR = 12;    // neighbour rows in flight per lane and per pipelined row
constexpr int DEPTH = 3;   // rows whose gathers are in flight per wave
constexpr int RPWB = KT256 / 4;  // rows per gather wave and block

// First <= 64 entries of one CSR row, one per lane, then compacted so that the entries with a non-zero
// value sit in lanes 0 .. nlive-1 (mbcnt rank + one ds_permute pair per ROW): entry u of the row is then
// lane u, a compile-time lane for v_readlane, and zero entries never issue a load.
struct RowEntries {
  int32_t cj;     // before compaction: column of entry `lane`; after: BYTE offset of live entry `lane`'s source
                  // row inside the plane (out of range for dead lanes: the buffer unit then returns zeros)
  int32_t cvi;    // value bits of live entry `lane` (0 for dead lanes)
  int32_t nlive;  // number of live entries among the first 64 (uniform)
  int32_t s, e;   // row bounds (uniform)
};

__device__ __forceinline__ RowEntries load_entries(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                   int32_t s, int32_t e, int lane) {
  RowEntries r;
  r.s = s; r.e = e;
  r.cj = 0; r.cvi = 0; r.nlive = 0;
  if (lane < e - s) {
    r.cj = col[s + lane];
    r.cvi = __float_as_int(val[s + lane]);
  }
  return r;
}

__device__ __forceinline__ void compact_entries(RowEntries& r, int lane, int row_bytes) {
  const bool live = r.cvi != 0;
  const unsigned long long m = __ballot(live);
  const int nlive = __builtin_amdgcn_readfirstlane(int(__builtin_popcountll(m)));
  const int below = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0));
  const int dst = live ? below : nlive + (lane - below);  // a permutation of 0..63: live first, order kept
  const int32_t off = live ? int32_t(uint32_t(r.cj) * uint32_t(row_bytes)) : int32_t(0xfffffff0u);
  r.cj = __builtin_amdgcn_ds_permute(dst * 4, off);
  r.cvi = __builtin_amdgcn_ds_permute(dst * 4, r.cvi);
  r.nlive = nlive;
}

using srd_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// Issue the gathers of live entries u0 .. u0+UNR-1 (u0 uniform).  One buffer load per entry: the plane is
// the buffer, the lane's 16 bytes are the vector offset, the neighbour row is a scalar offset.
// Per slot: one v_readlane (byte offset -> SGPR) and one buffer load; nothing is kept in scalar registers
// between issue and use (the value is read again with v_readlane when the row is consumed), and dead
// slots need no branch or select: their offset is out of range (zeros come back) and their value is 0.
template <bool CONST_U0>
__device__ __forceinline__ void issue_gathers(const RowEntries& r, int u0, srd_t srd, int voff, float4 (&x)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int idx = CONST_U0 ? u : min(u0 + u, 63);
    int32_t soff = __builtin_amdgcn_readlane(r.cj, idx);
    if (!CONST_U0 && u0 + u > 63) soff = int32_t(0xfffffff0u);
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0);
    x[u] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
}

template <bool CONST_U0>
__device__ __forceinline__ void accumulate(float4& y, const RowEntries& r, int u0, const float4 (&x)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int idx = CONST_U0 ? u : min(u0 + u, 63);
    float v = __int_as_float(__builtin_amdgcn_readlane(r.cvi, idx));
    if (!CONST_U0 && u0 + u > 63) v = 0.f;
    y.x += v * x[u].x; y.y += v * x[u].y; y.z += v * x[u].z; y.w += v * x[u].w;
  }
}

// Remainder of a row after its first UNR live entries: more rounds on the same 64 entries, then further
// 64-entry chunks (rows with more than 64 stored entries).  Not pipelined.
__device__ __forceinline__ void gather_rest(float4& y, const RowEntries& r, const int32_t* __restrict__ col,
                                            const float* __restrict__ val, srd_t srd, int voff, int row_bytes,
                                            int lane) {
  float4 x[UNR];
  for (int u0 = UNR; u0 < r.nlive; u0 += UNR) {
    issue_gathers<false>(r, u0, srd, voff, x);
    accumulate<false>(y, r, u0, x);
  }
  for (int32_t base = r.s + 64; base < r.e; base += 64) {
    RowEntries c = load_entries(col, val, base, r.e, lane);
    compact_entries(c, lane, row_bytes);
    for (int u0 = 0; u0 < c.nlive; u0 += UNR) {
      issue_gathers<false>(c, u0, srd, voff, x);
      accumulate<false>(y, c, u0, x);
    }
  }
}

template <int W>
__device__ __forceinline__ void mfma_wave(const FusedArgs& a, const float* __restrict__ tiles, int64_t nb, int lane) {
  f32x16 acc[9];
#pragma unroll
  for (int s = 0; s < 9; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  for (int64_t i = 0; i <= nb; ++i) {
    if (i > 0 && LGNN_DBG(a) != 1) gram256_block<W, KT256 / 2>(tiles + ((i - 1) & 1) * KT256 * 256, lane, acc);
    if (LGNN_DBG(a) != 4) __syncthreads();
  }
  gram256_flush<W>(a.scratch, a.width, lane, acc);
}

// MODE 0: general (optional self / float activation rows behind wave-uniform branches; the GCN path uses neither).
// MODE 1: GraphSAGE with ReLU, nothing conditional in the row loop: the self row is a buffer load whose offset is out
//         of range for rows flagged as all zero (no memory access), the derivative comes from the row's bit mask
//         (32 B instead of 1 KiB), both fetched one row ahead.
// HUB: rows with more than 64 stored entries ("long rows": hubs of a power-law graph) are not gathered here -- one wave
//      would walk thousands of neighbours while its workgroup waits at the barrier -- but arrive finished from
//      long_rows_spmm (longrows.hip) through a.hub, fetched like MODE 1's self rows: a buffer load whose offset is out of
//      range (no memory access) for ordinary rows, issued one row ahead of its use.
// LIST (MODE 1 only): a block is 32 rows of a.row_list instead of 32 consecutive rows; rows that are not listed have an
//      all-zero result and are never visited (their share of the MFMA work disappears with them).
template <int MODE, bool HUB, bool LIST = false>
__global__ __launch_bounds__(512, 2) void spmm_gram256_kernel(FusedArgs a) {
  __shared__ float tile[2][KT256][256];
  const int tid = threadIdx.x, lane = tid & 63;
  // Roles by SIMD: the 8 waves of a workgroup are dealt round-robin to the CU's 4 SIMDs, so hardware waves
  // {0,1,4,5} (SIMD 0/1) gather and {2,3,6,7} (SIMD 2/3) run the MFMAs.  A gather wave and an fp32-MFMA wave on
  // the SAME SIMD compete for its issue slot (every gather VALU instruction then costs twice); apart they do
  // not: 4.43 -> 3.90 ms per 20-plane launch.  `wave` = role index (0-3 gather, 4-7 MFMA).
  // (LIST / GraphSAGE, where the gather is light and the kernel MFMA bound: ONE MFMA wave and one gather wave per SIMD was
  //  measured -- 57.1 -> 69.8 ms per arxiv-shaped fit: a single fp32-MFMA wave does not keep its SIMD's matrix pipe busy,
  //  two per SIMD do.)
  const int hwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = ((hwave & 2) ? 4 : 0) + (hwave & 1) + ((hwave >> 2) << 1);
  const int64_t nrows_eff = LIST ? int64_t(*a.row_count) : a.nrows;  // rows visited per plane
  const int64_t blocks_per_plane = (nrows_eff + KT256 - 1) / KT256;
  const int64_t nblocks = blocks_per_plane * a.nplanes;
  const int64_t nb = nblocks > int64_t(blockIdx.x) ? (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  // (Dealing every XCD a contiguous range of each plane's row blocks instead of this round-robin was measured and
  //  dropped: +10 % per launch on the uniform graph, +18 % on the power-law one -- DESIGN.md section 8.)

  if (wave < 4) {
    // ------------------------------------------------ gather waves
    const int c0 = lane * 4;
    const bool col_ok = c0 < a.width;  // width % 4 == 0 (launcher)
    const int voff = (col_ok ? c0 : 0) * 4;            // this lane's 16 bytes inside a neighbour row
    const int row_bytes = int(a.in_ld) * 4;
    const int plane_bytes = int(uint32_t(a.nrows * a.in_ld * 4));  // < 2^32 (launcher); offsets are unsigned 32 bit
    const int32_t* __restrict__ rowptr = a.rowptr;
    const int32_t* __restrict__ colp = a.col;
    const float* __restrict__ valp = a.val;
    // (s, e) of this wave's rows in block i, one row per lane 0..RPWB-1 ... loaded one block ahead, and the
    // entries (col, val) of those rows, loaded while the previous block's gathers are in flight
    auto block_coords = [&](int64_t i, int64_t& plane, int64_t& rb) {
      const int64_t blk = blockIdx.x + i * gridDim.x;
      plane = blk / blocks_per_plane;
      rb = (blk - plane * blocks_per_plane) * KT256;
    };
    // node of the block's row `lane & 31` (LIST: through the list; -1 past the end)
    auto node_of = [&](int64_t i) -> int32_t {
      int32_t nd = -1;
      if (i < nb) {
        int64_t plane, rb;
        block_coords(i, plane, rb);
        const int64_t r = rb + (lane & 31);
        if (r < nrows_eff) nd = LIST ? a.row_list[r] : int32_t(r);
      }
      return nd;
    };
    // row pointers of block i: consecutive rows -> one load, lane r holds rowptr[rb + r] (the end of row r is lane r + 1's);
    // LIST -> lane r holds the begin of its listed row, re (by reference) its end
    auto load_rp = [&](int64_t i, int32_t& re) -> int32_t {
      int32_t rp = 0;
      re = 0;
      if constexpr (LIST) {
        const int32_t nd = node_of(i);
        if (nd >= 0 && lane < KT256) { rp = rowptr[nd]; re = rowptr[nd + 1]; }
      } else if (i < nb) {
        int64_t plane, rb;
        block_coords(i, plane, rb);
        if (lane <= KT256 && rb + lane <= a.nrows) rp = rowptr[rb + lane];
      }
      return rp;
    };
    auto load_block_entries = [&](int64_t i, int32_t rp, int32_t re, int32_t sl, RowEntries (&ent)[RPWB]) {
      int64_t plane = 0, rb = 0;
      if (i < nb) block_coords(i, plane, rb);
#pragma unroll
      for (int it = 0; it < RPWB; ++it) {
        const int r = it * 4 + wave;
        int32_t s = 0, e = 0;
        if (i < nb && rb + r < nrows_eff) {
          s = __builtin_amdgcn_readlane(rp, r);
          e = LIST ? __builtin_amdgcn_readlane(re, r) : __builtin_amdgcn_readlane(rp, r + 1);
          if constexpr (HUB) {
            if (__builtin_amdgcn_readlane(sl, r) >= 0) e = s;  // a long row: nothing to gather, it comes from a.hub
          }
        }
        ent[it] = load_entries(colp, valp, s, e, lane);
      }
    };
    // HUB: slot of the block's 32 rows in a.hub (-1: an ordinary row), one per lane, fetched with the row pointers
    auto load_slot = [&](int64_t i) -> int32_t {
      int32_t sl = -1;
      if constexpr (HUB) {
        if (i < nb) {
          const int32_t nd = node_of(i);
          if (nd >= 0) sl = a.long_slot[nd];
        }
      }
      return sl;
    };
    // MODE 1: self-row flags of the block's 32 rows, one per lane, fetched with the row pointers
    auto load_fl = [&](int64_t i) -> int32_t {
      if constexpr (MODE != 1) return 0;
      if constexpr (LIST) {
        const int32_t nd = node_of(i);
        return nd >= 0 ? int32_t(a.self_rows[nd]) : 0;
      }
      int64_t plane = 0, rb = 0;
      block_coords(i < nb ? i : (nb > 0 ? nb - 1 : 0), plane, rb);
      const int64_t r = rb + (lane & 31);
      return a.self_rows[r < a.nrows ? r : a.nrows - 1];
    };
    RowEntries ent[RPWB], ent_next[RPWB];
    int32_t re_next = 0;
    int32_t rp_next = load_rp(0, re_next);
    int32_t fl_cur = load_fl(0), fl_next = 0;
    int32_t nd_cur = LIST ? node_of(0) : 0, nd_next = 0;  // LIST: the nodes of the block's rows, one per lane
    int32_t sl_cur = load_slot(0), sl_next = load_slot(1);
    load_block_entries(0, rp_next, re_next, sl_cur, ent);
    rp_next = load_rp(1, re_next);
    for (int64_t i = 0; i <= nb; ++i) {
      if (i < nb && LGNN_DBG(a) != 2 && LGNN_DBG(a) != 4) {
        int64_t plane, rb;
        block_coords(i, plane, rb);
        const float* in = a.in + plane * a.in_plane_stride;
        // the plane as a buffer resource (wave-uniform by construction: kernel arguments and blockIdx only)
        const srd_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, plane_bytes, 0x00020000);
        float* __restrict__ t = &tile[i & 1][0][0];
#pragma unroll
        for (int it = 0; it < RPWB; ++it) compact_entries(ent[it], lane, row_bytes);
        // MODE 1: self row (zeros without a memory access when the row's flag is 0 or the row does not exist) and
        // mask word of row `itn` of this wave
        float4 q_nx = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t mw_nx = 0;
        srd_t srd_self = srd;
        if constexpr (MODE == 1)
          srd_self = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.self + plane * a.self_plane_stride), 0,
                                                       int(uint32_t(a.nrows * a.self_ld * 4)), 0x00020000);
        auto fetch_self_mask = [&](int itn) {
          if constexpr (MODE == 1) {
            const int rn = itn * 4 + wave;
            const bool exists = rb + rn < nrows_eff;
            const int64_t rown = LIST ? int64_t(__builtin_amdgcn_readlane(nd_cur, rn)) : rb + rn;  // (-1 past the list's end)
            const bool have = exists && __builtin_amdgcn_readlane(fl_cur, rn) != 0;
            const int32_t soff = have ? int32_t(uint32_t(rown) * uint32_t(a.self_ld * 4)) : int32_t(0xfffffff0u);
            const u32x4 tq = __builtin_amdgcn_raw_buffer_load_b128(srd_self, voff, soff, 0);
            q_nx = make_float4(__uint_as_float(tq.x), __uint_as_float(tq.y), __uint_as_float(tq.z), __uint_as_float(tq.w));
            const int64_t rc = exists ? rown : (LIST ? 0 : a.nrows - 1);
            mw_nx = a.mask_bits[rc * a.mask_words + (c0 < int(a.width) ? (c0 >> 5) : 0)];
          }
        };
        fetch_self_mask(0);
        // HUB: the finished row of a long row (zeros without a memory access for ordinary rows), issued one row ahead
        // and BEFORE the gathers of the rows behind it, so that waiting for it never waits for those
        float4 hub_nx = make_float4(0.f, 0.f, 0.f, 0.f);
        srd_t srd_hub = srd;
        if constexpr (HUB)
          srd_hub = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.hub + plane * a.hub_plane_stride), 0,
                                                      int(uint32_t(a.n_long * a.width * 4)), 0x00020000);
        auto fetch_hub = [&](int itn) {
          if constexpr (HUB) {
            const int32_t slot = __builtin_amdgcn_readlane(sl_cur, itn * 4 + wave);
            const int32_t soff = slot >= 0 ? int32_t(uint32_t(slot) * uint32_t(a.width * 4)) : int32_t(0xfffffff0u);
            const u32x4 th = __builtin_amdgcn_raw_buffer_load_b128(srd_hub, voff, soff, 0);
            hub_nx = make_float4(__uint_as_float(th.x), __uint_as_float(th.y), __uint_as_float(th.z), __uint_as_float(th.w));
          }
        };
        fetch_hub(0);
        // gathers DEPTH rows deep: rows it+1 .. it+DEPTH-1 are in flight while row it is consumed
        float4 x[DEPTH][UNR];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
          issue_gathers<true>(ent[d], 0, srd, voff, x[d]);
#pragma unroll
        for (int it = 0; it < RPWB; ++it) {
          const float4 hub_row = hub_nx;
          if (it + 1 < RPWB) fetch_hub(it + 1);
          if (it + DEPTH - 1 < RPWB)
            issue_gathers<true>(ent[it + DEPTH - 1], 0, srd, voff, x[(it + DEPTH - 1) % DEPTH]);
          if (it + DEPTH - 1 == RPWB - 1) {  // all gathers of this block are issued: fetch the next block's rows
            load_block_entries(i + 1, rp_next, re_next, sl_next, ent_next);
            rp_next = load_rp(i + 2, re_next);
            fl_next = load_fl(i + 1);
            if constexpr (LIST) nd_next = node_of(i + 1);
          }
          float4 y = hub_row;  // zeros unless HUB and the row is a long one
          accumulate<true>(y, ent[it], 0, x[it % DEPTH]);
          if (ent[it].nlive > UNR || ent[it].e - ent[it].s > 64)
            gather_rest(y, ent[it], colp, valp, srd, voff, row_bytes, lane);
          const int r = it * 4 + wave;
          // LIST: `row` is the listed node (where stores go); a row past the list's end is made invalid through a.nrows
          const int64_t row = LIST ? (rb + r < nrows_eff ? int64_t(__builtin_amdgcn_readlane(nd_cur, r)) : a.nrows) : rb + r;
          if constexpr (MODE == 1) {
            // self row and mask word of THIS row were issued one row ahead (q_nx / mw_nx); issue the next row's now
            const float4 q = q_nx;
            const uint32_t mword = mw_nx;
            if (it + 1 < RPWB) fetch_self_mask(it + 1);
            y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
            const uint32_t bits = mword >> (c0 & 31);
            y.x = (bits & 1u) ? y.x : 0.f; y.y = (bits & 2u) ? y.y : 0.f;
            y.z = (bits & 4u) ? y.z : 0.f; y.w = (bits & 8u) ? y.w : 0.f;
            if (!(row < a.nrows && col_ok)) y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.store && row < a.nrows && col_ok)
              *reinterpret_cast<float4*>(a.store + plane * a.store_plane_stride + row * a.store_ld + c0) = y;
          } else if (row < a.nrows && col_ok) {
            if (a.self) {
              const float4 q =
                  *reinterpret_cast<const float4*>(a.self + plane * a.self_plane_stride + row * a.self_ld + c0);
              y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
            }
            if (a.hact) {
              const float4 hh = *reinterpret_cast<const float4*>(a.hact + row * a.hact_ld + c0);
              y.x *= act_deriv_from_out(hh.x, a.act); y.y *= act_deriv_from_out(hh.y, a.act);
              y.z *= act_deriv_from_out(hh.z, a.act); y.w *= act_deriv_from_out(hh.w, a.act);
            }
            if (a.store)
              *reinterpret_cast<float4*>(a.store + plane * a.store_plane_stride + row * a.store_ld + c0) = y;
          } else {
            y = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          *reinterpret_cast<float4*>(t + r * 256 + c0) = y;
        }
#pragma unroll
        for (int it = 0; it < RPWB; ++it) ent[it] = ent_next[it];
        fl_cur = fl_next;
        if constexpr (LIST) nd_cur = nd_next;
        if constexpr (HUB) { sl_cur = sl_next; sl_next = load_slot(i + 2); }
      }
      if (LGNN_DBG(a) != 4) __syncthreads();
    }
  } else {
    // ------------------------------------------------ MFMA waves
    // each role runs its own complete block loop so the 144 accumulator registers stay put
    switch (wave - 4) {
      case 0: mfma_wave<0>(a, &tile[0][0][0], nb, lane); break;
      case 1: mfma_wave<1>(a, &tile[0][0][0], nb, lane); break;
      case 2: mfma_wave<2>(a, &tile[0][0][0], nb, lane); break;
      default: mfma_wave<3>(a, &tile[0][0][0], nb, lane); break;
    }
  }
}

}  // namespace

int launch_spmm_gram256(const FusedArgs& a_in, hipStream_t s) {
  FusedArgs a = a_in;
#ifdef LGNN_DEV  // make DEV=1: ablation switches for tools/sweep_fused.py (1 no MFMA, 2 no gather, 4 no barriers)
  if (const char* dbg = getenv("LGNN_FUSED_DEBUG")) a.debug = atoi(dbg);
#endif
  LGNN_REQUIRE(a.nrows * a.in_ld * 4 < (int64_t(1) << 32) - 4096, "plane too large for 32-bit buffer offsets");
  const int64_t nblocks = cdiv(a.nrows, KT256) * a.nplanes;
  const unsigned grid = unsigned(std::min<int64_t>(nblocks, 256));  // one persistent workgroup per CU
  const bool hub = a.long_slot != nullptr && a.hub != nullptr && a.n_long > 0;
  LGNN_REQUIRE(!hub || a.n_long * a.width * 4 < (int64_t(1) << 32) - 4096, "long-row buffer too large for 32-bit offsets");
  if (a.self && a.self_rows && a.mask_bits) {
    LGNN_REQUIRE(a.nrows * a.self_ld * 4 < (int64_t(1) << 32) - 4096, "self plane too large for 32-bit buffer offsets");
    LGNN_REQUIRE(a.mask_words * 32 >= a.width, "mask words do not cover the plane width");
    if (a.row_list) {  // only the listed rows; their number is on the device: one persistent workgroup per CU
      LGNN_REQUIRE(a.row_count != nullptr && a.store == nullptr, "row list: needs its count, and no stored planes");
      if (hub) hipLaunchKernelGGL((spmm_gram256_kernel<1, true, true>), dim3(256), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((spmm_gram256_kernel<1, false, true>), dim3(256), dim3(512), 0, s, a);
    } else if (hub) hipLaunchKernelGGL((spmm_gram256_kernel<1, true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((spmm_gram256_kernel<1, false>), dim3(grid), dim3(512), 0, s, a);
  } else {
    if (hub) hipLaunchKernelGGL((spmm_gram256_kernel<0, true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((spmm_gram256_kernel<0, false>), dim3(grid), dim3(512), 0, s, a);
  }
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn
