// Fused SpMM^T -> Gram for plane widths in (128, 256]: the dominant kernel of the KFAC path.
//
//   for every class plane p and every 32-row block of nodes:
//       y[r, :] = epi( self[r, :] + sum_j val[j] * in_p[col[j], :] )      r in block   (gather, 16 B / lane)
//       S      += y^T y                                                     (fp32 MFMA, registers)
//
// Wave-specialised, one 512-thread workgroup per CU: waves 0-3 gather, waves 4-7 run the MFMAs; the
// hardware places one wave of each kind on every SIMD.  The row block is double buffered in LDS
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

constexpr int KT256 = 32;  // rows per block
constexpr int UNR = 12;    // neighbour rows in flight per lane and per pipelined row
constexpr int DEPTH = 3;   // rows whose gathers are in flight per wave
constexpr int RPWB = KT256 / 4;  // rows per gather wave and block

// First <= 64 entries of one CSR row, one entry per lane, plus the lanes whose value is non-zero.
struct RowEntries {
  int32_t cj;               // column of entry `lane`
  int32_t cvi;              // value bits of entry `lane` (0 beyond the row)
  unsigned long long live;  // ballot(value != 0): zero entries never issue a load
  int32_t s, e;             // row bounds (uniform)
};

__device__ __forceinline__ RowEntries load_entries(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                   int32_t s, int32_t e, int lane) {
  RowEntries r;
  r.s = s; r.e = e;
  r.cj = 0; r.cvi = 0; r.live = 0ull;
  if (lane < e - s) {
    r.cj = col[s + lane];
    r.cvi = __float_as_int(val[s + lane]);
  }
  return r;
}

// Issue the gathers of the first (up to) UNR live entries of `m`; consumed bits are cleared.
__device__ __forceinline__ void issue_gathers(unsigned long long& m, int32_t cj, int32_t cvi,
                                              const float* __restrict__ in, int64_t in_ld, int cl,
                                              float4 (&x)[UNR], float (&v)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    v[u] = 0.f;
    x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m != 0ull) {
      const int b = __builtin_ctzll(m);
      m &= m - 1ull;
      const int32_t j = __builtin_amdgcn_readlane(cj, b);
      v[u] = __int_as_float(__builtin_amdgcn_readlane(cvi, b));
      x[u] = *reinterpret_cast<const float4*>(in + int64_t(j) * in_ld + cl);
    }
  }
}

__device__ __forceinline__ void accumulate(float4& y, const float4 (&x)[UNR], const float (&v)[UNR]) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    y.x += v[u] * x[u].x; y.y += v[u] * x[u].y; y.z += v[u] * x[u].z; y.w += v[u] * x[u].w;
  }
}

// Remainder of a row after its first UNR live entries were consumed: more rounds on the same 64
// entries, then further 64-entry chunks (rows with more than 64 stored entries).  Not pipelined.
__device__ __forceinline__ void gather_rest(float4& y, unsigned long long m, RowEntries r,
                                            const int32_t* __restrict__ col, const float* __restrict__ val,
                                            const float* __restrict__ in, int64_t in_ld, int lane, int cl) {
  float4 x[UNR];
  float v[UNR];
  while (m != 0ull) {
    issue_gathers(m, r.cj, r.cvi, in, in_ld, cl, x, v);
    accumulate(y, x, v);
  }
  for (int32_t base = r.s + 64; base < r.e; base += 64) {
    RowEntries c = load_entries(col, val, base, r.e, lane);
    unsigned long long mm = __ballot(c.cvi != 0);
    while (mm != 0ull) {
      issue_gathers(mm, c.cj, c.cvi, in, in_ld, cl, x, v);
      accumulate(y, x, v);
    }
  }
}

__global__ __launch_bounds__(512, 2) void spmm_gram256_kernel(FusedArgs a) {
  __shared__ float tile[2][KT256][256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t blocks_per_plane = (a.nrows + KT256 - 1) / KT256;
  const int64_t nblocks = blocks_per_plane * a.nplanes;
  const int64_t nb = nblocks > int64_t(blockIdx.x) ? (nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;

  if (wave < 4) {
    // ------------------------------------------------ gather waves
    const int c0 = lane * 4;
    const bool col_ok = c0 < a.width;  // width % 4 == 0 (launcher)
    const int cl = col_ok ? c0 : 0;
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
    auto load_rp = [&](int64_t i) -> int32_t {
      int32_t rp = 0;
      if (i < nb) {
        int64_t plane, rb;
        block_coords(i, plane, rb);
        if (lane <= KT256 && rb + lane <= a.nrows) rp = rowptr[rb + lane];
      }
      return rp;
    };
    auto load_block_entries = [&](int64_t i, int32_t rp, RowEntries (&ent)[RPWB]) {
      int64_t plane = 0, rb = 0;
      if (i < nb) block_coords(i, plane, rb);
#pragma unroll
      for (int it = 0; it < RPWB; ++it) {
        const int r = it * 4 + wave;
        int32_t s = 0, e = 0;
        if (i < nb && rb + r < a.nrows) {
          s = __builtin_amdgcn_readlane(rp, r);
          e = __builtin_amdgcn_readlane(rp, r + 1);
        }
        ent[it] = load_entries(colp, valp, s, e, lane);
      }
    };
    RowEntries ent[RPWB], ent_next[RPWB];
    int32_t rp_next = load_rp(0);
    load_block_entries(0, rp_next, ent);
    rp_next = load_rp(1);
    for (int64_t i = 0; i <= nb; ++i) {
      if (i < nb) {
        int64_t plane, rb;
        block_coords(i, plane, rb);
        const float* __restrict__ in = a.in + plane * a.in_plane_stride;
        float* __restrict__ t = &tile[i & 1][0][0];
#pragma unroll
        for (int it = 0; it < RPWB; ++it) ent[it].live = __ballot(ent[it].cvi != 0);
        // gathers DEPTH rows deep: rows it+1 .. it+DEPTH-1 are in flight while row it is consumed
        float4 x[DEPTH][UNR];
        float v[DEPTH][UNR];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
          issue_gathers(ent[d].live, ent[d].cj, ent[d].cvi, in, a.in_ld, cl, x[d], v[d]);
#pragma unroll
        for (int it = 0; it < RPWB; ++it) {
          if (it + DEPTH - 1 < RPWB)
            issue_gathers(ent[it + DEPTH - 1].live, ent[it + DEPTH - 1].cj, ent[it + DEPTH - 1].cvi, in, a.in_ld, cl,
                          x[(it + DEPTH - 1) % DEPTH], v[(it + DEPTH - 1) % DEPTH]);
          if (it + DEPTH - 1 == RPWB - 1) {  // all gathers of this block are issued: fetch the next block's rows
            load_block_entries(i + 1, rp_next, ent_next);
            rp_next = load_rp(i + 2);
          }
          float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
          accumulate(y, x[it % DEPTH], v[it % DEPTH]);
          if (ent[it].live != 0ull || ent[it].e - ent[it].s > 64)
            gather_rest(y, ent[it].live, ent[it], colp, valp, in, a.in_ld, lane, cl);
          const int r = it * 4 + wave;
          const int64_t row = rb + r;
          if (row < a.nrows && col_ok) {
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
      }
      __syncthreads();
    }
  } else {
    // ------------------------------------------------ MFMA waves
    f32x16 acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    const int mw = wave - 4;
    for (int64_t i = 0; i <= nb; ++i) {
      if (i > 0) {
        const float* __restrict__ t = &tile[(i - 1) & 1][0][0];
        switch (mw) {
          case 0: gram256_block<0, KT256 / 2>(t, lane, acc); break;
          case 1: gram256_block<1, KT256 / 2>(t, lane, acc); break;
          case 2: gram256_block<2, KT256 / 2>(t, lane, acc); break;
          default: gram256_block<3, KT256 / 2>(t, lane, acc); break;
        }
      }
      __syncthreads();
    }
    switch (mw) {
      case 0: gram256_flush<0>(a.scratch, a.width, lane, acc); break;
      case 1: gram256_flush<1>(a.scratch, a.width, lane, acc); break;
      case 2: gram256_flush<2>(a.scratch, a.width, lane, acc); break;
      default: gram256_flush<3>(a.scratch, a.width, lane, acc); break;
    }
  }
}

}  // namespace

int launch_spmm_gram256(const FusedArgs& a, hipStream_t s) {
  const int64_t nblocks = cdiv(a.nrows, KT256) * a.nplanes;
  const unsigned grid = unsigned(std::min<int64_t>(nblocks, 256));  // one persistent workgroup per CU
  hipLaunchKernelGGL(spmm_gram256_kernel, dim3(grid), dim3(512), 0, s, a);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn
