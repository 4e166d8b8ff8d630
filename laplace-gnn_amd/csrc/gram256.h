// Register-resident 256 x 256 fp32 Gram accumulator shared by the 4 MFMA waves of a workgroup (two on each of two SIMDs:
// one fp32-MFMA wave alone does not keep a SIMD's matrix pipe busy, see fused256.hip).
//
// The 36 upper-triangular 32x32 sub-tiles of the factor are split 9 per wave so that every wave
// touches at most 6 of the 8 column blocks of a staged row block:
//   pairs P0={0,1} P1={2,3} P2={4,5} P3={6,7};  D(a) = upper sub-tiles inside pair a (3 tiles),
//   O(a,b) = the 2x2 sub-tiles between pairs a<b (4 tiles)
//   wave 0: D0 + O01 + row 0 of O02     wave 1: D1 + O12 + row 1 of O02
//   wave 2: D2 + O23 + row 0 of O03     wave 3: D3 + O13 + row 1 of O03
// Per k-step (2 rows of the tile) a wave reads its <= 6 operand values once from LDS
// (lane l: X[k0 + (l>>5)][blk*32 + (l&31)], conflict free) and issues 9 v_mfma_f32_32x32x2_f32;
// the tables are compile-time so operands and accumulators stay in registers (no per-tile branch).
#pragma once
#include <hip/hip_runtime.h>

namespace lgnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int W> struct Tiles256;
template <> struct Tiles256<0> {
  static constexpr int si[9] = {0, 0, 1, 0, 0, 1, 1, 0, 0};
  static constexpr int sj[9] = {0, 1, 1, 2, 3, 2, 3, 4, 5};
};
template <> struct Tiles256<1> {
  static constexpr int si[9] = {2, 2, 3, 2, 2, 3, 3, 1, 1};
  static constexpr int sj[9] = {2, 3, 3, 4, 5, 4, 5, 4, 5};
};
template <> struct Tiles256<2> {
  static constexpr int si[9] = {4, 4, 5, 4, 4, 5, 5, 0, 0};
  static constexpr int sj[9] = {4, 5, 5, 6, 7, 6, 7, 6, 7};
};
template <> struct Tiles256<3> {
  static constexpr int si[9] = {6, 6, 7, 2, 2, 3, 3, 1, 1};
  static constexpr int sj[9] = {6, 7, 7, 6, 7, 6, 7, 6, 7};
};

template <int W> __device__ __forceinline__ constexpr bool tiles256_uses(int b) {
  for (int s = 0; s < 9; ++s)
    if (Tiles256<W>::si[s] == b || Tiles256<W>::sj[s] == b) return true;
  return false;
}

// acc += tile^T tile for a [rows x 256] LDS tile (row stride 256 floats), rows = 2 * ksteps.
// The operands of k-step kk+1 are read from LDS before the 9 MFMAs of k-step kk are issued, so the
// LDS latency hides behind 576 cycles of matrix work instead of stalling every k-step.
template <int W>
__device__ __forceinline__ void gram256_load(const float* __restrict__ p, float (&x)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) x[b] = tiles256_uses<W>(b) ? p[b * 32] : 0.f;
}
template <int W>
__device__ __forceinline__ void gram256_mfma9(const float (&x)[8], f32x16 (&acc)[9]) {
#pragma unroll
  for (int s = 0; s < 9; ++s)
    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[Tiles256<W>::si[s]], x[Tiles256<W>::sj[s]], acc[s], 0, 0, 0);
}
template <int W, int KSTEPS>
__device__ __forceinline__ void gram256_block(const float* __restrict__ tile, int lane, f32x16 (&acc)[9]) {
  static_assert(KSTEPS % 2 == 0, "k-steps are processed in pairs");
  const float* __restrict__ base = tile + (lane >> 5) * 256 + (lane & 31);
  float xa[8], xb[8];
  gram256_load<W>(base, xa);
#pragma unroll 2
  for (int kk = 0; kk < KSTEPS; kk += 2) {
    gram256_load<W>(base + (kk + 1) * 512, xb);
    __builtin_amdgcn_sched_barrier(0);
    gram256_mfma9<W>(xa, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (kk + 2 < KSTEPS) gram256_load<W>(base + (kk + 2) * 512, xa);
    __builtin_amdgcn_sched_barrier(0);
    gram256_mfma9<W>(xb, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int W>
__device__ __forceinline__ void gram256_flush(float* __restrict__ scratch, int64_t D, int lane, const f32x16 (&acc)[9]) {
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const int64_t j = Tiles256<W>::sj[s] * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t i = Tiles256<W>::si[s] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (i < D && j < D) atomicAdd(&scratch[i * D + j], acc[s][r]);
    }
  }
}

}  // namespace lgnn
