// Backward plane GEMM of the KFAC path (GCN):  U[p][n][:] = act'(h[n][:]) * (G[p][n][:] @ W)
// for the ACTIVE nodes n only (nodes whose top-layer gradient row is non-zero for this batch).
//
//   K  = width of G (number of classes at the top level, <= 64)  -> no K loop, W lives in LDS
//   Nout <= 256                                                  -> a wave owns 32 rows x all columns
// Persistent 256-thread workgroups walk (plane, 128-row tile) pairs of the compacted row list; the next
// A tile is prefetched into registers while the MFMAs (v_mfma_f32_32x32x2_f32) of the current one run.
// The activation derivative comes from a per-node bit mask for ReLU (32 B per node instead of a 1 KiB
// float row) and from the float activations otherwise.  Inactive rows are never written: the fused
// SpMM that consumes U skips them (their P^T values are zeroed, kfac.hip).
#include "device_utils.h"
#include "gram256.h"  // f32x16
#include "lgnn_internal.h"

namespace lgnn {

namespace {

constexpr int BGM = 64;  // rows per tile: 2 row groups of 32 x 2 column halves = 4 waves

// LDS: W [K2][NT*32] once, then TWO copies of (A tile [64][KP], node ids [64], mask words [64][8]).
// Per tile: issue the global loads of tile i+1 -> MFMAs on tile i -> park tile i+1 in the other LDS copy
// (the loads landed during the MFMAs, and this wait comes BEFORE the epilogue's stores, so it never has
// to drain them: vmcnt is in order) -> epilogue stores of tile i -> one barrier.
template <int NT, int VECA>
__global__ __launch_bounds__(256, 2) void backgemm_kernel(BackGemmArgs g) {
  extern __shared__ float smem[];
  const int K = int(g.K), K2 = (K + 1) & ~1, KP = K2 | 1;
  constexpr int NP = NT * 32;
  constexpr int NTW = NT >= 2 ? NT / 2 : 1;  // column tiles per wave (<= 4: 64 accumulator registers)
  float* __restrict__ Bs = smem;
  const int copy_words = BGM * KP + BGM + BGM * 8;
  float* __restrict__ copy0 = smem + K2 * NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rg = wave >> 1, ch = wave & 1;
  const bool wave_active = NT >= 2 || ch == 0;
  for (int f = tid; f < K2 * NP; f += 256) {
    const int k = f / NP, c = f - k * NP;
    Bs[f] = (k < K && c < g.Nout) ? g.W[int64_t(k) * g.ldw + c] : 0.f;
  }
  if (K2 != K)
    for (int r = tid; r < 2 * BGM; r += 256) (copy0 + (r / BGM) * copy_words)[(r % BGM) * KP + K] = 0.f;

  const int64_t na = g.rows ? int64_t(*g.na_dev) : g.N;
  const int64_t tiles_per_plane = (na + BGM - 1) / BGM;
  const int64_t ntiles = tiles_per_plane * g.planes;

  constexpr int MAXV = VECA == 4 ? 4 : 16;  // staged float4 / floats per thread (64 rows, K <= 64)
  float4 st4[VECA == 4 ? MAXV : 1];
  float st1[VECA == 1 ? MAXV : 1];
  uint2 mstage = make_uint2(~0u, ~0u);  // mask words (row tid>>2, words (tid&3)*2 ..+1) of the staged tile
  int32_t nstage = -1;                  // node id of row tid (tid < 64) of the staged tile
  const int step = 256 * VECA;
  const int q = step / K, rem = step - q * K;
  const int r_first = (tid * VECA) / K, k_first = (tid * VECA) - r_first * K;

  auto tile_coords = [&](int64_t tile, int64_t& plane, int64_t& t0) {
    plane = tile / tiles_per_plane;
    t0 = (tile - plane * tiles_per_plane) * BGM;
  };
  auto node_of = [&](int64_t t) -> int64_t { return g.rows ? int64_t(g.rows[t]) : t; };
  auto load_tile = [&](int64_t tile) {
    int64_t plane, t0;
    tile_coords(tile, plane, t0);
    const float* __restrict__ base = g.G + plane * g.N * g.K;
    int r = r_first, k = k_first;
#pragma unroll
    for (int e = 0; e < MAXV; ++e) {
      const bool ok = r < BGM && t0 + r < na;
      if constexpr (VECA == 4) {
        st4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) st4[e] = *reinterpret_cast<const float4*>(base + node_of(t0 + r) * g.K + k);
      } else {
        st1[e] = ok ? base[node_of(t0 + r) * g.K + k] : 0.f;
      }
      r += q; k += rem;
      if (k >= K) { k -= K; r += 1; }
    }
    nstage = (tid < BGM && t0 + tid < na) ? int32_t(node_of(t0 + tid)) : -1;
    mstage = make_uint2(~0u, ~0u);
    if (g.mask_bits) {
      const int mr = tid >> 2, mw = (tid & 3) * 2;
      if (t0 + mr < na) {
        const uint32_t* __restrict__ mp = g.mask_bits + node_of(t0 + mr) * g.mask_words;
        if (mw + 0 < g.mask_words) mstage.x = mp[mw + 0];
        if (mw + 1 < g.mask_words) mstage.y = mp[mw + 1];
      }
    }
  };
  auto park_tile = [&](int buf) {
    float* __restrict__ As = copy0 + buf * copy_words;
    int32_t* __restrict__ nodes = reinterpret_cast<int32_t*>(As + BGM * KP);
    uint32_t* __restrict__ mask_s = reinterpret_cast<uint32_t*>(nodes + BGM);
    int r = r_first, k = k_first;
#pragma unroll
    for (int e = 0; e < MAXV; ++e) {
      if (r < BGM) {
        float* d = As + r * KP + k;
        if constexpr (VECA == 4) { d[0] = st4[e].x; d[1] = st4[e].y; d[2] = st4[e].z; d[3] = st4[e].w; }
        else d[0] = st1[e];
      }
      r += q; k += rem;
      if (k >= K) { k -= K; r += 1; }
    }
    if (tid < BGM) nodes[tid] = nstage;
    *reinterpret_cast<uint2*>(mask_s + (tid >> 2) * 8 + (tid & 3) * 2) = mstage;
  };

  int64_t tile = blockIdx.x;
  int buf = 0;
  if (tile < ntiles) {
    load_tile(tile);
    park_tile(0);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const bool has_next = tile + gridDim.x < ntiles;
    if (has_next) load_tile(tile + gridDim.x);
    const float* __restrict__ As = copy0 + buf * copy_words;
    const int32_t* __restrict__ nodes = reinterpret_cast<const int32_t*>(As + BGM * KP);
    const uint32_t* __restrict__ mask_s = reinterpret_cast<const uint32_t*>(nodes + BGM);
    int64_t plane, t0;
    tile_coords(tile, plane, t0);
    float* __restrict__ Up = g.U + plane * g.N * g.Nout;

    f32x16 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    if (wave_active) {
      const float* __restrict__ arow = As + (rg * 32 + l31) * KP + lhi;
      const float* __restrict__ brow = Bs + lhi * NP + ch * NTW * 32 + l31;
      for (int kk = 0; kk < (g.debug == 2 ? 1 : K2 / 2); ++kk) {
        const float av = arow[2 * kk];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[2 * kk * NP + n * 32], acc[n], 0, 0, 0);
      }
    }
    // the prefetched tile goes to the other LDS copy now: its loads are older than every store below
    if (has_next) park_tile(buf ^ 1);
    if (wave_active) {
      const int colb = ch * NTW * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int32_t nd = nodes[lr];
        if (nd < 0) continue;
        float* __restrict__ urow = Up + int64_t(nd) * g.Nout;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int col = colb + n * 32;
          float v = acc[n][r];
          if (g.mask_bits) v = ((mask_s[lr * 8 + ch * NTW + n] >> l31) & 1u) ? v : 0.f;
          else if (g.hact && col < g.Nout) v *= act_deriv_from_out(g.hact[int64_t(nd) * g.hact_ld + col], g.act);
          if (col < g.Nout && g.debug != 1) urow[col] = v;
        }
      }
    }
    __syncthreads();
  }
}

template <int NT>
int backgemm_launch(const BackGemmArgs& g, size_t smem, bool vec, hipStream_t s) {
  const int64_t worst = cdiv(g.N, BGM) * g.planes;
  const unsigned grid = unsigned(std::min<int64_t>(worst, 512));
  static bool attr_set = false;  // more than 64 KiB of dynamic LDS needs an explicit opt-in
  if (!attr_set) {
    LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&backgemm_kernel<NT, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&backgemm_kernel<NT, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_set = true;
  }
  if (vec) hipLaunchKernelGGL((backgemm_kernel<NT, 4>), dim3(grid), dim3(256), smem, s, g);
  else hipLaunchKernelGGL((backgemm_kernel<NT, 1>), dim3(grid), dim3(256), smem, s, g);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

// bit j of word w of node n = (h[n][32 w + j] > 0)
__global__ void relu_mask_bits_kernel(const float* __restrict__ h, int64_t ld, int64_t N, int64_t H, int64_t words,
                                      uint32_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;  // one wave per (node, 64-col chunk)
  const int64_t chunks = (H + 63) / 64;
  if (wid >= N * chunks) return;
  const int64_t n = wid / chunks, ch = wid - n * chunks;
  const int64_t col = ch * 64 + lane;
  const bool on = col < H && h[n * ld + col] > 0.f;
  const unsigned long long m = __ballot(on);
  if (lane == 0) {
    bits[n * words + 2 * ch] = uint32_t(m);
    if (2 * ch + 1 < words) bits[n * words + 2 * ch + 1] = uint32_t(m >> 32);
  }
}

}  // namespace

bool backgemm_supported(int64_t K, int64_t Nout) {
  if (K < 1 || K > 64 || Nout < 1 || Nout > 256) return false;
  const int nt = Nout <= 32 ? 1 : (Nout <= 64 ? 2 : (Nout <= 128 ? 4 : 8));
  const int K2 = int((K + 1) & ~int64_t(1)), KP = K2 | 1;
  return (size_t(K2) * nt * 32 + 2 * (size_t(BGM) * KP + BGM + BGM * 8)) * 4 <= 78 * 1024;
}

int launch_backgemm(const BackGemmArgs& g_in, hipStream_t s) {
  BackGemmArgs g = g_in;
#ifdef LGNN_DEV  // make DEV=1: ablation switches for tools/time_kernels.py
  if (const char* dbg = getenv("LGNN_BACKGEMM_DEBUG")) g.debug = atoi(dbg);
#endif
  if (g.planes <= 0 || g.N <= 0) return 0;
  LGNN_REQUIRE(backgemm_supported(g.K, g.Nout), "backgemm shape not supported");
  const int nt = g.Nout <= 32 ? 1 : (g.Nout <= 64 ? 2 : (g.Nout <= 128 ? 4 : 8));
  const int K2 = int((g.K + 1) & ~int64_t(1)), KP = K2 | 1;
  const size_t smem = (size_t(K2) * nt * 32 + 2 * (size_t(BGM) * KP + BGM + BGM * 8)) * 4;
  const bool vec = g.K % 4 == 0 && (reinterpret_cast<uintptr_t>(g.G) & 15) == 0;
  switch (nt) {
    case 1: return backgemm_launch<1>(g, smem, vec, s);
    case 2: return backgemm_launch<2>(g, smem, vec, s);
    case 4: return backgemm_launch<4>(g, smem, vec, s);
    default: return backgemm_launch<8>(g, smem, vec, s);
  }
}

int launch_relu_mask_bits(const float* h, int64_t ld, int64_t N, int64_t H, uint32_t* bits, hipStream_t s) {
  const int64_t words = cdiv(H, 32), chunks = cdiv(H, 64);
  const int64_t waves = N * chunks;
  hipLaunchKernelGGL(relu_mask_bits_kernel, dim3(unsigned(cdiv(waves * 64, 256))), dim3(256), 0, s, h, ld, N, H, words,
                     bits);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn
