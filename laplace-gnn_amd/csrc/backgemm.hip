// Backward plane GEMM of the KFAC path (GCN):  U[p][n][:] = act'(h[n][:]) * (G[p][n][:] @ W)
// for the ACTIVE nodes n only (nodes whose top-layer gradient row is non-zero for this batch).
//
//   K  = width of G (number of classes at the top level, <= 64)  -> no K loop
//   Nout <= 256                                                  -> a wave owns 32 rows x 32 columns
//
// No LDS tiles, no barriers: every wave is an independent stream over its own work units (plane, 32-row tile of the
// compacted row list), always for the same 32 output columns.
//  * v_mfma_f32_32x32x2_f32 sums over k in any order as long as A and B agree, so k-step kk pairs k = kk (lanes 0-31)
//    with k = H + kk (lanes 32-63), H = K2 / 2.  A lane then needs H CONSECUTIVE floats of one row of G -- the first
//    or the second half of "its" row l & 31 -- which are plain aligned 16-byte loads straight into the MFMA operand
//    registers; nothing passes through LDS and no element is selected or moved.
//  * The wave's slice of W (32 columns, the same for all its units) is H registers per lane, loaded once.
//  * Software pipeline: rows of unit u+2 and the list entry of unit u+3 are issued BEFORE the stores of unit u.  vmcnt
//    is in order: a load younger than a store cannot be waited for without draining that store.
//  * Every load is unconditional: a branch around a load makes the compiler drain vmcnt at the join.
// The activation derivative comes from a per-node bit mask for ReLU (32 B per node instead of a 1 KiB float row), kept
// with the node ids in a wave-private LDS strip.  Inactive rows are never written: the fused SpMM that consumes U skips
// them (their P^T values are zeroed, kfac.hip).
// History: 64-row tiles staged through LDS with one barrier per tile ran at 0.80 ms per arxiv-shaped launch (2 GB of
// stores, 41 GFLOP), bound by neither the stores nor the MFMAs but by the serialisation between them.
#include "device_utils.h"
#include "gram256.h"  // f32x16
#include <numeric>

#include "lgnn_internal.h"

namespace lgnn {

namespace {

constexpr int BGM = 32;      // rows per unit
constexpr int META_LD = 12;  // wave-private strips (two, alternating): [32][12] words = node id, 3 unused, 8 mask words
constexpr int BG_THREADS = 256, BG_WAVES = BG_THREADS / 64;
constexpr int BG_OCC = 3;  // waves per SIMD the register budget is set for
using u4u = __attribute__((ext_vector_type(4), aligned(4))) unsigned int;

// KH: 16-byte pieces per half row (H = 4 KH floats >= ceil(K / 2)); VEC: K == 8 KH, rows are read with 16-byte loads
template <int KH, bool VEC, bool RELU>
__global__ __launch_bounds__(BG_THREADS, BG_OCC) void backgemm_kernel(BackGemmArgs g) {
  extern __shared__ int32_t meta_all[];
  constexpr int H = 4 * KH;  // k-steps
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int K = int(g.K);
  const int NCH = int((g.Nout + 31) / 32);  // column groups of 32 = units per row tile
  int32_t* __restrict__ meta = meta_all + wave * (2 * BGM * META_LD);

  const int64_t na = int64_t(*g.na_dev);
  const uint32_t tiles_per_plane = uint32_t((na + BGM - 1) / BGM);
  const int64_t ntiles = int64_t(tiles_per_plane) * g.planes;
  // wave w of the grid always works on column group w % NCH (the launcher makes the wave count a multiple of NCH)
  const int64_t gwave = int64_t(blockIdx.x) * BG_WAVES + wave, nwaves = int64_t(gridDim.x) * BG_WAVES;
  const int ch = int(gwave % NCH);
  const int64_t first = gwave / NCH, stride = nwaves / NCH;  // tiles
  if (first >= ntiles) return;
  const int mw = int(g.mask_words);
  const int col = ch * 32 + l31;
  const bool col_ok = col < g.Nout;
  const int colx = col_ok ? col : int(col % g.Nout);  // a column that exists, for the lanes past Nout
  const int32_t rowb = int32_t(g.Nout) * 4;           // the strip keeps a row's BYTE offset inside its plane (< 2^31)
  const int32_t trash = int32_t(g.N) * rowb;           // the plane's spare row

  // W[k][col] for this lane's k-steps: k = lhi * H + kk
  float bw[H];
#pragma unroll
  for (int kk = 0; kk < H; ++kk) {
    const int k = lhi * H + kk;
    bw[kk] = (k < K && col_ok) ? g.W[int64_t(k) * g.ldw + col] : 0.f;
  }

  auto tile_coords = [&](int64_t tile, uint32_t& plane, int64_t& t0) {  // 32-bit division (the launcher bounds ntiles)
    plane = uint32_t(tile) / tiles_per_plane;
    t0 = int64_t(uint32_t(tile) - plane * tiles_per_plane) * BGM;
  };
  const auto tile_at = [&](int64_t t) { return t < ntiles ? t : ntiles - 1; };  // past the end: loaded, never used
  // NOTHING below is conditional on a memory operation: a branch around a load or a store makes the compiler drain
  // vmcnt at the join (also a wave-uniform branch, also at the loop latch), which serialises every unit's stores with
  // the next unit's loads and MFMAs.  Rows past the end of the row list re-read its last entry and are stored to the
  // plane's spare row N; lanes past Nout do the same in a column that exists.
  // (the loaded id is not touched by any ALU instruction here: using it would make the wave wait for the load, and
  //  for every older load, at the point of use)
  auto load_id = [&](int64_t tile, int32_t& ndc, bool& valid) {
    uint32_t plane;
    int64_t t0;
    tile_coords(tile, plane, t0);
    const int64_t t = t0 + l31;
    valid = t < na;
    ndc = g.rows[valid ? t : na - 1];
  };
  auto load_rows = [&](int64_t tile, int32_t ndc, float (&a)[H], u4u& m0, u4u& m1) {
    uint32_t plane;
    int64_t t0;
    tile_coords(tile, plane, t0);
    const float* __restrict__ src = g.G + (int64_t(plane) * g.N + ndc) * g.K;
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < KH; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(src + lhi * H + 4 * j);
        a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < H; ++kk) {
        const int k = lhi * H + kk;
        a[kk] = src[k < K ? k : K - 1];  // k >= K meets a zero of W
      }
    }
    if constexpr (RELU) {
      // 8 words are read whatever mask_words is (the buffer carries the slack)
      const uint32_t* __restrict__ mp = g.mask_bits + int64_t(ndc) * mw;
      m0 = *reinterpret_cast<const u4u*>(mp);
      m1 = *reinterpret_cast<const u4u*>(mp + 4);
    } else {
      // no bit mask: the strip carries the node id (for the activation rows) in the first mask slot
      m0 = u4u{uint32_t(ndc), 0u, 0u, 0u};
      m1 = u4u{0u, 0u, 0u, 0u};
    }
  };
  auto park_meta = [&](int32_t* strip, int32_t off, const u4u& m0, const u4u& m1) {
    if (lhi == 0) {
      strip[l31 * META_LD] = off;
      *reinterpret_cast<uint4*>(strip + l31 * META_LD + 4) = make_uint4(m0.x, m0.y, m0.z, m0.w);
      *reinterpret_cast<uint4*>(strip + l31 * META_LD + 8) = make_uint4(m1.x, m1.y, m1.z, m1.w);
    }
    // read back by other lanes of this wave: LDS executes a wave's operations in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  float ac[H], an[H];              // MFMA A operands of the current unit, rows of the next one in flight
  int32_t nd_n, nd_nn, nd_nnn;     // node ids of tiles +1, +2 (rows in flight), +3 (id in flight)
  bool ok_n, ok_nn, ok_nnn;        // row inside the list (else it is stored to the spare row)
  u4u m0n, m1n;
  const auto row_offset = [&](int32_t nd, bool ok) { return (ok ? nd : int32_t(g.N)) * rowb; };
  {
    int32_t nd0;
    bool ok0;
    u4u m0, m1;
    load_id(first, nd0, ok0);
    load_id(tile_at(first + stride), nd_n, ok_n);
    load_id(tile_at(first + 2 * stride), nd_nn, ok_nn);
    load_rows(first, nd0, ac, m0, m1);
    park_meta(meta, row_offset(nd0, ok0), m0, m1);
    load_rows(tile_at(first + stride), nd_n, an, m0n, m1n);
  }
  const int sh = 31 - l31;
  int it = 0;
  // The wait-count pass merges the loop entry and the back edge at the loop header and keeps the stricter count: with
  // prologue loads pending it would re-issue their waits on every iteration, draining the stores of the iteration
  // before.  Entering the loop with nothing pending leaves only the back edge's own counts.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t tile = first; tile < ntiles; tile += stride, ++it) {
    const uint32_t plane = uint32_t(tile) / tiles_per_plane;
    char* __restrict__ Up = reinterpret_cast<char*>(g.U + int64_t(plane) * g.u_plane_stride + colx);
    const int32_t* __restrict__ mcur = meta + (it & 1) * (BGM * META_LD);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < H; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[kk], bw[kk], acc, 0, 0, 0);
    // land tile +1 (its offsets / mask words go to the other LDS strip), then issue rows(+2) and id(+3)
#pragma unroll
    for (int kk = 0; kk < H; ++kk) {
      asm volatile("" : "+v"(an[kk]));
      ac[kk] = an[kk];
    }
    park_meta(meta + ((it + 1) & 1) * (BGM * META_LD), row_offset(nd_n, ok_n), m0n, m1n);
    load_rows(tile_at(tile + 2 * stride), nd_nn, an, m0n, m1n);
    load_id(tile_at(tile + 3 * stride), nd_nnn, ok_nnn);
    nd_n = nd_nn; ok_n = ok_nn; nd_nn = nd_nnn; ok_nn = ok_nnn;

    // epilogue: offsets and mask words of 8 rows at a time, then their stores
#pragma unroll
    for (int rq = 0; rq < 2; ++rq) {
      int32_t offs[8];
      uint32_t wd[8];
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int r = rq * 8 + ri;
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        offs[ri] = mcur[lr * META_LD];
        wd[ri] = uint32_t(mcur[lr * META_LD + 4 + (RELU ? ch : 0)]);
      }
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        float v = acc[rq * 8 + ri];
        if constexpr (RELU) v = int32_t(wd[ri] << sh) < 0 ? v : 0.f;
        else if (g.hact) v *= act_deriv_from_out(g.hact[int64_t(wd[ri]) * g.hact_ld + colx], g.act);
        *reinterpret_cast<float*>(Up + uint32_t(col_ok ? offs[ri] : trash)) = v;
      }
    }
  }
}

// ---- producer / consumer variant for the hot shape (K % 8 == 0, ReLU mask, Nout == 256) ---------------------------
// In the streaming kernel above every wave loads, multiplies and stores, and hipcc's wait counts serialise a wave's
// stores with its next loads (vmcnt counts both, in order).  Here the roles are split like in the fused SpMM kernel:
//   * 4 loader waves fetch a stage of 128 rows (row list -> rows of G, mask words, output offsets) and put it into
//     one of two LDS buffers; they issue no stores, so their waits only ever concern their own loads;
//   * 8 compute waves (one per group of 32 output columns, W slice in registers) read the MFMA operands of the stage
//     from LDS and issue 20 MFMAs + 16 unconditional stores per 32-row tile; they issue no global loads inside the
//     loop, so nothing ever makes them wait for their stores.
// One s_barrier per stage, written as "s_waitcnt lgkmcnt(0); s_barrier" (LDS hand-over only): __syncthreads() would
// add a vmcnt(0) and drain the compute waves' stores.  Each row of G is read once per stage instead of once per column
// group.  One 768-thread workgroup per CU, persistent.
constexpr int PC_ROWS = 128, PC_LOADERS = 4, PC_COMPUTE = 8, PC_THREADS = 64 * (PC_LOADERS + PC_COMPUTE);

template <int KH>
__global__ __launch_bounds__(PC_THREADS) void backgemm_pc_kernel(BackGemmArgs g) {
  extern __shared__ float pc_smem[];
  constexpr int H = 4 * KH;
  constexpr int KP = 2 * H + 4;  // row stride of the staged rows: 16-byte aligned, conflict-free ds_read_b128
  constexpr int BUF_FLOATS = PC_ROWS * KP + PC_ROWS * META_LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  const int64_t na = int64_t(*g.na_dev);
  const uint32_t spp = uint32_t((na + PC_ROWS - 1) / PC_ROWS);  // stages per plane
  const int64_t total = int64_t(spp) * g.planes;
  if (int64_t(blockIdx.x) >= total) return;  // (uniform per workgroup, before any barrier)
  const int64_t nit = (total - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const auto stage_of = [&](int64_t j) {  // global stage of this workgroup's j-th; past the end: its last one
    const int64_t jj = j < nit ? j : nit - 1;
    return int64_t(blockIdx.x) + jj * gridDim.x;
  };
  const int32_t rowb = int32_t(g.Nout) * 4;
  const auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  if (wave < PC_LOADERS) {
    // ------------------------------------------------------------ loader waves
    // (written without lambdas over the staged registers: captured by reference they stay address-taken, and the
    //  "memory" clobber of the barrier then forces them through scratch memory)
    const int r = wave * 32 + l31;  // row of the stage
    const int mw = int(g.mask_words);
    // named registers, not an array: an array that is live across the barrier's memory clobber is kept in scratch
    float4 a0, a1, a2, a3, a4, a5, a6, a7;
    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = make_float4(0.f, 0.f, 0.f, 0.f);
    u4u m0, m1;
    int32_t nd, off;
    bool ok;
#define PC_EACH(OP) { if constexpr (KH > 0) OP(0, a0) if constexpr (KH > 1) OP(1, a1) if constexpr (KH > 2) OP(2, a2) \
    if constexpr (KH > 3) OP(3, a3) if constexpr (KH > 4) OP(4, a4) if constexpr (KH > 5) OP(5, a5) \
    if constexpr (KH > 6) OP(6, a6) if constexpr (KH > 7) OP(7, a7) }
#define PC_LD(Q, R) R = *reinterpret_cast<const float4*>(src_ + 4 * Q);
#define PC_ST(Q, R) *reinterpret_cast<float4*>(A_ + r * KP + lhi * H + 4 * Q) = R;
#define PC_ISSUE_ID(J)                                                            \
  {                                                                               \
    const uint32_t st_ = uint32_t(stage_of(J));                                   \
    const uint32_t plane_ = st_ / spp;                                            \
    const int64_t t_ = int64_t(st_ - plane_ * spp) * PC_ROWS + r;                 \
    ok = t_ < na;                                                                 \
    nd = g.rows[ok ? t_ : na - 1];                                                \
  }
#define PC_ISSUE_ROWS(J) /* uses nd / ok of stage J */                            \
  {                                                                               \
    const uint32_t plane_ = uint32_t(stage_of(J)) / spp;                          \
    const float* __restrict__ src_ = g.G + (int64_t(plane_) * g.N + nd) * g.K + lhi * H; \
    PC_EACH(PC_LD)                                                                \
    const uint32_t* __restrict__ mp_ = g.mask_bits + int64_t(nd) * mw;            \
    m0 = *reinterpret_cast<const u4u*>(mp_);                                      \
    m1 = *reinterpret_cast<const u4u*>(mp_ + 4);                                  \
    off = (ok ? nd : int32_t(g.N)) * rowb;                                        \
  }
#define PC_PARK(BUF)                                                              \
  {                                                                               \
    float* __restrict__ A_ = pc_smem + (BUF) * BUF_FLOATS;                        \
    int32_t* __restrict__ meta_ = reinterpret_cast<int32_t*>(A_ + PC_ROWS * KP);  \
    PC_EACH(PC_ST)                                                                \
    if (lhi == 0) {                                                               \
      meta_[r * META_LD] = off;                                                   \
      *reinterpret_cast<uint4*>(meta_ + r * META_LD + 4) = make_uint4(m0.x, m0.y, m0.z, m0.w); \
      *reinterpret_cast<uint4*>(meta_ + r * META_LD + 8) = make_uint4(m1.x, m1.y, m1.z, m1.w); \
    }                                                                             \
  }
    PC_ISSUE_ID(0)
    PC_ISSUE_ROWS(0)
    PC_ISSUE_ID(1)
    PC_PARK(0)        // stage 0 -> buffer 0
    PC_ISSUE_ROWS(1)  // rows of stage 1 in flight, id of stage 2 next
    PC_ISSUE_ID(2)
    barrier();
    for (int64_t i = 0; i < nit; ++i) {
      // stage i + 1 -> the buffer the compute waves are not reading; then its registers take stage i + 2
      PC_PARK(int((i + 1) & 1))
      PC_ISSUE_ROWS(i + 2)
      PC_ISSUE_ID(i + 3)
      barrier();
    }
#undef PC_ISSUE_ID
#undef PC_ISSUE_ROWS
#undef PC_PARK
#undef PC_EACH
#undef PC_LD
#undef PC_ST
  } else {
    // ------------------------------------------------------------ compute waves
    const int ch = wave - PC_LOADERS;  // column group
    const int col = ch * 32 + l31;
    float bw[H];
#pragma unroll
    for (int kk = 0; kk < H; ++kk) bw[kk] = g.W[int64_t(lhi * H + kk) * g.ldw + col];
    const int sh = 31 - l31;
    barrier();
    for (int64_t i = 0; i < nit; ++i) {
      const uint32_t plane = uint32_t(stage_of(i)) / spp;
      // wave-uniform plane base + 32-bit lane offset: the stores take the scalar-base form (no 64-bit vector adds)
      char* __restrict__ Up = reinterpret_cast<char*>(g.U + int64_t(plane) * g.u_plane_stride);
      const float* __restrict__ A = pc_smem + int(i & 1) * BUF_FLOATS;
      const int32_t* __restrict__ meta = reinterpret_cast<const int32_t*>(A + PC_ROWS * KP);
#pragma unroll
      for (int rt = 0; rt < PC_ROWS / 32; ++rt) {
        float av[H];
        const float* __restrict__ arow = A + (rt * 32 + l31) * KP + lhi * H;
#pragma unroll
        for (int q = 0; q < KH; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(arow + 4 * q);
          av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < H; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
        for (int rq = 0; rq < 2; ++rq) {
          int32_t offs[8];
          uint32_t wd[8];
#pragma unroll
          for (int ri = 0; ri < 8; ++ri) {
            const int e = rq * 8 + ri;
            const int lr = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            offs[ri] = meta[lr * META_LD];
            wd[ri] = uint32_t(meta[lr * META_LD + 4 + ch]);
          }
#pragma unroll
          for (int ri = 0; ri < 8; ++ri) {
            const float v = int32_t(wd[ri] << sh) < 0 ? acc[rq * 8 + ri] : 0.f;
            *reinterpret_cast<float*>(Up + uint32_t(offs[ri] + col * 4)) = v;
          }
        }
      }
      barrier();
    }
  }
}

template <int KH>
int backgemm_pc_launch(const BackGemmArgs& g, hipStream_t s) {
  constexpr int H = 4 * KH, KP = 2 * H + 4;
  const size_t smem = size_t(2) * (PC_ROWS * KP + PC_ROWS * META_LD) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    LGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&backgemm_pc_kernel<KH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    attr_set = true;
  }
  const int64_t worst = cdiv(g.N, PC_ROWS) * g.planes;
  const unsigned grid = unsigned(std::max<int64_t>(1, std::min<int64_t>(worst, 256)));
  hipLaunchKernelGGL(backgemm_pc_kernel<KH>, dim3(grid), dim3(PC_THREADS), smem, s, g);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int KH>
int backgemm_launch(const BackGemmArgs& g, hipStream_t s) {
  const size_t smem = size_t(2 * BG_WAVES * BGM * META_LD) * 4;
  const int64_t nch = cdiv(g.Nout, 32);
  const int64_t units = cdiv(g.N, BGM) * g.planes * nch;
  // waves: a multiple of the column groups and of the workgroup size; at most BG_OCC waves per SIMD on 256 CUs
  const int64_t quantum = nch * BG_WAVES / std::gcd<int64_t>(nch, BG_WAVES);
  int64_t waves = std::min<int64_t>(units, 256 * 4 * (BG_OCC + 1));  // K = 40 compiles to 126 VGPRs: 4 waves fit
  waves = std::max<int64_t>(quantum, waves / quantum * quantum);
  const unsigned grid = unsigned(waves / BG_WAVES);
  const bool vec = g.K == 8 * KH && (reinterpret_cast<uintptr_t>(g.G) & 15) == 0;
  const bool relu = g.mask_bits != nullptr;
  if (vec && relu) hipLaunchKernelGGL((backgemm_kernel<KH, true, true>), dim3(grid), dim3(BG_THREADS), smem, s, g);
  else if (vec) hipLaunchKernelGGL((backgemm_kernel<KH, true, false>), dim3(grid), dim3(BG_THREADS), smem, s, g);
  else if (relu) hipLaunchKernelGGL((backgemm_kernel<KH, false, true>), dim3(grid), dim3(BG_THREADS), smem, s, g);
  else hipLaunchKernelGGL((backgemm_kernel<KH, false, false>), dim3(grid), dim3(BG_THREADS), smem, s, g);
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

// with_mask: the ReLU bit mask has 8 words per node (256 columns); without it the width is only bounded by the
// 32-bit row offsets (checked at launch)
bool backgemm_supported(int64_t K, int64_t Nout, bool with_mask) {
  return K >= 1 && K <= 64 && Nout >= 1 && Nout <= (with_mask ? 256 : 1024);
}

int launch_backgemm(const BackGemmArgs& g_in, hipStream_t s) {
  BackGemmArgs g = g_in;
  if (g.planes <= 0 || g.N <= 0) return 0;
  LGNN_REQUIRE(backgemm_supported(g.K, g.Nout, g.mask_bits != nullptr), "backgemm shape not supported");
  LGNN_REQUIRE(g.mask_bits == nullptr || g.mask_words <= 8, "backgemm: at most 8 mask words per node");
  LGNN_REQUIRE(cdiv(g.N, BGM) * g.planes < (int64_t(1) << 31), "backgemm: too many row tiles for 32-bit indices");
  LGNN_REQUIRE((g.N + 1) * g.Nout * 4 < (int64_t(1) << 31), "backgemm: plane too large for 32-bit row offsets");
  LGNN_REQUIRE(g.rows && g.na_dev, "backgemm needs the compacted row list");
  LGNN_REQUIRE(g.u_plane_stride >= (g.N + 1) * g.Nout, "backgemm: planes of U need a spare row");
  const int kh = int(cdiv(cdiv(g.K, 2), 4));  // 16-byte pieces per half row
  bool hot = g.mask_bits && g.mask_words == 8 && g.Nout == 256 && g.K == 8 * kh &&
             (reinterpret_cast<uintptr_t>(g.G) & 15) == 0 && g.ldw >= 256;
#ifdef LGNN_DEV  // make DEV=1: LGNN_BACKGEMM_STREAM=1 forces the streaming kernel for A/B timing
  if (getenv("LGNN_BACKGEMM_STREAM")) hot = false;
#endif
  if (hot) {
    LGNN_REQUIRE(cdiv(g.N, PC_ROWS) * g.planes < (int64_t(1) << 31), "backgemm: too many stages for 32-bit indices");
    switch (kh) {
      case 1: return backgemm_pc_launch<1>(g, s);
      case 2: return backgemm_pc_launch<2>(g, s);
      case 3: return backgemm_pc_launch<3>(g, s);
      case 4: return backgemm_pc_launch<4>(g, s);
      case 5: return backgemm_pc_launch<5>(g, s);
      case 6: return backgemm_pc_launch<6>(g, s);
      case 7: return backgemm_pc_launch<7>(g, s);
      default: return backgemm_pc_launch<8>(g, s);
    }
  }
  switch (kh) {
    case 1: return backgemm_launch<1>(g, s);
    case 2: return backgemm_launch<2>(g, s);
    case 3: return backgemm_launch<3>(g, s);
    case 4: return backgemm_launch<4>(g, s);
    case 5: return backgemm_launch<5>(g, s);
    case 6: return backgemm_launch<6>(g, s);
    case 7: return backgemm_launch<7>(g, s);
    default: return backgemm_launch<8>(g, s);
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
