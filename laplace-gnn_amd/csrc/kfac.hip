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
__global__ __launch_bounds__(64) void seed_kernel(const float* __restrict__ logits, int64_t C,
                                                  const int64_t* __restrict__ idx, const int64_t* __restrict__ y,
                                                  int64_t M, int64_t N, const int32_t* __restrict__ pos, int fork_exact,
                                                  float* __restrict__ seeds, float* __restrict__ probs,
                                                  float* __restrict__ loss, int* __restrict__ bad) {
  extern __shared__ float sm[];
  float* f_s = sm;
  float* p_s = sm + C;
  const int lane = threadIdx.x;
  const int64_t m = blockIdx.x;
  const int64_t n = idx[m];
  if (n < 0 || n >= N) return;  // flagged by mark_batch_kernel
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
  __syncthreads();
  if (lane == 0 && loss) {
    const int64_t yy = y[m];
    if (yy < 0 || yy >= C) *bad = 2;
    else atomicAdd(loss, logf(se) + mx - f_s[yy]);
  }
  if (!seeds) return;
  const int64_t first = pos[n];
  float* __restrict__ dst = seeds + first * C * C;
  const int64_t CC = C * C;
  for (int64_t q = lane; q < CC; q += 64) {
    const int64_t c = q / C, k = q - c * C;
    const float pc = p_s[c], pk = p_s[k];
    const float d = (k == c) ? 1.f : 0.f;
    float v;
    if (fork_exact) v = sqrtf(pc) * (d - pk * (1.f + f_s[k] - mb) + 0.5f * (d - pk) * (f_s[c] - mb));
    else v = sqrtf(pc) * (d - pk);
    atomicAdd(&dst[q], v);
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

int record_event(lgnn_ctx* h, hipStream_t s) {
  if (h->ev_used >= h->ev.size()) {
    hipEvent_t e;
    LGNN_HIP_CHECK(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  LGNN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used++], s));
  return 0;
}

}  // namespace

// shared by kfac / diag / last layer: mark the batch, compute seeds/probs/loss.
int batch_prologue(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, bool want_seeds, bool fork_exact,
                   float* loss_out, hipStream_t s) {
  LGNN_REQUIRE(h->lik == LGNN_LIK_CLASSIFICATION, "only the classification likelihood is implemented on the GPU path");
  const int64_t N = h->N, C = h->dims[h->L];
  LGNN_REQUIRE(2 * C * 4 <= 64 * 1024, "too many classes for the seed kernel");
  int* bad = h->ws.flags.as<int>();  // allocated and zeroed by lgnn_create; sticky until lgnn_check_async_errors
  hipLaunchKernelGGL(mark_batch_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, N,
                     h->ws.pos.as<int32_t>(), bad);
  LGNN_CALL(h->ws.probs.reserve(size_t(M) * C * 4));
  float* seeds = nullptr;
  if (want_seeds) {
    LGNN_CALL(h->ws.seeds.reserve(size_t(M) * C * C * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(h->ws.seeds.p, 0, size_t(M) * C * C * 4, s));
    seeds = h->ws.seeds.as<float>();
  }
  hipLaunchKernelGGL(seed_kernel, dim3(unsigned(M)), dim3(64), size_t(2 * C) * 4, s, h->fc.out.as<float>(), C, idx,
                     static_cast<const int64_t*>(y), M, N, h->ws.pos.as<int32_t>(), fork_exact ? 1 : 0, seeds,
                     h->ws.probs.as<float>(), loss_out, bad);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int batch_epilogue(lgnn_ctx* h, const int64_t* idx, int64_t M, hipStream_t s) {
  hipLaunchKernelGGL(unmark_batch_kernel, dim3(unsigned(cdiv(M, 256))), dim3(256), 0, s, idx, M, h->N,
                     h->ws.pos.as<int32_t>());
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

int kfac_accumulate(lgnn_ctx* h, const int64_t* idx, const void* y, int64_t M, int64_t n_train, uint32_t flags,
                    int64_t cb, int64_t ce, float* const* A_out, float* const* B_out, float* loss_out, hipStream_t s) {
  LGNN_REQUIRE(M > 0 && idx && y, "empty batch or null batch pointers");
  LGNN_REQUIRE(h->L > 0, "no model bound");
  LGNN_REQUIRE(cb >= 0 && cb < ce && ce <= h->dims[h->L], "class range must satisfy 0 <= begin < end <= C");
  // B_l = sum over class columns c of g_c^T g_c: a class range is an exact additive share of the batch.
  // The share that contains class 0 also carries what exists once per batch: the loss and the A increment
  // (and, for GraphSAGE, the whole top-layer Gram, which is only M*C rows).
  const bool first = cb == 0;
  LGNN_REQUIRE(n_train > 0, "n_train must be positive");
  LGNN_REQUIRE(A_out && B_out && loss_out, "null output pointers");
  LGNN_CALL(forward_ensure_grams(h, s));
  const int64_t N = h->N;
  const int L = h->L;
  const int64_t C = h->dims[L];
  const int64_t CC = C * C;
  const bool no_fuse = (flags & LGNN_FLAG_NO_FUSE) != 0;
  LGNN_REQUIRE(M < INT32_MAX, "batch too large");

  LGNN_CALL(batch_prologue(h, idx, y, M, true, (flags & LGNN_FLAG_FORK_EXACT_SEED) != 0, first ? loss_out : nullptr, s));

  // A_l += in_l^T in_l / n_train   (kfac.py:870 divides by M, curvlinops.py:46-53 multiplies by M/N)
  for (int l = 0; first && l < L; ++l)
    LGNN_CALL(launch_sym_accumulate(h->fc.gram_raw[l].as<float>(), h->in_dim[l], 1.0f / float(n_train), A_out[l], s));

  for (int l = 0; l < L; ++l) {
    const int64_t D = h->dims[l + 1];
    LGNN_CALL(h->ws.gram_scratch[l].reserve(size_t(D) * D * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(h->ws.gram_scratch[l].p, 0, size_t(D) * D * 4, s));
  }

  // ---- top layer ---------------------------------------------------------------------------------
  float* gtop = nullptr;  // planes [C][N][C]
  if (h->kind == LGNN_KIND_GCN || L > 1) {
    LGNN_CALL(h->ws.top.reserve(size_t(N) * CC * 4));
    gtop = h->ws.top.as<float>();
  }
  LGNN_CALL(h->ws.active.reserve(size_t(N)));
  if (h->kind == LGNN_KIND_GCN) {
    const int64_t nq = (ce - cb) * C;
    int waves = int(std::max<int64_t>(1, std::min<int64_t>(4, (48 * 1024) / (nq * 4))));
    LGNN_REQUIRE(nq * 4 <= 60 * 1024, "too many classes for the seed SpMM kernel (use class ranges)");
    hipLaunchKernelGGL(seed_spmm_kernel, dim3(unsigned(cdiv(N, waves))), dim3(64 * waves), size_t(waves) * nq * 4, s,
                       h->PT.rowptr, h->PT.col, h->PT.val, N, C, h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), gtop,
                       h->ws.active.as<uint8_t>(), cb, ce);
    LGNN_HIP_CHECK(hipGetLastError());
    LGNN_CALL(launch_gram(gtop + cb * N * C, C, (ce - cb) * N, C, h->ws.gram_scratch[L - 1].as<float>(), s));
  } else {
    // rows (m, c) of the accumulated seeds; rows of non-first duplicates are zero
    if (first) LGNN_CALL(launch_gram(h->ws.seeds.as<float>(), C, M * C, C, h->ws.gram_scratch[L - 1].as<float>(), s));
    if (L > 1) {
      LGNN_HIP_CHECK(hipMemsetAsync(gtop + cb * N * C, 0, size_t(N) * (ce - cb) * C * 4, s));
      hipLaunchKernelGGL(scatter_seed_planes_kernel, dim3(unsigned(M)), dim3(256), 0, s, idx, M, N, C,
                         h->ws.pos.as<int32_t>(), h->ws.seeds.as<float>(), gtop, cb, ce);
      LGNN_HIP_CHECK(hipGetLastError());
    }
  }

  // ---- lower layers, chunked over classes ----------------------------------------------------------
  if (L > 1) {
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
      if (gcn) {
        row_active = h->ws.active.as<uint8_t>();
        LGNN_CALL(h->ws.act_list.reserve(size_t(N) * 4));
        LGNN_CALL(h->ws.act_count.reserve(64));
        LGNN_CALL(compact_flags(row_active, N, h->ws.act_list.as<int32_t>(), h->ws.act_count.as<int32_t>(),
                                h->ws.select_tmp, s));
      }
    }
    int64_t maxw = 0;
    for (int l = 0; l < L - 1; ++l) maxw = std::max(maxw, h->in_dim[l + 1]);  // GEMM output width of layer l+1
    const int64_t per_class = N * maxw * 4 * 2;  // ping + pong
    int64_t cc_max = std::max<int64_t>(1, std::min<int64_t>(C, h->ws_limit / std::max<int64_t>(per_class, 1)));
    LGNN_CALL(h->ws.planes_a.reserve(size_t(cc_max) * N * maxw * 4));
    LGNN_CALL(h->ws.planes_b.reserve(size_t(cc_max) * N * maxw * 4));
    for (int64_t c0 = cb; c0 < ce; c0 += cc_max) {
      const int64_t cc = std::min(cc_max, ce - c0);
      const float* g = gtop + c0 * N * C;  // planes [cc][N][dims[l+1]] of layer l
      float* ping = h->ws.planes_a.as<float>();
      float* pong = h->ws.planes_b.as<float>();
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
          const bool fuse_here = !no_fuse && fused_supported(d, d, N * d, ping);
          if (top_level && fuse_here) ep.row_active = row_active;  // inactive rows are never read below
          if (top_level && fuse_here && row_active && backgemm_supported(dout, d)) {
            BackGemmArgs bg{};
            bg.G = g; bg.W = h->W[l]; bg.ldw = d; bg.U = ping; bg.N = N; bg.K = dout; bg.Nout = d; bg.planes = cc;
            bg.rows = h->ws.act_list.as<int32_t>(); bg.na_dev = h->ws.act_count.as<int32_t>();
            if (h->act == LGNN_ACT_RELU) { bg.mask_bits = h->fc.mask_bits[l - 1].as<uint32_t>(); bg.mask_words = cdiv(d, 32); }
            else { bg.hact = ep.hact; bg.hact_ld = ep.hact_ld; bg.act = h->act; }
            LGNN_CALL(launch_backgemm(bg, s));
          } else {
            LGNN_CALL(launch_gemm(g, dout, h->W[l], d, ping, d, cc * N, dout, d, ep, s));
          }
          FusedArgs a{};
          a.rowptr = h->PT.rowptr; a.col = h->PT.col; a.val = (top_level && fuse_here) ? val_top : h->PT.val;
          a.nrows = N; a.nplanes = cc;
          a.in = ping; a.in_ld = d; a.in_plane_stride = N * d;
          a.store = store ? pong : nullptr; a.store_ld = d; a.store_plane_stride = N * d;
          a.width = d; a.scratch = scratch;
          if (fuse_here) {
            if (h->timing && dominant) LGNN_CALL(record_event(h, s));
            LGNN_CALL(launch_spmm_gram_ex(a, s));
            if (h->timing && dominant) { LGNN_CALL(record_event(h, s)); h->ev_planes += cc; }
          } else {
            SpmmArgs sa{};
            sa.rowptr = a.rowptr; sa.col = a.col; sa.val = a.val; sa.nrows = N;
            sa.in = ping; sa.in_ld = d; sa.in_plane_stride = N * d;
            sa.out = pong; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
            LGNN_CALL(launch_spmm_ex(sa, cc, s));
            LGNN_CALL(launch_gram(pong, d, cc * N, d, scratch, s));
          }
        } else {
          // dcat = g W_l  [cc*N, 2d];  g_{l-1} = act'(h_l) * (dcat[:, :d] + P^T dcat[:, d:])
          GemmEpilogue ep;
          LGNN_CALL(launch_gemm(g, dout, h->W[l], 2 * d, ping, 2 * d, cc * N, dout, 2 * d, ep, s));
          FusedArgs a{};
          a.rowptr = h->PT.rowptr; a.col = h->PT.col; a.val = (l == L - 1) ? val_top : h->PT.val;
          a.nrows = N; a.nplanes = cc;
          a.in = ping + d; a.in_ld = 2 * d; a.in_plane_stride = N * 2 * d;
          a.self = ping; a.self_ld = 2 * d; a.self_plane_stride = N * 2 * d;
          a.hact = h->fc.hact_p[l - 1]; a.hact_ld = h->fc.hact_ld[l - 1]; a.act = h->act;
          a.store = store ? pong : nullptr; a.store_ld = d; a.store_plane_stride = N * d;
          a.width = d; a.scratch = scratch;
          if (!no_fuse && fused_supported(d, a.in_ld, a.in_plane_stride, a.in) && h->fc.hact_ld[l - 1] % 4 == 0) {
            if (h->timing && dominant) LGNN_CALL(record_event(h, s));
            LGNN_CALL(launch_spmm_gram_ex(a, s));
            if (h->timing && dominant) { LGNN_CALL(record_event(h, s)); h->ev_planes += cc; }
          } else {
            SpmmArgs sa{};
            sa.rowptr = a.rowptr; sa.col = a.col; sa.val = a.val; sa.nrows = N;
            sa.in = a.in; sa.in_ld = a.in_ld; sa.in_plane_stride = a.in_plane_stride;
            sa.self = a.self; sa.self_ld = a.self_ld; sa.self_plane_stride = a.self_plane_stride;
            sa.hact = a.hact; sa.hact_ld = a.hact_ld; sa.act = a.act;
            sa.out = pong; sa.out_ld = d; sa.out_plane_stride = N * d; sa.width = d; sa.out_act = -1;
            LGNN_CALL(launch_spmm_ex(sa, cc, s));
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
    LGNN_CALL(launch_sym_accumulate(h->ws.gram_scratch[l].as<float>(), h->dims[l + 1], 1.0f, B_out[l], s));
  LGNN_CALL(batch_epilogue(h, idx, M, s));
  return 0;
}

}  // namespace lgnn
