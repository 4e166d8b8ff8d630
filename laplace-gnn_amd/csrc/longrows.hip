// Long rows of the backward propagation matrix P^T (hubs of a power-law graph).
//
// The fused SpMM^T -> Gram kernel (fused256.hip) gives every row to ONE gather wave; a row with thousands of stored
// entries keeps that wave -- and, at the block's barrier, its whole workgroup -- busy for hundreds of microseconds while
// the other 255 workgroups move on (power-law graph of the arxiv size: +40 % per launch).  Rows with more than kLongRow
// entries are therefore computed here, by a whole workgroup per (row slice, plane) with all four waves gathering, and
// handed to the fused kernel as finished rows ("hub" buffer): it fetches them like a self row and gathers nothing.
// The list of long rows depends on the graph only and is built once per context.
#include "lgnn_internal.h"

namespace lgnn {

namespace {

constexpr int kTask = 1024;  // stored entries per workgroup task; longer rows are split and combined with float atomics

__global__ void flag_long_rows_kernel(const int32_t* __restrict__ rowptr, int64_t N, uint8_t* __restrict__ flags) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < N) flags[r] = (rowptr[r + 1] - rowptr[r] > kLongRow) ? 1 : 0;
}
__global__ void long_slots_kernel(const int32_t* __restrict__ rows, int64_t n, const int32_t* __restrict__ rowptr,
                                  int32_t* __restrict__ slot, int32_t* __restrict__ bounds) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rows[i];
  slot[r] = int32_t(i);
  bounds[2 * i] = rowptr[r];
  bounds[2 * i + 1] = rowptr[r + 1];
}

// task t = (slot, begin, end): hub[plane][slot][:] += sum_{p in [begin, end)} val[p] * in[plane][col[p]][:]
// 256 threads: wave w takes entries begin + w, begin + w + 4, ... ; lane = 4 columns (width <= 256); 8 rows in flight.
__global__ __launch_bounds__(256) void long_rows_spmm_kernel(const int32_t* __restrict__ tasks,
                                                             const int32_t* __restrict__ col, const float* __restrict__ val,
                                                             const float* __restrict__ in, int64_t in_ld,
                                                             int64_t in_plane_stride, int64_t width,
                                                             float* __restrict__ hub, int64_t hub_plane_stride) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t slot = tasks[3 * blockIdx.x], begin = tasks[3 * blockIdx.x + 1], end = tasks[3 * blockIdx.x + 2];
  const int64_t plane = blockIdx.y;
  const float* __restrict__ inp = in + plane * in_plane_stride;
  const int c0 = lane * 4;
  const bool col_ok = c0 < width;
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int UNR = 8;
  for (int32_t p0 = begin + wave * UNR; p0 < end; p0 += 4 * UNR) {
    float v[UNR];
    int32_t j[UNR];
    float4 x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = p0 + u < end;
      v[u] = ok ? val[p0 + u] : 0.f;
      j[u] = ok ? col[p0 + u] : 0;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v[u] != 0.f && col_ok) x[u] = *reinterpret_cast<const float4*>(inp + int64_t(j[u]) * in_ld + c0);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      y.x += v[u] * x[u].x; y.y += v[u] * x[u].y; y.z += v[u] * x[u].z; y.w += v[u] * x[u].w;
    }
  }
  *reinterpret_cast<float4*>(&red[wave][c0]) = y;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < width) {
    const float sum = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    atomicAdd(&hub[plane * hub_plane_stride + int64_t(slot) * width + t], sum);
  }
}

}  // namespace

int long_rows_ensure(lgnn_ctx* h, hipStream_t s) {
  if (h->n_long >= 0) return 0;
  const int64_t N = h->N;
  h->n_long = 0;
  h->n_long_tasks = 0;
  if (h->nnz <= 0) return 0;
  DevBuf flags, cnt, bounds, tmp;
  auto done = [&](int rc) { flags.release(); cnt.release(); bounds.release(); tmp.release(); return rc; };
  if (flags.reserve(size_t(N)) || cnt.reserve(64)) return done(1);
  hipLaunchKernelGGL(flag_long_rows_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->PT.rowptr, N, flags.as<uint8_t>());
  if (h->long_rows.reserve(size_t(N) * 4)) return done(1);
  if (compact_flags(flags.as<uint8_t>(), N, h->long_rows.as<int32_t>(), cnt.as<int32_t>(), tmp, s)) return done(1);
  int32_t n = 0;
  if (hipMemcpyAsync(&n, cnt.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("long rows: count copy failed");
    return done(1);
  }
  if (n == 0) return done(0);
  if (h->long_slot.reserve(size_t(N) * 4) || bounds.reserve(size_t(n) * 8)) return done(1);
  if (launch_fill_i32(h->long_slot.as<int32_t>(), N, -1, s)) return done(1);
  hipLaunchKernelGGL(long_slots_kernel, dim3(unsigned(cdiv(n, 256))), dim3(256), 0, s, h->long_rows.as<int32_t>(), int64_t(n),
                     h->PT.rowptr, h->long_slot.as<int32_t>(), bounds.as<int32_t>());
  std::vector<int32_t> hb(size_t(n) * 2), tasks;
  if (hipMemcpyAsync(hb.data(), bounds.p, size_t(n) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_error("long rows: bounds copy failed");
    return done(1);
  }
  for (int32_t i = 0; i < n; ++i)
    for (int32_t b = hb[2 * i]; b < hb[2 * i + 1]; b += kTask) {
      tasks.push_back(i);
      tasks.push_back(b);
      tasks.push_back(std::min<int32_t>(b + kTask, hb[2 * i + 1]));
    }
  if (h->long_tasks.reserve(tasks.size() * 4)) return done(1);
  if (hipMemcpyAsync(h->long_tasks.p, tasks.data(), tasks.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_error("long rows: task upload failed");
    return done(1);
  }
  // hubs the top-layer kernel cuts into slices of kTopSlice entries (kfac.hip)
  std::vector<int32_t> ids(static_cast<size_t>(n)), multi;
  if (hipMemcpyAsync(ids.data(), h->long_rows.p, size_t(n) * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    set_error("long rows: id copy failed");
    return done(1);
  }
  int64_t slices = 0;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t deg = hb[2 * i + 1] - hb[2 * i];
    if (deg > kTopSlice) { multi.push_back(ids[i]); slices += (deg + kTopSlice - 1) / kTopSlice; }
  }
  const int64_t n_multi = int64_t(multi.size());
  if (!multi.empty()) {
    multi.push_back(int32_t(n_multi));  // the list's length behind it: the finishing launch reads its task count there
    if (h->top_multi.reserve(multi.size() * 4)) return done(1);
    if (hipMemcpyAsync(h->top_multi.p, multi.data(), multi.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      set_error("long rows: hub list upload failed");
      return done(1);
    }
  }
  h->n_top_multi = n_multi;
  h->n_top_slices = slices;
  h->n_long = n;
  h->n_long_tasks = int64_t(tasks.size() / 3);
  return done(0);
}

int long_rows_fwd_ensure(lgnn_ctx* h, hipStream_t s) {
  if (h->n_long_fwd >= 0) return 0;
  h->n_long_fwd = 0;
  if (h->nnz <= 0) return 0;
  if (h->P.rowptr == h->PT.rowptr) {  // symmetric graph: one structure serves both directions
    LGNN_CALL(long_rows_ensure(h, s));
    h->n_long_fwd = h->n_long;
    return 0;
  }
  const int64_t N = h->N;
  DevBuf flags, cnt, tmp;
  auto done = [&](int rc) { flags.release(); cnt.release(); tmp.release(); return rc; };
  if (flags.reserve(size_t(N)) || cnt.reserve(64)) return done(1);
  hipLaunchKernelGGL(flag_long_rows_kernel, dim3(unsigned(cdiv(N, 256))), dim3(256), 0, s, h->P.rowptr, N, flags.as<uint8_t>());
  if (h->long_rows_fwd.reserve(size_t(N) * 4)) return done(1);
  if (compact_flags(flags.as<uint8_t>(), N, h->long_rows_fwd.as<int32_t>(), cnt.as<int32_t>(), tmp, s)) return done(1);
  int32_t n = 0;
  if (hipMemcpyAsync(&n, cnt.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("long rows: count copy failed");
    return done(1);
  }
  h->n_long_fwd = n;
  return done(0);
}

int launch_long_rows_spmm(lgnn_ctx* h, const float* val, const float* in, int64_t in_ld, int64_t in_plane_stride,
                          int64_t nplanes, int64_t width, hipStream_t s) {
  if (h->n_long <= 0 || nplanes <= 0) return 0;
  LGNN_REQUIRE(width <= 256 && width % 4 == 0 && in_ld % 4 == 0 && in_plane_stride % 4 == 0, "long rows: alignment");
  LGNN_REQUIRE(nplanes < 65536, "too many planes for one launch");
  const size_t bytes = size_t(nplanes) * h->n_long * width * 4;
  LGNN_CALL(h->hub.reserve(bytes));
  LGNN_HIP_CHECK(hipMemsetAsync(h->hub.p, 0, bytes, s));
  hipLaunchKernelGGL(long_rows_spmm_kernel, dim3(unsigned(h->n_long_tasks), unsigned(nplanes)), dim3(256), 0, s,
                     h->long_tasks.as<int32_t>(), h->PT.col, val, in, in_ld, in_plane_stride, width, h->hub.as<float>(),
                     h->n_long * width);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace lgnn
