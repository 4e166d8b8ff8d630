// Graph ingest on the GPU: COO int64 edge list -> deduplicated row-major CSR of the 0/1 adjacency
// the reference keeps as a dense N x N parameter, its transpose, and the fp32 values of the
// propagation matrix.  Integer path: bit exact with the reference's dense construction
// (gnn/utils.py:325-330, gnn/marglik_training.py:405, gnn/models/models.py:23,47,
//  gnn/models/base_gnn.py:68-72); values: gnn/models/utils.py:106-112, gnn/models/layers.py:18-24.
//
// HBM-bound integer work: one 64-bit radix sort of (row*N+col) keys (rocPRIM), adjacent-unique,
// histogram + scan for the row pointers; a second sort of (col*N+row) for the transpose.
#include "lgnn_internal.h"

#include <rocprim/rocprim.hpp>

namespace lgnn {

namespace {

constexpr uint64_t kSentinel = ~uint64_t(0);

__global__ void make_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int symmetric,
                                 int add_loops, uint64_t* __restrict__ keys, int* __restrict__ bad) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  const int64_t per = symmetric ? 2 * E : E;
  for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < per + (add_loops ? N : 0); e += stride) {
    uint64_t key;
    if (e < E) {
      int64_t s = ei[e], t = ei[E + e];
      if (s < 0 || s >= N || t < 0 || t >= N) {
        *bad = 1;
        key = kSentinel;
      } else {
        key = (s == t) ? kSentinel : uint64_t(s) * uint64_t(N) + uint64_t(t);  // diagonal is overwritten
      }
    } else if (e < per) {
      int64_t s = ei[e - E], t = ei[E + e - E];
      if (s < 0 || s >= N || t < 0 || t >= N) key = kSentinel;
      else key = (s == t) ? kSentinel : uint64_t(t) * uint64_t(N) + uint64_t(s);
    } else {
      uint64_t n = uint64_t(e - per);
      key = n * uint64_t(N) + n;  // GCN: fill_diagonal_(1)
    }
    keys[e] = key;
  }
}

// keys sorted + unique (sentinel possibly last) -> col, row histogram
__global__ void split_keys_kernel(const uint64_t* __restrict__ keys, int64_t nnz, int64_t N,
                                  int32_t* __restrict__ col, int32_t* __restrict__ rowcount) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nnz; i += stride) {
    uint64_t k = keys[i];
    uint64_t r = k / uint64_t(N);
    col[i] = int32_t(k - r * uint64_t(N));
    atomicAdd(&rowcount[r], 1);
  }
}

__global__ void transpose_keys_kernel(const uint64_t* __restrict__ keys, int64_t nnz, int64_t N,
                                      uint64_t* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nnz; i += stride) {
    uint64_t k = keys[i];
    uint64_t r = k / uint64_t(N), c = k - r * uint64_t(N);
    out[i] = c * uint64_t(N) + r;
  }
}

__global__ void compare_keys_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t n,
                                    int* __restrict__ differ) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    if (a[i] != b[i]) *differ = 1;
}

// scale[i]: GCN rowsum(A)^-1/2 with inf -> 0 ; SAGE 1 / max(rowsum, 1)
__global__ void degree_scale_kernel(const int32_t* __restrict__ rowptr, int64_t N, int kind,
                                    float* __restrict__ scale) {
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float deg = float(rowptr[i + 1] - rowptr[i]);
  float v;
  if (kind == LGNN_KIND_GCN) {
    v = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;  // pow(-0.5); isinf -> 0 (utils.py:108-109)
  } else {
    v = 1.0f / (deg == 0.f ? 1.0f : deg);  // row_sum[row_sum == 0] = 1 (layers.py:19-20)
  }
  scale[i] = v;
}

// values aligned with a CSR (rowptr/col): mode 0: scale[row]*scale[col]; 1: scale[row]; 2: scale[col]
__global__ void csr_values_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                  const float* __restrict__ scale, int mode, float* __restrict__ val) {
  // one wave per row
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (row >= N) return;
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  const float sr = scale[row];
  for (int32_t p = s + lane; p < e; p += 64) {
    float v;
    if (mode == 0) v = sr * scale[col[p]];
    else if (mode == 1) v = sr;
    else v = scale[col[p]];
    val[p] = v;
  }
}

__global__ void export_coo_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                  const float* __restrict__ val, int64_t N, int64_t* __restrict__ rows,
                                  int64_t* __restrict__ cols, float* __restrict__ vals) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (row >= N) return;
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  for (int32_t p = s + lane; p < e; p += 64) {
    rows[p] = row;
    cols[p] = col[p];
    if (vals) vals[p] = val[p];
  }
}

// adj_to_edge_index: per row count of off-diagonal entries, then compacted write
__global__ void offdiag_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                     int64_t N, int32_t* __restrict__ cnt) {
  int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= N) return;
  int32_t c = 0;
  for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) c += (col[p] != row);
  cnt[row] = c;
}
__global__ void offdiag_write_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                     int64_t N, const int32_t* __restrict__ off, int64_t total,
                                     int64_t* __restrict__ out) {
  int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= N) return;
  int64_t o = off[row];
  for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
    if (col[p] != row) {
      out[o] = row;
      out[total + o] = col[p];
      ++o;
    }
  }
}


// ---- in-place edits of the stored 0/1 adjacency (lgnn_update_adjacency) ---------------------------------------------------
// one wave per row: the sorted key list (row * N + col) of a CSR
__global__ void expand_keys_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                   uint64_t* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (row >= N) return;
  for (int32_t p = rowptr[row] + lane; p < rowptr[row + 1]; p += 64) keys[p] = uint64_t(row) * uint64_t(N) + uint64_t(col[p]);
}
// flips (i, j, state) -> keys of the adjacency and of its transpose; the diagonal is not editable (a GCN's self loops are
// overwritten ones, GraphSAGE's zeros: gnn/models/models.py:23, 47, 114): sentinel
__global__ void flip_keys_kernel(const int64_t* __restrict__ fi, const int64_t* __restrict__ fj, int64_t K, int64_t N,
                                 uint64_t* __restrict__ keys, uint64_t* __restrict__ tkeys, int* __restrict__ bad) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t i = fi[k], j = fj[k];
  uint64_t a = kSentinel, b = kSentinel;
  if (i < 0 || i >= N || j < 0 || j >= N) *bad = 1;
  else if (i != j) { a = uint64_t(i) * uint64_t(N) + uint64_t(j); b = uint64_t(j) * uint64_t(N) + uint64_t(i); }
  keys[k] = a; tkeys[k] = b;
}
__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
// sorted flips against the sorted stored keys: add[k] = 1 for a flip that inserts, rem[p] = 1 for a stored entry that goes;
// counters {inserted, removed}; dup: the same pair listed twice
__global__ void flip_classify_kernel(const uint64_t* __restrict__ old, int64_t nnz, const uint64_t* __restrict__ fk,
                                     const uint8_t* __restrict__ st, int64_t K, int32_t* __restrict__ add,
                                     int32_t* __restrict__ rem, int* __restrict__ counters, int* __restrict__ dup) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const uint64_t key = fk[k];
  int a = 0;
  if (key != kSentinel) {
    if (k > 0 && fk[k - 1] == key) *dup = 1;
    const int64_t lb = lower_bound_u64(old, nnz, key);
    const bool present = lb < nnz && old[lb] == key;
    if (st[k] && !present) { a = 1; atomicAdd(&counters[0], 1); }
    if (!st[k] && present) { rem[lb] = 1; atomicAdd(&counters[1], 1); }
  }
  add[k] = a;
}
// the flips of a graph whose transpose aliases it must come in mirrored pairs, or the alias ends here
__global__ void flip_symmetry_kernel(const uint64_t* __restrict__ fk, const uint8_t* __restrict__ st, int64_t K, int64_t N,
                                     int* __restrict__ asym) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K || fk[k] == kSentinel) return;
  const uint64_t i = fk[k] / uint64_t(N), j = fk[k] - i * uint64_t(N);
  const uint64_t t = j * uint64_t(N) + i;
  const int64_t lb = lower_bound_u64(fk, K, t);
  if (!(lb < K && fk[lb] == t && st[lb] == st[k])) *asym = 1;
}
// the merged key list: kept stored entries ...
__global__ void merge_kept_kernel(const uint64_t* __restrict__ old, int64_t nnz, const int32_t* __restrict__ rem,
                                  const int32_t* __restrict__ rem_prefix, const uint64_t* __restrict__ fk,
                                  const int32_t* __restrict__ add_prefix, int64_t K, uint64_t* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < nnz; p += stride) {
    if (rem[p]) continue;
    const int64_t lb = lower_bound_u64(fk, K, old[p]);  // inserting flips before this key: add_prefix[lb] (add_prefix has K + 1 entries)
    out[p - rem_prefix[p] + add_prefix[lb]] = old[p];
  }
}
// ... and the inserted ones (rem_prefix has nnz + 1 entries)
__global__ void merge_added_kernel(const uint64_t* __restrict__ fk, const int32_t* __restrict__ add,
                                   const int32_t* __restrict__ add_prefix, int64_t K, const uint64_t* __restrict__ old, int64_t nnz,
                                   const int32_t* __restrict__ rem_prefix, uint64_t* __restrict__ out) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K || !add[k]) return;
  const int64_t lb = lower_bound_u64(old, nnz, fk[k]);
  out[lb - rem_prefix[lb] + add_prefix[k]] = fk[k];
}

int grid_for(int64_t n, int block = 256) { return int(std::min<int64_t>(cdiv(std::max<int64_t>(n, 1), block), 4096)); }

// sort `n` uint64 keys: in -> out (tmp grown on demand)
int sort_keys(uint64_t* in, uint64_t* out, int64_t n, int end_bit, DevBuf& tmp, hipStream_t s) {
  size_t bytes = 0;
  LGNN_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, in, out, size_t(n), 0, end_bit, s));
  LGNN_CALL(tmp.reserve(bytes));
  LGNN_HIP_CHECK(rocprim::radix_sort_keys(tmp.p, bytes, in, out, size_t(n), 0, end_bit, s));
  return 0;
}

}  // namespace

int exclusive_scan_i32(int32_t* in, int32_t* out, int64_t n, DevBuf& tmp, hipStream_t s) {
  size_t bytes = 0;
  LGNN_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, int32_t(0), size_t(n), rocprim::plus<int32_t>(), s));
  LGNN_CALL(tmp.reserve(bytes));
  LGNN_HIP_CHECK(rocprim::exclusive_scan(tmp.p, bytes, in, out, int32_t(0), size_t(n), rocprim::plus<int32_t>(), s));
  return 0;
}

namespace {

// keys (sorted, unique, no sentinel) -> CSR arrays
int keys_to_csr(const uint64_t* keys, int64_t nnz, int64_t N, DevBuf& rowptr, DevBuf& col, DevBuf& tmp, DevBuf& cnt,
                hipStream_t s) {
  LGNN_CALL(rowptr.reserve(size_t(N + 1) * 4));
  LGNN_CALL(col.reserve(size_t(std::max<int64_t>(nnz, 1)) * 4));
  LGNN_CALL(cnt.reserve(size_t(N + 1) * 4));
  LGNN_HIP_CHECK(hipMemsetAsync(cnt.p, 0, size_t(N + 1) * 4, s));
  if (nnz > 0)
    hipLaunchKernelGGL(split_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, keys, nnz, N, col.as<int32_t>(),
                       cnt.as<int32_t>());
  LGNN_CALL(exclusive_scan_i32(cnt.as<int32_t>(), rowptr.as<int32_t>(), N + 1, tmp, s));
  return 0;
}

}  // namespace

int compact_flags(const uint8_t* flags, int64_t n, int32_t* out, int32_t* count_dev, DevBuf& tmp, hipStream_t s) {
  rocprim::counting_iterator<int32_t> ids(0);
  size_t bytes = 0;
  LGNN_HIP_CHECK(rocprim::select(nullptr, bytes, ids, flags, out, count_dev, size_t(n), s));
  LGNN_CALL(tmp.reserve(bytes));
  LGNN_HIP_CHECK(rocprim::select(tmp.p, bytes, ids, flags, out, count_dev, size_t(n), s));
  return 0;
}


// Degree scales and the fp32 values of P / P^T for the CSRs in h->A / h->AT (`same`: A^T aliases A), and the P / P^T views.
int graph_values(lgnn_ctx* h, bool same, hipStream_t s) {
  const int64_t N = h->N, nnz = h->nnz;
  LGNN_CALL(h->deg_scale.reserve(size_t(N) * 4));
  hipLaunchKernelGGL(degree_scale_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, h->A.rowptr, N, h->kind,
                     h->deg_scale.as<float>());
  const size_t vb = size_t(std::max<int64_t>(nnz, 1)) * 4;
  const dim3 rg(cdiv(N * 64, 256));
  if (h->kind == LGNN_KIND_GCN) {
    // P = D A^T D: forward CSR = A^T rows, backward CSR = A rows, value d_i d_j for both
    LGNN_CALL(h->val_bwd.reserve(vb));
    hipLaunchKernelGGL(csr_values_kernel, rg, dim3(256), 0, s, h->A.rowptr, h->A.col, N,
                       h->deg_scale.as<float>(), 0, h->val_bwd.as<float>());
    if (same) {
      h->A.val = h->AT.val = h->val_bwd.as<float>();
    } else {
      LGNN_CALL(h->val_fwd.reserve(vb));
      hipLaunchKernelGGL(csr_values_kernel, rg, dim3(256), 0, s, h->AT.rowptr, h->AT.col, N,
                         h->deg_scale.as<float>(), 0, h->val_fwd.as<float>());
      h->A.val = h->val_bwd.as<float>();
      h->AT.val = h->val_fwd.as<float>();
    }
    h->P = h->AT;
    h->PT = h->A;
  } else {
    // P = A / rowsum: forward CSR = A rows (value 1/deg_row); backward CSR = A^T rows (value 1/deg_col)
    LGNN_CALL(h->val_fwd.reserve(vb));
    LGNN_CALL(h->val_bwd.reserve(vb));
    hipLaunchKernelGGL(csr_values_kernel, rg, dim3(256), 0, s, h->A.rowptr, h->A.col, N,
                       h->deg_scale.as<float>(), 1, h->val_fwd.as<float>());
    hipLaunchKernelGGL(csr_values_kernel, rg, dim3(256), 0, s, h->AT.rowptr, h->AT.col, N,
                       h->deg_scale.as<float>(), 2, h->val_bwd.as<float>());
    h->P = h->A;
    h->P.val = h->val_fwd.as<float>();
    h->PT = h->AT;
    h->PT.val = h->val_bwd.as<float>();
    h->A.val = h->val_fwd.as<float>();
  }
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}


// Edit the stored adjacency in place: K pairs (i, j) with the state (1 = stored, 0 = absent) they shall have afterwards.
// No re-ingest: the flips are sorted (K keys), classified against the stored keys by binary search, and the new key list is
// the MERGE of the kept entries and the inserted ones (positions from two prefix sums) -- the big list is never sorted again.
// The transpose gets the same treatment (with the transposed flips) unless it aliases the adjacency and the flips are
// mirrored.  Degree scales and the values of P / P^T are recomputed (one pass over the entries); everything cached from the
// graph (forward pass, P X, rowsum(P), long-row lists, two-hop count) is dropped, what depends on X only (X^T X) stays.
int graph_update(lgnn_ctx* h, const int64_t* fi, const int64_t* fj, const uint8_t* st, int64_t K, hipStream_t s) {
  if (K <= 0) return 0;
  const int64_t N = h->N, nnz = h->nnz;
  LGNN_REQUIRE(fi && fj && st, "null flip list");
  const bool same = h->A.rowptr == h->AT.rowptr;
  DevBuf kA, kT, fk, tk, fks, tks, sts, tsts, addA, addT, remA, remT, pre, nk, tmp, cnt, flag;
  DevBuf* all[] = {&kA, &kT, &fk, &tk, &fks, &tks, &sts, &tsts, &addA, &addT, &remA, &remT, &pre, &nk, &tmp, &cnt, &flag};
  auto cleanup = [&]() { for (DevBuf* b : all) b->release(); };
  int rc = [&]() -> int {
    const size_t kb = size_t(std::max<int64_t>(nnz, 1)) * 8;
    LGNN_CALL(kA.reserve(kb));
    LGNN_CALL(kT.reserve(kb));
    for (DevBuf* b : {&fk, &tk, &fks, &tks}) LGNN_CALL(b->reserve(size_t(K) * 8));
    for (DevBuf* b : {&sts, &tsts}) LGNN_CALL(b->reserve(size_t(K)));
    for (DevBuf* b : {&addA, &addT}) LGNN_CALL(b->reserve(size_t(K + 1) * 4));
    for (DevBuf* b : {&remA, &remT}) LGNN_CALL(b->reserve(size_t(nnz + 1) * 4));
    LGNN_CALL(flag.reserve(64));
    LGNN_HIP_CHECK(hipMemsetAsync(flag.p, 0, 64, s));
    int* d_flag = flag.as<int>();  // [0] bad id, [1] duplicate, [2] asymmetric, [4..5] counters of A, [6..7] of A^T
    const dim3 rg(cdiv(N * 64, 256)), kg(cdiv(K, 256));
    hipLaunchKernelGGL(expand_keys_kernel, rg, dim3(256), 0, s, h->A.rowptr, h->A.col, N, kA.as<uint64_t>());
    hipLaunchKernelGGL(expand_keys_kernel, rg, dim3(256), 0, s, h->AT.rowptr, h->AT.col, N, kT.as<uint64_t>());
    hipLaunchKernelGGL(flip_keys_kernel, kg, dim3(256), 0, s, fi, fj, K, N, fk.as<uint64_t>(), tk.as<uint64_t>(), d_flag);
    auto sort_pairs = [&](uint64_t* kin, uint64_t* kout, uint8_t* vout) -> int {
      size_t bytes = 0;
      LGNN_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, st, vout, size_t(K), 0, 64, s));
      LGNN_CALL(tmp.reserve(bytes));
      LGNN_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, st, vout, size_t(K), 0, 64, s));
      return 0;
    };
    LGNN_CALL(sort_pairs(fk.as<uint64_t>(), fks.as<uint64_t>(), sts.as<uint8_t>()));
    LGNN_CALL(sort_pairs(tk.as<uint64_t>(), tks.as<uint64_t>(), tsts.as<uint8_t>()));
    hipLaunchKernelGGL(flip_symmetry_kernel, kg, dim3(256), 0, s, fks.as<uint64_t>(), sts.as<uint8_t>(), K, N, d_flag + 2);
    LGNN_HIP_CHECK(hipMemsetAsync(remA.p, 0, size_t(nnz + 1) * 4, s));
    LGNN_HIP_CHECK(hipMemsetAsync(remT.p, 0, size_t(nnz + 1) * 4, s));
    hipLaunchKernelGGL(flip_classify_kernel, kg, dim3(256), 0, s, kA.as<uint64_t>(), nnz, fks.as<uint64_t>(), sts.as<uint8_t>(),
                       K, addA.as<int32_t>(), remA.as<int32_t>(), d_flag + 4, d_flag + 1);
    hipLaunchKernelGGL(flip_classify_kernel, kg, dim3(256), 0, s, kT.as<uint64_t>(), nnz, tks.as<uint64_t>(), tsts.as<uint8_t>(),
                       K, addT.as<int32_t>(), remT.as<int32_t>(), d_flag + 6, d_flag + 1);
    int hf[8] = {};
    LGNN_HIP_CHECK(hipMemcpyAsync(hf, flag.p, 32, hipMemcpyDeviceToHost, s));
    LGNN_HIP_CHECK(hipStreamSynchronize(s));  // documented: the new entry count has to reach the host
    LGNN_REQUIRE(hf[0] == 0, "flip entry out of [0, num_nodes)");
    LGNN_REQUIRE(hf[1] == 0, "a pair is listed twice in the flips");
    const bool new_same = same && hf[2] == 0;
    const int64_t nnz2 = nnz - hf[5] + hf[4];
    LGNN_REQUIRE(hf[4] - hf[5] == hf[6] - hf[7], "internal: adjacency and transpose disagree");
    LGNN_REQUIRE(nnz2 < (int64_t(1) << 31) - 64, "nnz must fit int32");
    LGNN_CALL(nk.reserve(size_t(std::max<int64_t>(nnz2, 1)) * 8));
    LGNN_CALL(pre.reserve(size_t(nnz + K + 2) * 4));
    auto merged_csr = [&](DevBuf& old, DevBuf& fkeys, DevBuf& add, DevBuf& rem, DevBuf& rowptr, DevBuf& col) -> int {
      // exclusive prefix sums (n + 1 entries each; the extra entry of the inputs is zero) -> positions
      int32_t* rem_prefix = pre.as<int32_t>();
      int32_t* add_prefix = pre.as<int32_t>() + (nnz + 1);
      LGNN_HIP_CHECK(hipMemsetAsync(add.as<int32_t>() + K, 0, 4, s));
      LGNN_CALL(exclusive_scan_i32(rem.as<int32_t>(), rem_prefix, nnz + 1, tmp, s));
      LGNN_CALL(exclusive_scan_i32(add.as<int32_t>(), add_prefix, K + 1, tmp, s));
      if (nnz > 0)
        hipLaunchKernelGGL(merge_kept_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, old.as<uint64_t>(), nnz, rem.as<int32_t>(),
                           rem_prefix, fkeys.as<uint64_t>(), add_prefix, K, nk.as<uint64_t>());
      hipLaunchKernelGGL(merge_added_kernel, kg, dim3(256), 0, s, fkeys.as<uint64_t>(), add.as<int32_t>(), add_prefix, K,
                         old.as<uint64_t>(), nnz, rem_prefix, nk.as<uint64_t>());
      LGNN_CALL(keys_to_csr(nk.as<uint64_t>(), nnz2, N, rowptr, col, tmp, cnt, s));
      return 0;
    };
    LGNN_CALL(merged_csr(kA, fks, addA, remA, h->A_rowptr, h->A_col));
    h->A.rowptr = h->A_rowptr.as<int32_t>();
    h->A.col = h->A_col.as<int32_t>();
    if (new_same) {
      h->AT = h->A;
    } else {
      LGNN_CALL(merged_csr(kT, tks, addT, remT, h->AT_rowptr, h->AT_col));
      h->AT.rowptr = h->AT_rowptr.as<int32_t>();
      h->AT.col = h->AT_col.as<int32_t>();
    }
    h->nnz = nnz2;
    h->sym = new_same;
    LGNN_CALL(graph_values(h, new_same, s));
    // everything cached from the graph
    h->fc.valid = false; h->fc.aux_valid = false; h->fc.px_valid = false;
    for (int l = 0; l < kMaxLayers; ++l) h->fc.gram_valid[l] = (l == 0 && h->kind == LGNN_KIND_GCN) ? h->fc.gram_valid[0] : false;
    h->n_long = -1; h->n_long_fwd = -1; h->n_top_multi = 0; h->n_top_slices = 0; h->n_long_tasks = 0; h->two_hop = -1.0; h->two_hop_max = -1.0;
    h->ws.planes_a_zero_ptr = nullptr;
    LGNN_HIP_CHECK(hipStreamSynchronize(s));  // temporaries are released below
    return 0;
  }();
  cleanup();
  return rc;
}

int graph_build(lgnn_ctx* h, const int64_t* ei, int64_t E, hipStream_t s) {
  const int64_t N = h->N;
  LGNN_REQUIRE(N > 0 && N < (int64_t(1) << 31) - 64, "num_nodes out of range");
  LGNN_REQUIRE(E >= 0, "negative edge count");
  const int sym = h->sym ? 1 : 0;  // requested symmetrisation (base_gnn.py:68-72)
  const int loops = h->kind == LGNN_KIND_GCN ? 1 : 0;
  const int64_t total = (sym ? 2 * E : E) + (loops ? N : 0);
  LGNN_REQUIRE(total < (int64_t(1) << 31) - 64, "nnz must fit int32");

  DevBuf keys_a, keys_b, tmp, cnt, flag;
  auto cleanup = [&]() {
    keys_a.release(); keys_b.release(); tmp.release(); cnt.release(); flag.release();
  };
  int rc = [&]() -> int {
    LGNN_CALL(keys_a.reserve(size_t(std::max<int64_t>(total, 1)) * 8));
    LGNN_CALL(keys_b.reserve(size_t(std::max<int64_t>(total, 1)) * 8 + 8));
    LGNN_CALL(flag.reserve(64));
    LGNN_HIP_CHECK(hipMemsetAsync(flag.p, 0, 64, s));
    int* d_bad = flag.as<int>();
    int* d_differ = flag.as<int>() + 1;
    size_t* d_count = reinterpret_cast<size_t*>(flag.as<char>() + 16);

    int64_t nnz = 0;
    if (total > 0) {
      hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(total)), dim3(256), 0, s, ei, E, N, sym, loops,
                         keys_a.as<uint64_t>(), d_bad);
      LGNN_CALL(sort_keys(keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), total, 64, tmp, s));
      // adjacent-unique == the clamp `adj[adj > 1] = 1`
      size_t bytes = 0;
      LGNN_HIP_CHECK(rocprim::unique(nullptr, bytes, keys_b.as<uint64_t>(), keys_a.as<uint64_t>(), d_count,
                                     size_t(total), rocprim::equal_to<uint64_t>(), s));
      LGNN_CALL(tmp.reserve(bytes));
      LGNN_HIP_CHECK(rocprim::unique(tmp.p, bytes, keys_b.as<uint64_t>(), keys_a.as<uint64_t>(), d_count,
                                     size_t(total), rocprim::equal_to<uint64_t>(), s));
      size_t h_count = 0;
      int h_flags[2] = {0, 0};
      uint64_t last = 0;
      LGNN_HIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof(size_t), hipMemcpyDeviceToHost, s));
      LGNN_HIP_CHECK(hipMemcpyAsync(h_flags, flag.p, 8, hipMemcpyDeviceToHost, s));
      LGNN_HIP_CHECK(hipStreamSynchronize(s));  // documented: nnz has to reach the host
      LGNN_REQUIRE(h_flags[0] == 0, "edge_index entry out of [0, num_nodes)");
      nnz = int64_t(h_count);
      if (nnz > 0) {
        LGNN_HIP_CHECK(hipMemcpyAsync(&last, keys_a.as<uint64_t>() + (nnz - 1), 8, hipMemcpyDeviceToHost, s));
        LGNN_HIP_CHECK(hipStreamSynchronize(s));
        if (last == kSentinel) --nnz;  // dropped (diagonal / invalid) entries sort last
      }
    }
    h->nnz = nnz;
    // keys_a[0:nnz) = sorted unique keys of A
    LGNN_CALL(keys_to_csr(keys_a.as<uint64_t>(), nnz, N, h->A_rowptr, h->A_col, tmp, cnt, s));
    h->A.rowptr = h->A_rowptr.as<int32_t>();
    h->A.col = h->A_col.as<int32_t>();

    // transpose
    bool same = true;
    if (nnz > 0) {
      hipLaunchKernelGGL(transpose_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, keys_a.as<uint64_t>(), nnz, N,
                         keys_b.as<uint64_t>());
      DevBuf keys_c;
      int rc2 = [&]() -> int {
        LGNN_CALL(keys_c.reserve(size_t(nnz) * 8));
        LGNN_CALL(sort_keys(keys_b.as<uint64_t>(), keys_c.as<uint64_t>(), nnz, 64, tmp, s));
        hipLaunchKernelGGL(compare_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, keys_a.as<uint64_t>(),
                           keys_c.as<uint64_t>(), nnz, d_differ);
        int differ = 0;
        LGNN_HIP_CHECK(hipMemcpyAsync(&differ, d_differ, 4, hipMemcpyDeviceToHost, s));
        LGNN_HIP_CHECK(hipStreamSynchronize(s));
        same = differ == 0;
        if (!same) {
          LGNN_CALL(keys_to_csr(keys_c.as<uint64_t>(), nnz, N, h->AT_rowptr, h->AT_col, tmp, cnt, s));
          LGNN_HIP_CHECK(hipStreamSynchronize(s));
        }
        return 0;
      }();
      keys_c.release();
      if (rc2) return rc2;
    }
    h->sym = same;
    if (same) {
      h->AT = h->A;
    } else {
      h->AT.rowptr = h->AT_rowptr.as<int32_t>();
      h->AT.col = h->AT_col.as<int32_t>();
    }

    LGNN_CALL(graph_values(h, same, s));
    LGNN_HIP_CHECK(hipStreamSynchronize(s));  // temporaries are released below
    LGNN_HIP_CHECK(hipGetLastError());
    return 0;
  }();
  cleanup();
  return rc;
}

}  // namespace lgnn

using namespace lgnn;

extern "C" int lgnn_update_adjacency(lgnn_ctx* h, const int64_t* rows, const int64_t* cols, const uint8_t* state,
                                     int64_t num_flips, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  LGNN_REQUIRE(num_flips >= 0, "negative flip count");
  return graph_update(h, rows, cols, state, num_flips, static_cast<hipStream_t>(stream));
}

extern "C" int lgnn_export_adj(const lgnn_ctx* h, int64_t* rows, int64_t* cols, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(export_coo_kernel, dim3(cdiv(h->N * 64, 256)), dim3(256), 0, s, h->A.rowptr, h->A.col,
                     (const float*)nullptr, h->N, rows, cols, (float*)nullptr);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int lgnn_export_propagation(const lgnn_ctx* h, int64_t* rows, int64_t* cols, float* vals, void* stream) {
  if (!h) { set_error("null context"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(export_coo_kernel, dim3(cdiv(h->N * 64, 256)), dim3(256), 0, s, h->P.rowptr, h->P.col, h->P.val,
                     h->N, rows, cols, vals);
  LGNN_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int lgnn_adj_to_edge_index(const lgnn_ctx* h, int64_t* out, int64_t* num_out, void* stream) {
  if (!h || !num_out) { set_error("null argument"); return 2; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t N = h->N;
  DevBuf cnt, off, tmp;
  int rc = [&]() -> int {
    LGNN_CALL(cnt.reserve(size_t(N + 1) * 4));
    LGNN_CALL(off.reserve(size_t(N + 1) * 4));
    LGNN_HIP_CHECK(hipMemsetAsync(cnt.p, 0, size_t(N + 1) * 4, s));
    hipLaunchKernelGGL(offdiag_count_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, h->A.rowptr, h->A.col, N,
                       cnt.as<int32_t>());
    LGNN_CALL(exclusive_scan_i32(cnt.as<int32_t>(), off.as<int32_t>(), N + 1, tmp, s));
    int32_t total = 0;
    LGNN_HIP_CHECK(hipMemcpyAsync(&total, off.as<int32_t>() + N, 4, hipMemcpyDeviceToHost, s));
    LGNN_HIP_CHECK(hipStreamSynchronize(s));
    *num_out = total;
    if (out && total > 0) {
      hipLaunchKernelGGL(offdiag_write_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, h->A.rowptr, h->A.col, N,
                         off.as<int32_t>(), int64_t(total), out);
      LGNN_HIP_CHECK(hipStreamSynchronize(s));
    }
    return 0;
  }();
  cnt.release(); off.release(); tmp.release();
  return rc;
}
