/* TEST INFRASTRUCTURE ONLY (the oracle's side of bench.py's cpu_baseline leg): out = A @ X for a CSR matrix A (fp32 values,
 * int32 indices) and a row-major dense X [n_cols, w], one output row per OpenMP task -- the same row-sequential fp32 sums as
 * scipy's csr_matvecs, which runs on ONE core (BASELINE.md section 3 plans the baseline on all host cores).
 * Built by oracle/Makefile into oracle/_build/libspmm_omp.so; loaded with ctypes by gnn_laplace_oracle.ThreadedCsr. */
#include <stdint.h>
#include <string.h>

void csr_spmm_f32(int64_t nrows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t w,
                  float* out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < nrows; ++r) {
    float* o = out + r * w;
    memset(o, 0, (size_t)w * sizeof(float));
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      const float v = val[p];
      const float* x = X + (int64_t)col[p] * w;
      for (int64_t j = 0; j < w; ++j) o[j] += v * x[j];
    }
  }
}
