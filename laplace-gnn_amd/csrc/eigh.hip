// Batched symmetric eigendecomposition of the Kronecker factors (the step right after the accumulation inside
// KronLaplace.fit: laplace/baselaplace.py:1610 -> laplace/utils/matrix.py:118-145 -> utils.py:193-226).
//
// The reference calls torch.linalg.eigh once per factor; on this stack that is one rocSOLVER syevd per matrix, each a
// chain of ~100 single-workgroup kernels (tridiagonalisation panels, divide and conquer): ~3.3 ms per 256 x 256 factor,
// 13 ms for the four distinct factors of the arxiv-shaped model -- strictly serial and using one CU.  The factors are
// independent, so they go through ONE strided-batched syevd: every kernel of the chain then carries all matrices in its
// grid and the whole decomposition costs what the largest factor costs.  (The caller pads smaller factors to the
// common size with a decoupled negative diagonal block, see matrix.py.)
#include <rocsolver/rocsolver.h>

#include "lgnn_internal.h"

namespace lgnn {
namespace {
rocblas_handle g_handle = nullptr;
DevBuf g_e;  // off-diagonal workspace [batch, n]
}  // namespace
}  // namespace lgnn

namespace lgnn {
// the process-wide rocBLAS handle, bound to stream `s` (also used by jacobian.hip); nullptr on failure
void* blas_handle(hipStream_t s) {
  if (!g_handle && rocblas_create_handle(&g_handle) != rocblas_status_success) return nullptr;
  if (rocblas_set_stream(g_handle, s) != rocblas_status_success) return nullptr;
  return g_handle;
}
}  // namespace lgnn

using namespace lgnn;

// A: [batch][n][n] fp32, symmetric, overwritten: row j of matrix b = eigenvector j (unit norm) for eigenvalue W[b][j],
// eigenvalues ascending (column-major eigenvector columns of the solver == rows of the row-major view).
// info: device int32 [batch], 0 = converged.  Asynchronous on `stream` (apart from workspace growth on first use).
extern "C" int lgnn_symeig_batched(float* A, int64_t n, int64_t batch, float* W, int32_t* info, void* stream) {
  if (!A || !W || !info) { set_error("null argument"); return 2; }
  LGNN_REQUIRE(n > 0 && n <= 32768 && batch > 0 && batch <= 65535, "symeig: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!blas_handle(s)) { set_error("rocBLAS handle / stream setup failed"); return 3; }
  LGNN_CALL(g_e.reserve(size_t(batch) * n * 4));
  const rocblas_status st = rocsolver_ssyevd_strided_batched(
      g_handle, rocblas_evect_original, rocblas_fill_upper, rocblas_int(n), A, rocblas_int(n), rocblas_stride(n * n), W,
      rocblas_stride(n), g_e.as<float>(), rocblas_stride(n), info, rocblas_int(batch));
  if (st != rocblas_status_success) { set_error("rocsolver_ssyevd_strided_batched failed"); return 3; }
  return 0;
}
