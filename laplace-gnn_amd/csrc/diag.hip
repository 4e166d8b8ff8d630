// Diagonal GGN and last-layer full GGN of one mini-batch (placeholder until the kernels land).
#include "lgnn_internal.h"

namespace lgnn {

int diag_accumulate(lgnn_ctx*, const int64_t*, const void*, int64_t, uint32_t, float*, float*, hipStream_t) {
  set_error("lgnn_diag_accumulate: not implemented yet");
  return 3;
}

int lastlayer_full_accumulate(lgnn_ctx*, const int64_t*, const void*, int64_t, float*, float*, hipStream_t) {
  set_error("lgnn_lastlayer_full_accumulate: not implemented yet");
  return 3;
}

}  // namespace lgnn
