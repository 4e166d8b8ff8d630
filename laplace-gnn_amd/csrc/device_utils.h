// Small device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/laplace_gnn_hip.h"

namespace lgnn {

__device__ __forceinline__ float act_apply(float x, int act) {
  return act == LGNN_ACT_RELU ? fmaxf(x, 0.f) : tanhf(x);
}
// derivative of the activation expressed through its OUTPUT h (relu: h > 0 <=> pre > 0)
__device__ __forceinline__ float act_deriv_from_out(float h, int act) {
  return act == LGNN_ACT_RELU ? (h > 0.f ? 1.f : 0.f) : (1.f - h * h);
}

}  // namespace lgnn
