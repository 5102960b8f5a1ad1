// placeholder — replaced by the real solver below in this round
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/crx.h"
#define CRX_MPC_MAX_T 64
namespace crx {
inline hipError_t mpc_launch(int, int, const float*, const float*, const crx_mpc_params&, float*, int*, double*, hipStream_t) {
  return hipErrorNotSupported;
}
}
