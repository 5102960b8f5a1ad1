/* crx_experimental.h — measurement-only entry points of libcrx.so (prefix crx_x_).
 *
 * NOT part of the drop-in boundary (include/crx.h): these force a kernel variant or a launch geometry that the product entry
 * points choose by themselves, so that scripts/ and bench.py can A/B them on the same inputs.  Nothing here has a user in
 * cpprobotics_amd/ outside `cpprobotics_amd/experimental.py`; results are the same bits as the product path's. */
#ifndef CRX_EXPERIMENTAL_H
#define CRX_EXPERIMENTAL_H
#include "crx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* crx_dare_from_v_batch_dev with the register layout forced: lanes_per_agent = 1 (one agent per lane), 4 (one agent per DPP
 * quad) or 0 (what the product entry point would pick for this n). */
int crx_x_dare_from_v_lanes_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                int* iters, void* stream, int lanes_per_agent);

#ifdef __cplusplus
}
#endif
#endif
