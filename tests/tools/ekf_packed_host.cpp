// ekf_packed_host.cpp — host build of the engine's per-vehicle EKF arithmetic (csrc/ekf_math.h), so
// the packed fast step of the fused HIP kernel can be checked against the oracle without a GPU.
// Built by tests/test_ekf_packed_host.py: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC.
// Mirrors the kernel's control flow per step: packed step; if the lane left the fast domain
// (|yaw| >= 120, non-finite, extreme determinant) the step is redone with the general code.
// (On the host recip_fast is the IEEE quotient itself; the un-scaled Newton sequence is device-only
// and is covered by the GPU parity tests.)
#include <cstring>
#include "../../cpprobotics_amd/csrc/ekf_math.h"

extern "C" int ekf_packed_run(int n, int T, float* x, float* P, const float* z, const float* u,
                              float* x_hist, const float* Q, const float* R, double dt,
                              long long* n_slow) {
  crx::EkfConsts k;
  std::memcpy(k.Q, Q, sizeof(k.Q));
  std::memcpy(k.R, R, sizeof(k.R));
  k.dt = dt;
  const crx::EkfConstsP kp = crx::pack_consts(k);
  long long slow = 0;
  for (int a = 0; a < n; ++a) {
    crx::EkfState s;
    s.x0 = x[4 * a + 0]; s.x1 = x[4 * a + 1]; s.x2 = x[4 * a + 2]; s.x3 = x[4 * a + 3];
    std::memcpy(s.P, P + 16 * (size_t)a, sizeof(s.P));
    crx::EkfStateP sp;
    crx::pack_state(sp, s);
    for (int t = 0; t < T; ++t) {
      const size_t o = (size_t)t * n + a;
      const crx::v2f zc = {z[2 * o], z[2 * o + 1]}, uc = {u[2 * o], u[2 * o + 1]};
      const crx::EkfStateP s_in = sp;
      crx::FastDomain dom = crx::fast_domain_init();
      if (crx::dt_split_is_exact(dt)) crx::ekf_step_packed<true>(sp, zc, uc, kp, dom);       // as the host side of the engine picks it
      else crx::ekf_step_packed<false>(sp, zc, uc, kp, dom);
      if (!crx::fast_domain_ok(dom)) {
        ++slow;
        crx::unpack_state(s, s_in);
        crx::ekf_step_dev(s, zc[0], zc[1], uc[0], uc[1], k);
        crx::pack_state(sp, s);
      }
      if (x_hist) {
        x_hist[4 * o + 0] = sp.x01[0]; x_hist[4 * o + 1] = sp.x01[1];
        x_hist[4 * o + 2] = sp.x23[0]; x_hist[4 * o + 3] = sp.x23[1];
      }
    }
    crx::unpack_state(s, sp);
    x[4 * a + 0] = s.x0; x[4 * a + 1] = s.x1; x[4 * a + 2] = s.x2; x[4 * a + 3] = s.x3;
    std::memcpy(P + 16 * (size_t)a, s.P, sizeof(s.P));
  }
  if (n_slow) *n_slow = slow;
  return 0;
}
