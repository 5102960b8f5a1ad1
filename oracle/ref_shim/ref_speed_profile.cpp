// ref_speed_profile.cpp — TEST INFRASTRUCTURE.  The three calc_speed_profile functions of the reference compiled from their own
// lines (/root/reference/src/lqr_speed_steer_control.cpp:40-63, src/lqr_steer_control.cpp:35-52,
// src/model_predictive_control.cpp:83-105).  Two of them write outside their vector — speed_profile[size() - 0] in the
// slow-down loop of the 5-state file (:55-56, k = 0) and speed_profile[-1] in the MPC file (:102) — undefined behaviour that a
// std::vector would turn into heap corruption here, so in THIS unit `Vec_f` is a vector whose operator[] sends an
// out-of-range index to a scratch cell (what the stray writes hit is not observable in the reference either).
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

namespace guarded {
struct Vec_f {
  std::vector<float> v;
  float scratch = 0.0f;
  Vec_f() {}
  Vec_f(std::size_t n, float x) : v(n, x) {}
  Vec_f(const float* a, const float* b) : v(a, b) {}
  std::size_t size() const { return v.size(); }
  float& operator[](long i) { return (i >= 0 && (std::size_t)i < v.size()) ? v[(std::size_t)i] : scratch; }
  float& operator[](std::size_t i) { return i < v.size() ? v[i] : scratch; }
  float& operator[](int i) { return (*this)[(long)i]; }
  float& operator[](unsigned int i) { return (*this)[(std::size_t)i]; }
  struct It {                                   // *(speed_profile.end() - k), :55-58
    Vec_f* o; long i;
    It operator-(int k) const { return It{o, i - k}; }
    float& operator*() const { return (*o)[i]; }
  };
  It end() { return It{this, (long)v.size()}; }
};
}  // namespace guarded

#define YAW_P2P(angle) std::fmod(std::fmod((angle)+M_PI, 2*M_PI)-2*M_PI, 2*M_PI)+M_PI   /* include/motion_model.h:18 */

namespace sp_lqr5 { using guarded::Vec_f;
#include "lqr5_speed_profile.inc"
}
namespace sp_lqr4 { using guarded::Vec_f;
#include "lqr4_speed_profile.inc"
}
namespace sp_mpc { using guarded::Vec_f;
#include "mpc_speed_profile.inc"
}

extern "C" void ref_calc_speed_profile(int which /* 5, 4, 0 = MPC */, int n, const float* rx, const float* ry, const float* ryaw,
                                       float target_speed, float* out) {
  guarded::Vec_f x(rx, rx + n), y(ry, ry + n), yaw(ryaw, ryaw + n), sp;
  if (which == 5) sp = sp_lqr5::calc_speed_profile(x, y, yaw, target_speed);
  else if (which == 4) sp = sp_lqr4::calc_speed_profile(x, y, yaw, target_speed);
  else sp = sp_mpc::calc_speed_profile(x, y, yaw, target_speed);
  std::memcpy(out, sp.v.data(), sizeof(float) * (std::size_t)n);
}
