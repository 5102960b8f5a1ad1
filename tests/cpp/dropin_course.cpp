// dropin_course.cpp — the set-up functions of the reference's tracking mains through include/crx_dropin.hpp (host-only: these
// helpers need no GPU): calc_speed_profile of the three files and the MPC file's smooth_yaw, printed as hex floats.
#define CRX_DROPIN_NO_EIGEN 1
#include "crx_dropin.hpp"
#include <cstdio>
#include <cmath>

static void dump(const char* tag, const crx_dropin::Vec_f& v) {
  std::printf("%s", tag);
  for (float x : v) std::printf(" %a", x);
  std::printf("\n");
}

int main() {
  using namespace crx_dropin;
  Vec_f rx, ry, ryaw;
  for (int i = 0; i < 90; ++i) {
    const float t = 0.07f * i;
    rx.push_back(20.0f * std::cos(t)); ry.push_back(15.0f * std::sin(1.3f * t));
    ryaw.push_back(std::atan2(15.0f * 1.3f * std::cos(1.3f * t), -20.0f * std::sin(t)));   // jumps by 2 pi where the heading wraps
  }
  dump("yaw_in", ryaw);
  dump("sp5", lqr_speed_steer::calc_speed_profile(rx, ry, ryaw, 2.7777777f));
  dump("sp4", lqr_steer::calc_speed_profile(rx, ry, ryaw, 2.7777777f));
  dump("sp0", mpc::calc_speed_profile(rx, ry, ryaw, 2.7777777f));
  dump("x", rx); dump("y", ry);
  mpc::smooth_yaw(ryaw);
  dump("yaw_out", ryaw);
  return 0;
}
