// ref_dwa.cpp — TEST INFRASTRUCTURE.  The reference's dynamic-window planner compiled from its own lines
// (/root/reference/src/dynamic_window_approach.cpp:16-41 types and Config, :43-155 motion … dwa_control; the main loop
// :192-194 and goal test :221 are restated in the wrapper because main() interleaves them with drawing).
#include <array>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <vector>

namespace ref_dwa {
#include "dwa_types.inc"
#include "dwa_fns.inc"
}

extern "C" {

// cfg: the 12 floats of Config in declaration order.  One dwa_control + motion per agent; returns u and the moved state.
void ref_dwa_run(int n, int max_ticks, float* state, float* u, const float* goal, const float* ob, int nob, const float* cfg, float* traj,
                 int* ticks_done) {
  using namespace ref_dwa;
  Config config;
  std::memcpy(&config, cfg, sizeof(float) * 12);
  Obstacle obs;
  for (int k = 0; k < nob; ++k) obs.push_back({{ob[2 * k], ob[2 * k + 1]}});
  for (int a = 0; a < n; ++a) {
    State x; std::memcpy(x.data(), state + 5 * (size_t)a, 20);
    Control uu{{u[2 * a], u[2 * a + 1]}};
    Point g{{goal[2 * a], goal[2 * a + 1]}};
    int ticks = 0;
    for (int i = 0; i < max_ticks; ++i) {                                  // :192 `for(int i=0; i<1000 && !terminal; i++)`
      Traj ltraj = dwa_control(x, uu, config, g, obs);                     // :193
      x = motion(x, uu, config.dt);                                        // :194
      ticks = i + 1;
      if (traj) std::memcpy(traj + ((size_t)i * n + a) * 5, x.data(), 20);
      if (std::sqrt(std::pow((x[0] - g[0]), 2) + std::pow((x[1] - g[1]), 2)) <= config.robot_radius) break;   // :221
    }
    std::memcpy(state + 5 * (size_t)a, x.data(), 20);
    u[2 * a] = uu[0]; u[2 * a + 1] = uu[1];
    ticks_done[a] = ticks;
  }
}

int ref_dwa_config_floats(float* out) {      // the defaults of class Config (:25-41), in declaration order
  ref_dwa::Config c;
  std::memcpy(out, &c, sizeof(float) * 12);
  return (int)(sizeof(c) / sizeof(float));
}

}  // extern "C"
