// dwa_ref.cpp — TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).
// Nothing under cpprobotics_amd/ may include, link or call this file.
//
// CPU restatement of the reference's dynamic-window planner, /root/reference/src/dynamic_window_approach.cpp:
//   Config :25-41, motion :43-50, calc_dynamic_window :52-60, calc_trajectory :63-74, calc_obstacle_cost :77-101,
//   calc_to_goal_cost :103-113, calc_final_input :115-145, dwa_control :148-155, main loop :192-194 + goal test :221,
// statement by statement with the host libm (cosf, sinf, acosf, sqrtf, pow, sqrt).  
// PINNED against the reference's own lines (oracle/ref_build.sh compiles them unmodified — this file needs no Eigen — and tests/test_oracle_vs_ref.py demands equal bits).
#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#define CRX_TRIG_FMA 1
#include "../cpprobotics_amd/csrc/crx_trig.h"

namespace {

int g_trig = 0;
inline float o_cos(float x) { return g_trig == 0 ? std::cos(x) : crx::cosf_(x); }
inline float o_sin(float x) { return g_trig == 0 ? std::sin(x) : crx::sinf_(x); }

using State = std::array<float, 5>;
using Control = std::array<float, 2>;

struct Config {   // :25-41 (values supplied by the caller; same field order)
  float max_speed, min_speed, max_yawrate, max_accel, robot_radius, max_dyawrate, v_reso, yawrate_reso, dt, predict_time,
      to_goal_cost_gain, speed_cost_gain;
};

State motion(State x, Control u, float dt) {   // :43-50
  x[2] += u[1] * dt;
  x[0] += u[0] * o_cos(x[2]) * dt;
  x[1] += u[0] * o_sin(x[2]) * dt;
  x[3] = u[0];
  x[4] = u[1];
  return x;
}

// One dwa_control (:148-155).  Returns the number of sampled trajectories; best_idx = index (in loop order) of the winner or -1.
int dwa_control(State x, Control& u, const Config& c, const float* goal, const float* ob, int nob, std::vector<State>* best_traj,
                int* best_idx) {
  const float dw0 = std::max((x[3] - c.max_accel * c.dt), c.min_speed);           // :55-58
  const float dw1 = std::min((x[3] + c.max_accel * c.dt), c.max_speed);
  const float dw2 = std::max((x[4] - c.max_dyawrate * c.dt), -c.max_yawrate);
  const float dw3 = std::min((x[4] + c.max_dyawrate * c.dt), c.max_yawrate);
  float min_cost = 10000.0;                                                         // :120
  Control min_u = u;
  min_u[0] = 0.0;
  int idx = 0, best = -1;
  std::vector<State> traj;
  for (float v = dw0; v <= dw1; v += c.v_reso) {                                    // :126
    for (float y = dw2; y <= dw3; y += c.yawrate_reso) {                            // :127
      traj.clear();
      State s = x;
      traj.push_back(s);
      float time = 0.0;
      while (time <= c.predict_time) {                                              // :68-72
        s = motion(s, Control{{v, y}}, c.dt);
        traj.push_back(s);
        time += c.dt;
      }
      // calc_to_goal_cost :103-113
      float goal_magnitude = std::sqrt(goal[0] * goal[0] + goal[1] * goal[1]);
      float traj_magnitude = std::sqrt(std::pow(traj.back()[0], 2) + std::pow(traj.back()[1], 2));
      float dot_product = (goal[0] * traj.back()[0]) + (goal[1] * traj.back()[1]);
      float error = dot_product / (goal_magnitude * traj_magnitude);
      float error_angle = std::acos(error);
      float to_goal_cost = c.to_goal_cost_gain * error_angle;
      float speed_cost = c.speed_cost_gain * (c.max_speed - traj.back()[3]);         // :132
      // calc_obstacle_cost :77-101
      float ob_cost;
      {
        int skip_n = 2;
        float minr = FLT_MAX;
        bool hit = false;
        for (unsigned ii = 0; ii < traj.size() && !hit; ii += skip_n) {
          for (int i = 0; i < nob; i++) {
            float dx = traj[ii][0] - ob[2 * i];
            float dy = traj[ii][1] - ob[2 * i + 1];
            float r = std::sqrt(dx * dx + dy * dy);
            if (r <= c.robot_radius) { hit = true; break; }
            if (minr >= r) minr = r;
          }
        }
        ob_cost = hit ? FLT_MAX : (float)(1.0 / minr);
      }
      float final_cost = to_goal_cost + speed_cost + ob_cost;                        // :134
      if (min_cost >= final_cost) {                                                   // :136
        min_cost = final_cost;
        min_u = Control{{v, y}};
        best = idx;
        if (best_traj) *best_traj = traj;
      }
      ++idx;
    }
  }
  u = min_u;
  if (best_idx) *best_idx = best;
  return idx;
}

}  // namespace

extern "C" {

void oracle_dwa_set_trig_mode(int m) { g_trig = m; }

// state [n][5] = (x, y, yaw, v, yawrate), u [n][2] in/out, goal [n][2], ob [nob][2] shared, cfg[12] floats (Config order).
// n_samples / best_idx (may be NULL) per agent.
void oracle_dwa_control(int n, const float* state, float* u, const float* goal, const float* ob, int nob, const float* cfg,
                        int* n_samples, int* best_idx, int a0, int a1) {
  Config c; std::memcpy(&c, cfg, sizeof(c));
  for (int a = a0; a < a1; ++a) {
    State x; std::memcpy(x.data(), state + 5 * (size_t)a, 20);
    Control uu{{u[2 * a], u[2 * a + 1]}};
    int bi;
    int ns = dwa_control(x, uu, c, goal + 2 * (size_t)a, ob, nob, nullptr, &bi);
    u[2 * a] = uu[0]; u[2 * a + 1] = uu[1];
    if (n_samples) n_samples[a] = ns;
    if (best_idx) best_idx[a] = bi;
  }
}

// main loop :192-194 + goal test :221, max_ticks iterations at most: dwa_control -> motion -> goal test.
void oracle_dwa_run(int n, int max_ticks, float* state, float* u, const float* goal, const float* ob, int nob, const float* cfg,
                    float* traj_hist, int* ticks_done, int a0, int a1) {
  Config c; std::memcpy(&c, cfg, sizeof(c));
  for (int a = a0; a < a1; ++a) {
    State x; std::memcpy(x.data(), state + 5 * (size_t)a, 20);
    Control uu{{u[2 * a], u[2 * a + 1]}};
    const float* g = goal + 2 * (size_t)a;
    int ticks = 0;
    for (int i = 0; i < max_ticks; ++i) {
      dwa_control(x, uu, c, g, ob, nob, nullptr, nullptr);
      x = motion(x, uu, c.dt);
      ticks = i + 1;
      if (traj_hist) std::memcpy(traj_hist + ((size_t)i * n + a) * 5, x.data(), 20);
      if (std::sqrt(std::pow((x[0] - g[0]), 2) + std::pow((x[1] - g[1]), 2)) <= c.robot_radius) break;
    }
    std::memcpy(state + 5 * (size_t)a, x.data(), 20);
    u[2 * a] = uu[0]; u[2 * a + 1] = uu[1];
    if (ticks_done) ticks_done[a] = ticks;
  }
}

}  // extern "C"
