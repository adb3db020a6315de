// Drives the C++ host mirror (include/dddmr_rollout.hpp) through the reference's
// only fixed-input scenario, the local-planner playground
// (local_planner_play_ground_node.cpp:206-298, local_planner_play_ground.yaml:63-125),
// and prints the chosen command; tests/test_cpp_mirror_gpu.py checks it.
#include <cstdio>
#include <cstring>
#include <vector>

#include "dddmr_rollout.hpp"

static dddmr_theory_config playground_theory() {
  dddmr_theory_config t;
  std::memset(&t, 0, sizeof(t));
  std::strcpy(t.name, "differential_drive_simple");
  t.kind = DDDMR_THEORY_DD_SIMPLE;
  t.max_vel_x = 1.0; t.min_vel_x = 0.1; t.max_vel_theta = 0.6; t.min_vel_theta = 0.15;
  t.acc_lim_x = 1.0; t.acc_lim_theta = 3.0; t.deceleration_ratio = 2.0;
  t.max_motor_shaft_rpm = 3000.0; t.wheel_diameter = 0.16; t.gear_ratio = 1.0; t.robot_radius = 0.25;
  t.controller_frequency = 10.0; t.sim_time = 5.0; t.linear_x_sample = 5.0; t.angular_z_sample = 10.0;
  t.sim_granularity = 0.05; t.angular_sim_granularity = 0.025;
  t.min_vel_y = -0.1; t.max_vel_y = 0.1; t.max_vel_trans = 0.1; t.acc_lim_y = 0.3; t.linear_y_sample = 10.0;
  t.rotation_speed = 0.4;
  // push order blb, brb, blt, flb, brt, frt, flt, frb (dd_simple...cpp:211-218)
  const float c[8][3] = {{-0.35f, 0.36f, 0.f}, {-0.35f, -0.36f, 0.f}, {-0.35f, 0.36f, 0.6f}, {0.42f, 0.36f, 0.f},
                         {-0.35f, -0.36f, 0.6f}, {0.42f, -0.36f, 0.6f}, {0.42f, 0.36f, 0.6f}, {0.42f, -0.36f, 0.f}};
  std::memcpy(t.cuboid, c, sizeof(c));
  t.n_critics = 4;
  t.critics[0].kind = DDDMR_CRITIC_COLLISION; t.critics[0].weight = 1.0;
  t.critics[1].kind = DDDMR_CRITIC_STICK_PATH; t.critics[1].weight = 0.1;
  t.critics[2].kind = DDDMR_CRITIC_PURE_PURSUIT; t.critics[2].translation_weight = 1.0; t.critics[2].orientation_weight = 0.01;
  t.critics[3].kind = DDDMR_CRITIC_TOWARD_GLOBAL_PLAN; t.critics[3].weight = 1.0;
  return t;
}

int main(int argc, char** argv) {
  const double gx = argc > 2 ? atof(argv[1]) : 3.0, gy = argc > 2 ? atof(argv[2]) : 1.0;
  try {
    dddmr_amd::LocalPlanner lp({playground_theory()}, 0, 1024, 4096, 256, 64);
    // 5 obstacle points near (0.8, 0.6, 0.2), as PCL PointXYZI records (stride 32)
    float cloud[5][8] = {{0.80f, 0.60f, 0.2f}, {0.75f, 0.65f, 0.2f}, {0.85f, 0.55f, 0.2f}, {0.70f, 0.70f, 0.2f}, {0.90f, 0.50f, 0.2f}};
    lp.setCloud(&cloud[0][0], 5, 32);
    std::vector<double> plan(20 * 7, 0.0);
    for (int i = 0; i < 20; ++i) { plan[7 * i] = gx / 20 * i; plan[7 * i + 1] = gy / 20 * i; plan[7 * i + 6] = 1.0; }
    lp.setPlan(plan.data(), 20);
    dddmr_tick_input in;
    std::memset(&in, 0, sizeof(in));
    in.robot_pose[6] = 1.0;
    in.robot_twist[0] = 0.4;
    in.allowed_max_linear_speed = -1.0;
    dddmr_amd::Trajectory best;
    const dddmr_amd::PlannerState st = lp.computeVelocityCommand("differential_drive_simple", best, in);
    const auto poses = lp.bestPoses();
    std::printf("%d %d %.17g %.17g %.17g %.17g %u %zu\n", (int)st, best.index_, best.cost_, best.xv_, best.yv_, best.thetav_,
                lp.lastResult().n_samples, poses.size());
    // PathBlockedStrategy: 6 observation points (one more than the "no obstacle" rule) next to the
    // third of four plan points, the first one tagged backward
    float cloud6[6][8] = {{1.0f, 0.05f, 0.f}, {1.0f, 0.05f, 0.f}, {1.0f, 0.05f, 0.f}, {1.0f, 0.05f, 0.f}, {1.0f, 0.05f, 0.f}, {1.0f, 0.05f, 0.f}};
    lp.setCloud(&cloud6[0][0], 6, 32);
    const float pcl_prune_plan[4][4] = {{0.f, 0.f, 0.f, -1.f}, {0.5f, 0.f, 0.f, 1.f}, {1.0f, 0.f, 0.f, 1.f}, {1.5f, 0.f, 0.f, 1.f}};
    dddmr_perception_opinion op = DDDMR_OPINION_PASS;
    const double ratio = lp.pathBlockedRatio(&pcl_prune_plan[0][0], 4, 0.2, &op);
    std::printf("blocked %.17g %d\n", ratio, (int)op);
    dddmr_amd::Trajectory none;
    try {
      lp.computeVelocityCommand("no_such_theory", none, in);
      std::printf("missing-error\n");
      return 2;
    } catch (const dddmr_amd::RolloutError& e) {
      std::printf("error %d\n", e.code);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  return 0;
}
