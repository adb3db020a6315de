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
    // In-library RCCL exchange from plain C++ (no torch, no gloo): a 1-rank communicator; the tick then runs
    // k_score -> ncclAllReduce(min) -> k_resolve and must return the same winner.
    lp.setCloud(&cloud[0][0], 5, 32);
    lp.commInit(dddmr_amd::LocalPlanner::commUniqueId(), 0, 1);
    dddmr_amd::Trajectory via_comm;
    lp.computeVelocityCommand("differential_drive_simple", via_comm, in);
    const auto words = lp.winnerWords();
    const auto resolved = lp.resolveWords({words[0], words[1]});
    std::printf("comm %d %.17g %d %.17g %zu\n", via_comm.index_, via_comm.cost_, resolved.best_index, resolved.best_cost,
                lp.samples("differential_drive_simple", in).size());
    lp.commDestroy();
    // Global-mode marking / clearing layer from plain C++: a 6 x 6 m ground lattice, one compact obstacle at (0, 2)
    // and one at (-3, 3); mark both, then let the first one vanish and clear it.
    std::vector<float> ground, obs;
    for (int i = 0; i <= 40; ++i)
      for (int j = 0; j <= 40; ++j) { ground.push_back(-5.f + 0.25f * i); ground.push_back(-5.f + 0.25f * j); ground.push_back(0.f); }
    auto blob = [&](float cx, float cy) {
      for (int z = 2; z < 10; ++z)
        for (int a = -1; a <= 1; a += 2)
          for (int b = -1; b <= 1; b += 2) { obs.push_back(cx + 0.03f * a); obs.push_back(cy + 0.03f * b); obs.push_back(0.1f * z); obs.push_back(0.f); }
    };
    dddmr_marking_config mc;
    std::memset(&mc, 0, sizeof(mc));
    mc.xy_resolution = 0.05; mc.height_resolution = 0.05; mc.marking_height = 2.0; mc.perception_window_size = 5.0;
    mc.vertical_FOV_top = 15.0; mc.vertical_FOV_bottom = -15.0;
    mc.scan_effective_positive_start = 30.0; mc.scan_effective_positive_end = 180.0;
    mc.scan_effective_negative_start = -30.0; mc.scan_effective_negative_end = -180.0;
    mc.euclidean_cluster_extraction_tolerance = 0.25; mc.euclidean_cluster_extraction_min_cluster_size = 1;
    mc.segmentation_ignore_ratio = 1.1; mc.inscribed_radius = 0.5; mc.inflation_radius = 1.5; mc.max_obstacle_distance = 9999.0;
    mc.max_markings = 1024; mc.max_cluster_points = 1 << 16;
    lp.markingCreate(mc, ground.data(), ground.size() / 3, 12, nullptr, 0, 12);
    const double t_bs[7] = {0, 0, 0.5, 0, 0, 0, 1}, t_gb[7] = {0, 0, 0, 0, 0, 0, 1};
    blob(0.f, 2.f); blob(-3.f, 3.f);
    lp.setCloud(obs.data(), obs.size() / 4, 16);
    const dddmr_marking_stats m1 = lp.markingUpdate(t_bs, t_gb);
    size_t touched = 0, lethal = 0;
    for (double d : lp.dGraph()) touched += d < 9999.0 ? 1 : 0;
    for (uint8_t f : lp.lethal()) lethal += f;
    obs.clear(); blob(-3.f, 3.f);                      // the obstacle at (0, 2) leaves
    lp.setCloud(obs.data(), obs.size() / 4, 16);
    lp.markingUpdate(t_bs, t_gb);                      // cleared against the OLD observation: still blocked
    const dddmr_marking_stats m3 = lp.markingUpdate(t_bs, t_gb);
    std::printf("marking %u %u %u %zu %zu %u %u %zu\n", m1.n_clusters, m1.n_marked, m1.n_alive, touched, lethal, m3.n_cleared,
                m3.n_alive, lp.markedVoxels().size());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  return 0;
}
