// dddmr_rollout.hpp -- C++ host-side mirror of the reference's local-planner
// surface over the C-ABI (include/dddmr_rollout.h).  Header-only, no ROS, no
// HIP types: this is what a C++ caller (the patched Local_Planner, the ROS 2
// adapter of INTEGRATION.md, a unit test) includes.
//
// Names and argument meaning follow the reference:
//   base_trajectory::Trajectory      base_trajectory/include/base_trajectory/trajectory.h:47-126
//   dddmr_sys_core::PlannerState     dddmr_sys_core/include/dddmr_sys_core/dddmr_enum_states.h:46-54
//   Local_Planner::computeVelocityCommand / setPlan
//                                    local_planner/include/local_planner/local_planner.h:72-85
// (paths relative to /root/reference/src/dddmr_local_planner/ or /root/reference/src/).
#ifndef DDDMR_ROLLOUT_HPP_
#define DDDMR_ROLLOUT_HPP_

#include <array>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "dddmr_rollout.h"

namespace dddmr_amd {

// dddmr_sys_core::PlannerState (same numeric values)
enum PlannerState {
  TF_FAIL = DDDMR_TF_FAIL,
  PRUNE_PLAN_FAIL = DDDMR_PRUNE_PLAN_FAIL,
  ALL_TRAJECTORIES_FAIL = DDDMR_ALL_TRAJECTORIES_FAIL,
  PERCEPTION_MALFUNCTION = DDDMR_PERCEPTION_MALFUNCTION,
  TRAJECTORY_FOUND = DDDMR_TRAJECTORY_FOUND,
  PATH_BLOCKED_WAIT = DDDMR_PATH_BLOCKED_WAIT,
  PATH_BLOCKED_REPLANNING = DDDMR_PATH_BLOCKED_REPLANNING
};

// The fields every consumer of best_traj reads (p2p_move_base.cpp:338,415,492);
// default-constructed like trajectory.cpp:34-37.
struct Trajectory {
  double xv_ = 0.0, yv_ = 0.0, thetav_ = 0.0;
  double cost_ = -1.0;
  double time_delta_ = 0.0;
  int index_ = -1;  // global sample index of the trajectory (build extension)
};

struct RolloutError : std::runtime_error {
  int code;
  RolloutError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// One rollout context = the trajectory generators + critics of one robot
// (Trajectory_Generators_ROS + MPC_Critics_ROS plugin sets of the reference).
class LocalPlanner {
 public:
  LocalPlanner(const std::vector<dddmr_theory_config>& theories, int device = 0,
               uint32_t max_points = 600000, uint32_t max_trajectories = 65536, uint32_t max_steps = 256,
               uint32_t max_plan_poses = 256, int rank = 0, int world_size = 1)
      : theories_(theories) {
    dddmr_rollout_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = DDDMR_ROLLOUT_ABI_VERSION;
    cfg.device = device;
    cfg.rank = rank;
    cfg.world_size = world_size;
    cfg.max_points = max_points;
    cfg.max_trajectories = max_trajectories;
    cfg.max_steps = max_steps;
    cfg.max_plan_poses = max_plan_poses;
    cfg.n_theories = (int32_t)theories_.size();
    cfg.theories = theories_.data();
    const int rc = dddmr_rollout_create(&cfg, &ctx_);
    if (rc != DDDMR_OK) throw RolloutError(rc, "dddmr_rollout_create failed (HIP device + gfx950 build required; no CPU fallback)");
  }
  ~LocalPlanner() { dddmr_rollout_destroy(ctx_); }
  LocalPlanner(const LocalPlanner&) = delete;
  LocalPlanner& operator=(const LocalPlanner&) = delete;

  // aggregate observation (global frame); PCL PointXYZI clouds pass stride 32
  void setCloud(const float* xyzi, size_t n_points, size_t stride_bytes) {
    check(dddmr_rollout_set_cloud(ctx_, xyzi, n_points, stride_bytes));
  }
  // fused local-mode perception feed (MultiLayerSpinningLidar::cbSensor)
  uint32_t setScan(const float* xyz, size_t n_points, size_t stride_bytes, const double T_base_sensor[7],
                   const double T_gbl_base[7], double perception_window_size, double marking_height) {
    uint32_t n_out = 0;
    check(dddmr_rollout_set_scan(ctx_, xyz, n_points, stride_bytes, T_base_sensor, T_gbl_base,
                                 perception_window_size, marking_height, &n_out));
    return n_out;
  }
  // prune plan poses, x y z qx qy qz qw each (output of Local_Planner::prunePlan)
  void setPlan(const double* poses_xyz_qxyzw, size_t n_poses) {
    check(dddmr_rollout_set_prune_plan(ctx_, poses_xyz_qxyzw, n_poses));
  }

  // Local_Planner::computeVelocityCommand(traj_gen_name, best_traj), the section
  // local_planner.cpp:535-587 (the caller keeps the TF / prune-plan / perception
  // guards and the perception opinions around it).
  PlannerState computeVelocityCommand(const std::string& traj_gen_name, Trajectory& best_traj,
                                      const dddmr_tick_input& in) {
    check(dddmr_rollout_tick(ctx_, traj_gen_name.c_str(), &in, &last_));
    best_traj.xv_ = last_.vx;
    best_traj.yv_ = last_.vy;
    best_traj.thetav_ = last_.wz;
    best_traj.cost_ = last_.best_cost;
    best_traj.index_ = last_.best_index;
    return static_cast<PlannerState>(last_.planner_state);
  }

  // multi-rank hosts: min-reduce lastResult().key over the ranks, then resolve
  const dddmr_rollout_result& lastResult() const { return last_; }
  dddmr_rollout_result resolve(int64_t reduced_key) {
    dddmr_rollout_result r = last_;
    check(dddmr_rollout_resolve(ctx_, reduced_key, &r));
    return r;
  }

  // best trajectory poses for the "best_trajectory" debug topic
  std::vector<std::array<double, 7>> bestPoses() {
    size_t n = 0;
    check(dddmr_rollout_get_best_poses(ctx_, nullptr, 0, &n));
    std::vector<std::array<double, 7>> poses(n);
    if (n) check(dddmr_rollout_get_best_poses(ctx_, &poses[0][0], n, &n));
    return poses;
  }

  // `trajectory` (accepted_only = false) / `accepted_trajectory` debug pose arrays
  std::vector<std::array<double, 7>> poseArrays(bool accepted_only) {
    size_t n = 0;
    check(dddmr_rollout_get_pose_arrays(ctx_, accepted_only ? 1 : 0, nullptr, 0, &n));
    std::vector<std::array<double, 7>> poses(n);
    if (n) check(dddmr_rollout_get_pose_arrays(ctx_, accepted_only ? 1 : 0, &poses[0][0], n, &n));
    return poses;
  }

  // PathBlockedStrategy::selfMark (path_blocked_strategy.cpp:56-100) on the current
  // aggregate observation; pcl_prune_plan: x y z intensity records as prunePlan fills
  // them (local_planner.cpp:402-430).  Returns prune_plan_blocked_ratio_ (percent).
  double pathBlockedRatio(const float* pcl_prune_plan_xyzi, size_t n_points, double check_radius,
                          dddmr_perception_opinion* opinion = nullptr) {
    double ratio = 0.0;
    int32_t op = DDDMR_OPINION_PASS;
    check(dddmr_rollout_path_blocked(ctx_, pcl_prune_plan_xyzi, n_points, check_radius, &ratio, &op, nullptr));
    if (opinion) *opinion = static_cast<dddmr_perception_opinion>(op);
    return ratio;
  }

 private:
  void check(int rc) {
    if (rc != DDDMR_OK) throw RolloutError(rc, dddmr_rollout_last_error(ctx_));
  }
  std::vector<dddmr_theory_config> theories_;
  dddmr_rollout_ctx* ctx_ = nullptr;
  dddmr_rollout_result last_{};
};

}  // namespace dddmr_amd
#endif
