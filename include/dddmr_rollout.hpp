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

  // cbSensor's stitcher_num (multilayer_spinning_lidar.cpp:185-200): later setScan() calls feed the last n raw scans
  void setStitcher(int stitcher_num) { check(dddmr_rollout_set_stitcher(ctx_, stitcher_num)); }

  // the theory's initialise() alone: the (vx, vy, wz) sample list a tick with these inputs rolls out
  std::vector<std::array<float, 3>> samples(const std::string& traj_gen_name, const dddmr_tick_input& in) {
    size_t n = 0;
    check(dddmr_rollout_samples(ctx_, traj_gen_name.c_str(), &in, nullptr, 0, &n));
    std::vector<std::array<float, 3>> out(n);
    if (n) check(dddmr_rollout_samples(ctx_, traj_gen_name.c_str(), &in, &out[0][0], n, &n));
    return out;
  }

  // ---- multi-rank hosts (one context per GPU, created with rank / world_size) ----
  const dddmr_rollout_result& lastResult() const { return last_; }
  // (a) in-library exchange: RCCL communicator owned by the context; afterwards every tick returns the GLOBAL winner
  static std::array<uint8_t, DDDMR_COMM_ID_BYTES> commUniqueId() {
    std::array<uint8_t, DDDMR_COMM_ID_BYTES> id{};
    const int rc = dddmr_rollout_comm_unique_id(id.data());
    if (rc != DDDMR_OK) throw RolloutError(rc, "dddmr_rollout_comm_unique_id failed (librccl not loadable?)");
    return id;
  }
  void commInit(const std::array<uint8_t, DDDMR_COMM_ID_BYTES>& id, int rank, int n_ranks) {
    check(dddmr_rollout_comm_init(ctx_, id.data(), rank, n_ranks));
  }
  void commDestroy() { check(dddmr_rollout_comm_destroy(ctx_)); }
  int commRanks() {
    int32_t n = 0;
    check(dddmr_rollout_comm_ranks(ctx_, &n));
    return n;
  }
  // (b) host-side exchange: min-all-reduce a 2 * n_ranks int64 vector holding every rank's winnerWords() in its
  // slots (INT64_MAX elsewhere), then resolveWords() on every rank -- exact; or the 8-byte key + resolve()
  std::array<int64_t, 2> winnerWords() const {
    std::array<int64_t, 2> w{};
    dddmr_rollout_winner_words(&last_, w.data());
    return w;
  }
  dddmr_rollout_result resolveWords(const std::vector<int64_t>& slots) {
    dddmr_rollout_result r = last_;
    check(dddmr_rollout_resolve_words(ctx_, slots.data(), (int32_t)(slots.size() / 2), &r));
    return r;
  }
  dddmr_rollout_result resolve(int64_t reduced_key) {
    dddmr_rollout_result r = last_;
    check(dddmr_rollout_resolve(ctx_, reduced_key, &r));
    return r;
  }

  // ---- global-mode marking / clearing layer of the lidar plugin (multilayer_spinning_lidar.cpp:306-746) ----
  void markingCreate(const dddmr_marking_config& cfg, const float* ground_xyz, size_t n_ground, size_t ground_stride,
                     const float* map_xyz, size_t n_map, size_t map_stride) {
    check(dddmr_rollout_marking_create(ctx_, &cfg, ground_xyz, n_ground, ground_stride, map_xyz, n_map, map_stride));
    n_ground_ = n_ground;
  }
  // one doClear_then_Mark pass on the current aggregate observation (setScan / setCloud)
  dddmr_marking_stats markingUpdate(const double T_base_sensor[7], const double T_gbl_base[7]) {
    dddmr_marking_stats st{};
    check(dddmr_rollout_marking_update(ctx_, T_base_sensor, T_gbl_base, &st));
    return st;
  }
  void markingReset() { check(dddmr_rollout_marking_reset(ctx_)); }
  std::vector<double> dGraph() {                        // get_dGraphValue(index) for index 0..n_ground
    std::vector<double> v(n_ground_ + 1);
    check(dddmr_rollout_marking_get_dgraph(ctx_, v.data(), v.size()));
    return v;
  }
  std::vector<uint8_t> lethal() {                       // lethal_map_ keys as flags
    std::vector<uint8_t> v(n_ground_ + 1);
    check(dddmr_rollout_marking_get_lethal(ctx_, v.data(), v.size()));
    return v;
  }
  std::vector<std::array<int32_t, 3>> markedVoxels() {  // alive markings (the global_marking topic's keys)
    size_t n = 0;
    check(dddmr_rollout_marking_get_voxels(ctx_, nullptr, 0, &n));
    std::vector<std::array<int32_t, 3>> v(n);
    if (n) check(dddmr_rollout_marking_get_voxels(ctx_, &v[0][0], n, &n));
    return v;
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
  size_t n_ground_ = 0;
};

}  // namespace dddmr_amd
#endif
