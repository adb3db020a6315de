// planner_bridge.h -- the section local_planner.cpp:535-587 (re-initialise theories -> hasMoreTrajectories /
// nextTrajectory loop -> updateSharedData() -> getBestTrajectory()) as ONE call, with every return code of the library
// looked at.  Called by patches/local_planner_gpu_rollout.patch and patches/rotate_inplace_behavior_gpu_rollout.patch.
//
// A template over the ROS / PCL types it is handed (pcl::PointCloud<pcl::PointXYZI>, nav_msgs::msg::Path,
// geometry_msgs::msg::TransformStamped, nav_msgs::msg::Odometry, base_trajectory::Trajectory): this header includes
// none of them, so tests/test_adapters_cpu.py syntax-checks and RUNS it against minimal stand-in types.
#ifndef DDDMR_ROLLOUT_ADAPTER_PLANNER_BRIDGE_H_
#define DDDMR_ROLLOUT_ADAPTER_PLANNER_BRIDGE_H_

#include <string>
#include <vector>

#include "dddmr_rollout.h"
#include "dddmr_rollout_adapter/shared_context.h"

namespace dddmr_rollout_adapter
{

enum class TickOutcome
{
  kTrajectoryFound,          // best_traj filled                                  -> dddmr_sys_core::TRAJECTORY_FOUND path
  kAllTrajectoriesFail,      // every sample rejected: best_traj.cost_ = -1       -> ALL_TRAJECTORIES_FAIL
  kPerceptionMalfunction,    // the observation could not be handed over (too large, bad pointer): the tick did NOT run --
                             // never plan against the previous cloud                -> PERCEPTION_MALFUNCTION
  kEngineError               // prune plan / tick failed in the library (HIP error, capacity, unknown theory): logged,
                             // zero velocity                                        -> ALL_TRAJECTORIES_FAIL
};

// `error` receives dddmr_rollout_last_error() of the failing call (empty on the first two outcomes).
template<class ObsCloud, class Path, class TransformStamped, class Odometry, class Trajectory>
inline TickOutcome rolloutTick(
  dddmr_rollout_ctx * ctx, const ObsCloud & aggregate_observation, const Path & prune_plan, const TransformStamped & trans_gbl2b,
  const Odometry & robot_state, double allowed_max_linear_speed, double heading_deviation, const std::string & traj_gen_name,
  Trajectory & best_traj, dddmr_rollout_result * res_out, std::string * error)
{
  if (error) {error->clear();}
  best_traj = Trajectory();                       // cost_ -1, zero velocities (trajectory.cpp:34-37)
  dddmr_rollout_result res{};
  res.best_index = -1;
  if (res_out) {*res_out = res;}
  auto fail = [&](TickOutcome o) {
      if (error) {*error = dddmr_rollout_last_error(ctx);}
      return o;
    };
  // the aggregate observation: already on the device when the lidar plugin fed this cycle's scan (perception_bridge.h),
  // else the CPU aggregate (pcl::PointXYZI, 32-byte records)
  if (!SharedContext::consumeDeviceFeed()) {
    const size_t n = aggregate_observation.points.size();
    const int rc = dddmr_rollout_set_cloud(ctx, n ? &aggregate_observation.points[0].x : nullptr, n, sizeof(aggregate_observation.points[0]));
    if (rc != DDDMR_OK) {return fail(TickOutcome::kPerceptionMalfunction);}
  }
  std::vector<double> plan(7 * prune_plan.poses.size());
  for (size_t i = 0; i < prune_plan.poses.size(); ++i) {
    const auto & p = prune_plan.poses[i].pose;
    double * o = &plan[7 * i];
    o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
    o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
  }
  if (dddmr_rollout_set_prune_plan(ctx, plan.data(), prune_plan.poses.size()) != DDDMR_OK) {return fail(TickOutcome::kEngineError);}
  dddmr_tick_input in{};
  in.robot_pose[0] = trans_gbl2b.transform.translation.x;
  in.robot_pose[1] = trans_gbl2b.transform.translation.y;
  in.robot_pose[2] = trans_gbl2b.transform.translation.z;
  in.robot_pose[3] = trans_gbl2b.transform.rotation.x;
  in.robot_pose[4] = trans_gbl2b.transform.rotation.y;
  in.robot_pose[5] = trans_gbl2b.transform.rotation.z;
  in.robot_pose[6] = trans_gbl2b.transform.rotation.w;
  in.robot_twist[0] = robot_state.twist.twist.linear.x;
  in.robot_twist[1] = robot_state.twist.twist.linear.y;
  in.robot_twist[2] = robot_state.twist.twist.angular.z;
  in.allowed_max_linear_speed = allowed_max_linear_speed;
  in.heading_deviation = heading_deviation;
  const int rc = dddmr_rollout_tick(ctx, traj_gen_name.c_str(), &in, &res);
  if (res_out) {*res_out = res;}
  if (rc != DDDMR_OK) {return fail(TickOutcome::kEngineError);}      // (the library presets best_index = -1: looked at only now)
  if (res.best_index < 0) {return TickOutcome::kAllTrajectoriesFail;}
  best_traj.xv_ = res.vx;
  best_traj.yv_ = res.vy;
  best_traj.thetav_ = res.wz;
  best_traj.cost_ = res.best_cost;
  return TickOutcome::kTrajectoryFound;
}

// best_trajectory pose array (local_planner.cpp:472-478), on demand
template<class PoseArray>
inline int bestPoses(dddmr_rollout_ctx * ctx, PoseArray & out)
{
  size_t n_poses = 0;
  int rc = dddmr_rollout_get_best_poses(ctx, nullptr, 0, &n_poses);
  if (rc != DDDMR_OK) {return rc;}
  std::vector<double> poses(7 * n_poses);
  rc = dddmr_rollout_get_best_poses(ctx, poses.data(), n_poses, &n_poses);
  if (rc != DDDMR_OK) {return rc;}
  for (size_t i = 0; i < n_poses; ++i) {
    typename PoseArray::_poses_type::value_type a_pose;
    a_pose.position.x = poses[7 * i]; a_pose.position.y = poses[7 * i + 1]; a_pose.position.z = poses[7 * i + 2];
    a_pose.orientation.x = poses[7 * i + 3]; a_pose.orientation.y = poses[7 * i + 4]; a_pose.orientation.z = poses[7 * i + 5];
    a_pose.orientation.w = poses[7 * i + 6];
    out.poses.push_back(a_pose);
  }
  return DDDMR_OK;
}

}  // namespace dddmr_rollout_adapter
#endif
