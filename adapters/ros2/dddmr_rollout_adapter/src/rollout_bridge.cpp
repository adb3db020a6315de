// rollout_bridge.cpp -- see rollout_bridge.h.  NOT compiled in this repository's containers.
#include "dddmr_rollout_adapter/rollout_bridge.h"

#include <cstring>
#include <stdexcept>

#include <rclcpp/rclcpp.hpp>

namespace dddmr_rollout_adapter
{

RolloutBridge & RolloutBridge::instance()
{
  static RolloutBridge bridge;
  return bridge;
}

RolloutBridge::~RolloutBridge()
{
  if (ctx_) {dddmr_rollout_destroy(ctx_);}
}

void RolloutBridge::registerTheory(const dddmr_theory_config & theory)
{
  std::lock_guard<std::mutex> lk(mu_);
  if (ctx_) {
    throw std::runtime_error("dddmr_rollout_adapter: theory registered after the first tick");
  }
  theories_.push_back(theory);
}

void RolloutBridge::ensureContext()
{
  if (ctx_) {return;}
  dddmr_rollout_config cfg{};
  cfg.abi_version = DDDMR_ROLLOUT_ABI_VERSION;
  cfg.device = 0;
  cfg.rank = 0;
  cfg.world_size = 1;
  cfg.max_points = 600000;
  cfg.max_trajectories = 1u << 16;
  cfg.max_steps = 512;
  cfg.max_plan_poses = 512;
  cfg.n_theories = static_cast<int32_t>(theories_.size());
  cfg.theories = theories_.data();
  const int rc = dddmr_rollout_create(&cfg, &ctx_);
  if (rc != DDDMR_OK) {
    // no CPU fallback: the node must not come up half-working
    throw std::runtime_error("dddmr_rollout_create failed with " + std::to_string(rc));
  }
}

const std::vector<float> & RolloutBridge::beginBatch(
  const std::string & theory_name, const geometry_msgs::msg::TransformStamped & robot_pose,
  const nav_msgs::msg::Odometry & robot_state, double allowed_max_linear_speed)
{
  std::lock_guard<std::mutex> lk(mu_);
  ensureContext();
  Batch & b = batches_[theory_name];
  b.scored = false;
  b.in.robot_pose[0] = robot_pose.transform.translation.x;
  b.in.robot_pose[1] = robot_pose.transform.translation.y;
  b.in.robot_pose[2] = robot_pose.transform.translation.z;
  b.in.robot_pose[3] = robot_pose.transform.rotation.x;
  b.in.robot_pose[4] = robot_pose.transform.rotation.y;
  b.in.robot_pose[5] = robot_pose.transform.rotation.z;
  b.in.robot_pose[6] = robot_pose.transform.rotation.w;
  b.in.robot_twist[0] = robot_state.twist.twist.linear.x;
  b.in.robot_twist[1] = robot_state.twist.twist.linear.y;
  b.in.robot_twist[2] = robot_state.twist.twist.angular.z;
  b.in.allowed_max_linear_speed = allowed_max_linear_speed;
  b.in.heading_deviation = 0.0;   // known only to the critics' shared data: filled in by scoreBatch
  size_t n = 0;
  if (dddmr_rollout_samples(ctx_, theory_name.c_str(), &b.in, nullptr, 0, &n) != DDDMR_OK) {
    RCLCPP_FATAL(rclcpp::get_logger("dddmr_rollout_adapter"), "%s", dddmr_rollout_last_error(ctx_));
    b.samples.clear();
    return b.samples;   // zero trajectories, like an unknown theory in the reference (stacked_generator.cpp:82-91)
  }
  b.samples.resize(3 * n);
  dddmr_rollout_samples(ctx_, theory_name.c_str(), &b.in, b.samples.data(), n, &n);
  return b.samples;
}

void RolloutBridge::scoreBatch(
  const std::string & theory_name, const pcl::PointCloud<pcl::PointXYZI> & aggregate_observation,
  const nav_msgs::msg::Path & prune_plan, double heading_deviation)
{
  std::lock_guard<std::mutex> lk(mu_);
  Batch & b = batches_.at(theory_name);
  b.in.heading_deviation = heading_deviation;
  // pcl::PointXYZI is 32 bytes wide: x y z pad | intensity pad pad pad
  dddmr_rollout_set_cloud(
    ctx_, aggregate_observation.empty() ? nullptr : &aggregate_observation.points[0].x,
    aggregate_observation.size(), sizeof(pcl::PointXYZI));
  std::vector<double> plan(7 * prune_plan.poses.size());
  for (size_t i = 0; i < prune_plan.poses.size(); ++i) {
    const auto & p = prune_plan.poses[i].pose;
    double * o = &plan[7 * i];
    o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
    o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
  }
  dddmr_rollout_set_prune_plan(ctx_, plan.data(), prune_plan.poses.size());
  const int rc = dddmr_rollout_tick(ctx_, theory_name.c_str(), &b.in, &b.result);
  const size_t n = b.samples.size() / 3;
  b.costs.assign(n, DDDMR_COST_COLLISION);   // a failed tick rejects everything: zero velocity, ALL_TRAJECTORIES_FAIL
  if (rc == DDDMR_OK && b.result.n_local == n) {
    dddmr_rollout_debug dbg{};
    dbg.costs = b.costs.data();
    dddmr_rollout_get_debug(ctx_, &dbg);
  } else {
    RCLCPP_ERROR(rclcpp::get_logger("dddmr_rollout_adapter"), "tick failed (%d): %s", rc, dddmr_rollout_last_error(ctx_));
  }
  b.scored = true;
}

bool RolloutBridge::batchScored(const std::string & theory_name) const
{
  std::lock_guard<std::mutex> lk(mu_);
  auto it = batches_.find(theory_name);
  return it != batches_.end() && it->second.scored;
}

double RolloutBridge::cost(const std::string & theory_name, size_t index) const
{
  std::lock_guard<std::mutex> lk(mu_);
  const Batch & b = batches_.at(theory_name);
  return index < b.costs.size() ? b.costs[index] : DDDMR_COST_NOT_GENERATED;
}

const dddmr_rollout_result & RolloutBridge::result(const std::string & theory_name) const
{
  std::lock_guard<std::mutex> lk(mu_);
  return batches_.at(theory_name).result;
}

}  // namespace dddmr_rollout_adapter
