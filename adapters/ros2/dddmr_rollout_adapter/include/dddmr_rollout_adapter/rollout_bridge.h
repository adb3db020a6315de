// rollout_bridge.h -- process-wide owner of the MI355X rollout context shared by the theory plugin and
// the pass-through critic of variant (i) (adapters/ros2/README.md).  NOT compiled in this repository's
// containers (no ROS 2 there); written against the reference's headers:
//   trajectory_generators/include/trajectory_generators/trajectory_shared_data.h:58-75  (inputs of a tick)
//   mpc_critics/include/mpc_critics/model_shared_data.h:67-116                          (cloud, plan, heading_deviation_)
#ifndef DDDMR_ROLLOUT_ADAPTER_ROLLOUT_BRIDGE_H_
#define DDDMR_ROLLOUT_ADAPTER_ROLLOUT_BRIDGE_H_

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <geometry_msgs/msg/transform_stamped.hpp>
#include <nav_msgs/msg/odometry.hpp>
#include <nav_msgs/msg/path.hpp>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include "dddmr_rollout.h"

namespace dddmr_rollout_adapter
{

class RolloutBridge
{
public:
  static RolloutBridge & instance();

  // theory plugins register their configuration while the node loads them (onInitialize)
  void registerTheory(const dddmr_theory_config & theory);

  // initialise() of a theory: remember the tick's inputs, return the sample list
  // (local_planner.cpp:528-535 copies exactly these four fields into the generator shared data)
  const std::vector<float> & beginBatch(
    const std::string & theory_name, const geometry_msgs::msg::TransformStamped & robot_pose,
    const nav_msgs::msg::Odometry & robot_state, double allowed_max_linear_speed);

  // first scoreTrajectory() of the tick: upload cloud + prune plan, run dddmr_rollout_tick, fetch costs
  void scoreBatch(
    const std::string & theory_name, const pcl::PointCloud<pcl::PointXYZI> & aggregate_observation,
    const nav_msgs::msg::Path & prune_plan, double heading_deviation);

  bool batchScored(const std::string & theory_name) const;
  // cost of sample `index` of the scored batch (reject codes -1 / -4 / -12, -100 = not generated)
  double cost(const std::string & theory_name, size_t index) const;
  const dddmr_rollout_result & result(const std::string & theory_name) const;

private:
  RolloutBridge() = default;
  ~RolloutBridge();
  void ensureContext();

  struct Batch
  {
    dddmr_tick_input in{};
    std::vector<float> samples;   // [n][3]
    std::vector<double> costs;
    dddmr_rollout_result result{};
    bool scored = false;
  };

  mutable std::mutex mu_;
  std::vector<dddmr_theory_config> theories_;
  std::map<std::string, Batch> batches_;
  dddmr_rollout_ctx * ctx_ = nullptr;
};

}  // namespace dddmr_rollout_adapter
#endif
