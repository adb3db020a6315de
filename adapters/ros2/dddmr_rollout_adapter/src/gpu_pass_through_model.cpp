// gpu_pass_through_model.cpp -- see gpu_pass_through_model.h.  NOT compiled in this repository's containers.
#include "dddmr_rollout_adapter/gpu_pass_through_model.h"

#include <pluginlib/class_list_macros.hpp>

#include "dddmr_rollout_adapter/rollout_bridge.h"

PLUGINLIB_EXPORT_CLASS(dddmr_rollout_adapter::GpuPassThroughModel, mpc_critics::ScoringModel)

namespace dddmr_rollout_adapter
{

void GpuPassThroughModel::onInitialize()
{
  // the theory this critic is bound to: the same "<critic>.trajectory_generator" parameter MPC_Critics_ROS
  // reads to file the critic under a theory (mpc_critics_ros.cpp:71-79)
  node_->get_parameter(name_ + ".trajectory_generator", theory_name_);
}

double GpuPassThroughModel::scoreTrajectory(base_trajectory::Trajectory & traj)
{
  RolloutBridge & bridge = RolloutBridge::instance();
  if (!bridge.batchScored(theory_name_)) {
    // first trajectory of the tick: updateSharedData() (local_planner.cpp:586) has just made the current
    // aggregate observation, prune plan and heading_deviation_ visible -- score the whole batch now
    bridge.scoreBatch(theory_name_, *shared_data_->pcl_perception_, shared_data_->prune_plan_, shared_data_->heading_deviation_);
  }
  const size_t index = static_cast<size_t>(traj.time_delta_);   // set by GpuRolloutTheory::nextTrajectory
  // StackedScoringModel adds a non-negative return to cost_ (0 from the generator) and stores a negative one
  // as the reject code (stacked_scoring_model.cpp:83-90): either way cost_ ends up as the engine's value
  return bridge.cost(theory_name_, index);
}

}  // namespace dddmr_rollout_adapter
