// gpu_rollout_theory.cpp -- see gpu_rollout_theory.h.  NOT compiled in this repository's containers.
#include "dddmr_rollout_adapter/gpu_rollout_theory.h"

#include <cstring>
#include <map>
#include <stdexcept>

#include <pluginlib/class_list_macros.hpp>

#include "dddmr_rollout_adapter/rollout_bridge.h"

PLUGINLIB_EXPORT_CLASS(dddmr_rollout_adapter::GpuRolloutTheory, trajectory_generators::TrajectoryGeneratorTheory)

namespace dddmr_rollout_adapter
{

namespace
{
template<typename T>
T param(const rclcpp::Node::SharedPtr & node, const std::string & name, const T & def)
{
  if (!node->has_parameter(name)) {node->declare_parameter(name, rclcpp::ParameterValue(def));}
  T v = def;
  node->get_parameter(name, v);
  return v;
}
}  // namespace

void GpuRolloutTheory::onInitialize()
{
  // `name_` is the theory name of the YAML (e.g. differential_drive_simple): critics bind to it through
  // "<critic>.trajectory_generator" (mpc_critics_ros.cpp:71-79)
  std::memset(&config_, 0, sizeof(config_));
  std::strncpy(config_.name, name_.c_str(), DDDMR_NAME_LEN - 1);
  const std::string wrapped = param<std::string>(node_, name_ + ".wrapped_theory", "dd_simple");
  static const std::map<std::string, int> kinds = {
    {"dd_simple", DDDMR_THEORY_DD_SIMPLE}, {"omni_simple", DDDMR_THEORY_OMNI_SIMPLE},
    {"dd_rotate_inplace", DDDMR_THEORY_DD_ROTATE_INPLACE}};
  if (!kinds.count(wrapped)) {throw std::runtime_error(name_ + ".wrapped_theory: " + wrapped);}
  config_.kind = kinds.at(wrapped);
  const std::string p = name_ + ".";
  // parameter names and defaults of the wrapped plugins (dd_simple...cpp:47-134, omni_simple...cpp:47-158,
  // dd_rotate_inplace_theory.cpp:47-129)
  config_.use_motor_constraint = param<bool>(node_, p + "use_motor_constraint", false) ? 1 : 0;
  config_.min_vel_x = param<double>(node_, p + "min_vel_x", 0.01);
  config_.max_vel_x = param<double>(node_, p + "max_vel_x", 0.1);
  config_.min_vel_y = param<double>(node_, p + "min_vel_y", -0.1);
  config_.max_vel_y = param<double>(node_, p + "max_vel_y", 0.1);
  config_.min_vel_trans = param<double>(node_, p + "min_vel_trans", 0.0);
  config_.max_vel_trans = param<double>(node_, p + "max_vel_trans", 0.1);
  config_.min_vel_theta = param<double>(node_, p + "min_vel_theta", 0.1);
  config_.max_vel_theta = param<double>(node_, p + "max_vel_theta", 0.1);
  config_.acc_lim_x = param<double>(node_, p + "acc_lim_x", 0.3);
  config_.acc_lim_y = param<double>(node_, p + "acc_lim_y", 0.3);
  config_.acc_lim_theta = param<double>(node_, p + "acc_lim_theta", 0.5);
  config_.deceleration_ratio = param<double>(node_, p + "deceleration_ratio", 2.0);
  config_.max_motor_shaft_rpm = param<double>(node_, p + "max_motor_shaft_rpm", 3000.0);
  config_.wheel_diameter = param<double>(node_, p + "wheel_diameter", 0.15);
  config_.gear_ratio = param<double>(node_, p + "gear_ratio", 30.0);
  config_.robot_radius = param<double>(node_, p + "robot_radius", 0.25);
  config_.controller_frequency = param<double>(node_, p + "controller_frequency", 10.0);
  config_.sim_time = param<double>(node_, p + "sim_time", 2.0);
  config_.linear_x_sample = param<double>(node_, p + "linear_x_sample", 10.0);
  config_.linear_y_sample = param<double>(node_, p + "linear_y_sample", 10.0);
  config_.angular_z_sample = param<double>(node_, p + "angular_z_sample", 10.0);
  config_.sim_granularity = param<double>(node_, p + "sim_granularity", 0.1);
  config_.angular_sim_granularity = param<double>(node_, p + "angular_sim_granularity", 0.05);
  config_.rotation_speed = param<double>(node_, p + "rotation_speed", 0.4);
  // cuboid: the eight named vertices, pushed in the reference order blb brb blt flb brt frt flt frb
  // (dd_simple_trajectory_generator_theory.cpp:211-218)
  const char * order[8] = {"blb", "brb", "blt", "flb", "brt", "frt", "flt", "frb"};
  for (int v = 0; v < 8; ++v) {
    const auto xyz = param<std::vector<double>>(node_, p + "cuboid." + order[v], std::vector<double>{0.0, 0.0, 0.0});
    for (int a = 0; a < 3; ++a) {config_.cuboid[v][a] = static_cast<float>(xyz.at(a));}
  }
  // the critic stack the engine applies, in this order (= the `plugins` order of mpc_critics, a9)
  static const std::map<std::string, int> critic_kinds = {
    {"collision", DDDMR_CRITIC_COLLISION}, {"collision_min_max", DDDMR_CRITIC_COLLISION_MIN_MAX},
    {"stick_path", DDDMR_CRITIC_STICK_PATH}, {"pure_pursuit", DDDMR_CRITIC_PURE_PURSUIT},
    {"toward_global_plan", DDDMR_CRITIC_TOWARD_GLOBAL_PLAN}, {"shortest_angle", DDDMR_CRITIC_SHORTEST_ANGLE},
    {"twirling", DDDMR_CRITIC_TWIRLING}};
  const auto critics = param<std::vector<std::string>>(node_, p + "critics", std::vector<std::string>{"collision"});
  if (critics.size() > DDDMR_MAX_CRITICS) {throw std::runtime_error(name_ + ".critics: too many");}
  config_.n_critics = static_cast<int32_t>(critics.size());
  for (size_t i = 0; i < critics.size(); ++i) {
    const std::string c = p + "critic." + critics[i] + ".";
    const std::string model = param<std::string>(node_, c + "model", critics[i]);
    if (!critic_kinds.count(model)) {throw std::runtime_error(c + "model: " + model);}
    config_.critics[i].kind = critic_kinds.at(model);
    config_.critics[i].weight = param<double>(node_, c + "weight", 1.0);
    config_.critics[i].translation_weight = param<double>(node_, c + "translation_weight", 0.5);
    config_.critics[i].orientation_weight = param<double>(node_, c + "orientation_weight", 0.5);
  }
  RolloutBridge::instance().registerTheory(config_);
}

void GpuRolloutTheory::initialise()
{
  // StackedGenerator::initializeTheories_wi_Shared_data re-initialises EVERY theory each tick
  // (stacked_generator.cpp:67-76); only the sample list is formed here (host side), nothing runs on the GPU
  // until the pass-through critic is asked for the first cost of the theory the caller actually iterates.
  samples_ = &RolloutBridge::instance().beginBatch(
    name_, shared_data_->robot_pose_, shared_data_->robot_state_, shared_data_->current_allowed_max_linear_speed_);
  next_ = 0;
}

bool GpuRolloutTheory::hasMoreTrajectories()
{
  return samples_ && next_ < samples_->size() / 3;
}

bool GpuRolloutTheory::nextTrajectory(base_trajectory::Trajectory & traj)
{
  if (!hasMoreTrajectories()) {return false;}
  const float * s = samples_->data() + 3 * next_;
  // a seed: velocities only.  time_delta_ carries the sample index for the pass-through critic (the reference's
  // consumers read xv_, yv_, thetav_ and cost_ only, p2p_move_base.cpp:338,415,492).
  traj = base_trajectory::Trajectory(s[0], s[1], s[2], static_cast<double>(next_), 0);
  traj.cost_ = 0.0;   // generators hand trajectories out with cost 0 (dd_simple...cpp:399)
  ++next_;
  return true;
}

}  // namespace dddmr_rollout_adapter
