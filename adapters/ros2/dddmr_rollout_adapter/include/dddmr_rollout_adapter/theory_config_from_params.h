// theory_config_from_params.h -- variant (ii) (adapters/ros2/README.md): build the rollout context from the
// reference's UNCHANGED YAML.  Reads, from the Trajectory_Generators_ROS node, every theory of `plugins` with the
// parameter names the reference's plugins declare (dd_simple_trajectory_generator_theory.cpp:47-234,
// omni_simple...:47-258, dd_rotate_inplace_theory.cpp:47-227) and, from the MPC_Critics_ROS node, every critic of
// `plugins` with its "<critic>.plugin" type, "<critic>.trajectory_generator" binding and weights
// (mpc_critics_ros.cpp:60-81, models/*.cpp onInitialize) -- in `plugins` order, which is the scoring order.
// NOT compiled in this repository's containers (no ROS 2 there).
#ifndef DDDMR_ROLLOUT_ADAPTER_THEORY_CONFIG_FROM_PARAMS_H_
#define DDDMR_ROLLOUT_ADAPTER_THEORY_CONFIG_FROM_PARAMS_H_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <rclcpp/rclcpp.hpp>

#include "dddmr_rollout.h"

namespace dddmr_rollout_adapter
{

template<typename T>
inline T readParam(rclcpp::Node & node, const std::string & name, const T & def)
{
  // the plugins declared their parameters when the node loaded them; undeclared ones take the plugin's default
  T v = def;
  if (node.has_parameter(name)) {node.get_parameter(name, v);}
  return v;
}

inline dddmr_theory_config theoryConfigFromParams(rclcpp::Node & gen, const std::string & name)
{
  dddmr_theory_config t;
  std::memset(&t, 0, sizeof(t));
  std::strncpy(t.name, name.c_str(), DDDMR_NAME_LEN - 1);
  const std::string type = readParam<std::string>(gen, name + ".plugin", "");
  if (type.find("DDSimple") != std::string::npos) {t.kind = DDDMR_THEORY_DD_SIMPLE;}
  else if (type.find("OmniSimple") != std::string::npos) {t.kind = DDDMR_THEORY_OMNI_SIMPLE;}
  else if (type.find("DDRotateInplace") != std::string::npos) {t.kind = DDDMR_THEORY_DD_ROTATE_INPLACE;}
  else {throw std::runtime_error("dddmr_rollout_adapter: no GPU equivalent of theory plugin " + type);}
  const std::string p = name + ".";
  t.use_motor_constraint = readParam<bool>(gen, p + "use_motor_constraint", false) ? 1 : 0;
  t.min_vel_x = readParam<double>(gen, p + "min_vel_x", 0.01);
  t.max_vel_x = readParam<double>(gen, p + "max_vel_x", 0.1);
  t.min_vel_y = readParam<double>(gen, p + "min_vel_y", -0.1);
  t.max_vel_y = readParam<double>(gen, p + "max_vel_y", 0.1);
  t.min_vel_trans = readParam<double>(gen, p + "min_vel_trans", 0.0);
  t.max_vel_trans = readParam<double>(gen, p + "max_vel_trans", 0.1);
  t.min_vel_theta = readParam<double>(gen, p + "min_vel_theta", 0.1);
  t.max_vel_theta = readParam<double>(gen, p + "max_vel_theta", 0.1);
  t.acc_lim_x = readParam<double>(gen, p + "acc_lim_x", 0.3);
  t.acc_lim_y = readParam<double>(gen, p + "acc_lim_y", 0.3);
  t.acc_lim_theta = readParam<double>(gen, p + "acc_lim_theta", 0.5);
  t.deceleration_ratio = readParam<double>(gen, p + "deceleration_ratio", 2.0);
  t.max_motor_shaft_rpm = readParam<double>(gen, p + "max_motor_shaft_rpm", 3000.0);
  t.wheel_diameter = readParam<double>(gen, p + "wheel_diameter", 0.15);
  t.gear_ratio = readParam<double>(gen, p + "gear_ratio", 30.0);
  t.robot_radius = readParam<double>(gen, p + "robot_radius", 0.25);
  t.controller_frequency = readParam<double>(gen, p + "controller_frequency", 10.0);
  t.sim_time = readParam<double>(gen, p + "sim_time", 2.0);
  t.linear_x_sample = readParam<double>(gen, p + "linear_x_sample", 10.0);
  t.linear_y_sample = readParam<double>(gen, p + "linear_y_sample", 10.0);
  t.angular_z_sample = readParam<double>(gen, p + "angular_z_sample", 10.0);
  t.sim_granularity = readParam<double>(gen, p + "sim_granularity", 0.1);
  t.angular_sim_granularity = readParam<double>(gen, p + "angular_sim_granularity", 0.05);
  t.rotation_speed = readParam<double>(gen, p + "rotation_speed", 0.4);
  // reference push order blb brb blt flb brt frt flt frb (dd_simple_trajectory_generator_theory.cpp:211-218)
  const char * order[8] = {"blb", "brb", "blt", "flb", "brt", "frt", "flt", "frb"};
  for (int v = 0; v < 8; ++v) {
    const auto xyz = readParam<std::vector<double>>(gen, p + "cuboid." + order[v], std::vector<double>{0.0, 0.0, 0.0});
    for (int a = 0; a < 3; ++a) {t.cuboid[v][a] = static_cast<float>(xyz.at(a));}
  }
  return t;
}

inline void appendCritics(rclcpp::Node & critics, std::vector<dddmr_theory_config> & theories)
{
  static const std::map<std::string, int> kinds = {
    {"CollisionMinMaxModel", DDDMR_CRITIC_COLLISION_MIN_MAX}, {"CollisionModel", DDDMR_CRITIC_COLLISION},
    {"StickPathModel", DDDMR_CRITIC_STICK_PATH}, {"PurePursuitModel", DDDMR_CRITIC_PURE_PURSUIT},
    {"TowardGlobalPlanModel", DDDMR_CRITIC_TOWARD_GLOBAL_PLAN}, {"ShortestAngleModel", DDDMR_CRITIC_SHORTEST_ANGLE},
    {"TwirlingModel", DDDMR_CRITIC_TWIRLING}};
  const auto names = readParam<std::vector<std::string>>(critics, "plugins", {});
  for (const auto & c : names) {
    const std::string type = readParam<std::string>(critics, c + ".plugin", "");
    const std::string bound = readParam<std::string>(critics, c + ".trajectory_generator", "");
    int kind = -1;
    for (const auto & kv : kinds) {
      if (type.size() >= kv.first.size() && type.compare(type.size() - kv.first.size(), kv.first.size(), kv.first) == 0 &&
          (kind < 0 || kv.first.size() > 0)) {
        // "CollisionMinMaxModel" also ends in "...Model": take the exact class name after "::"
        if (type.substr(type.rfind(':') + 1) == kv.first) {kind = kv.second;}
      }
    }
    if (kind < 0) {throw std::runtime_error("dddmr_rollout_adapter: no GPU equivalent of critic plugin " + type);}
    for (auto & t : theories) {
      if (bound != t.name) {continue;}
      if (t.n_critics >= DDDMR_MAX_CRITICS) {throw std::runtime_error("too many critics for " + bound);}
      dddmr_critic_config & cc = t.critics[t.n_critics++];
      cc.kind = kind;
      cc.weight = readParam<double>(critics, c + ".weight", 1.0);
      cc.translation_weight = readParam<double>(critics, c + ".translation_weight", 0.5);
      cc.orientation_weight = readParam<double>(critics, c + ".orientation_weight", 0.5);
    }
  }
}

// Capacities of the context, derived from what the YAML can make a tick ask for (a theory's sample grid, its longest
// horizon) instead of literals; every one can be overridden on the generators node ("gpu_rollout.max_points" ...).
// max_points has no counterpart in the reference's parameters (the aggregate observation is as large as the sensors make
// it): default 2^20 - 1, the library's limit; a larger cloud makes the tick fail loudly (PERCEPTION_MALFUNCTION in
// the patched planner), never plan against an older one.
inline void sizeContext(rclcpp::Node & generators, const std::vector<dddmr_theory_config> & theories, dddmr_rollout_config & cfg)
{
  double samples = 64.0, steps = 32.0;
  for (const auto & t : theories) {
    const double nx = std::max(2.0, t.linear_x_sample) + 1.0, ny = std::max(2.0, t.linear_y_sample) + 1.0;   // (+1: the inserted zero)
    const double nth = std::max(2.0, t.angular_z_sample) + 1.0;
    samples = std::max(samples, t.kind == DDDMR_THEORY_OMNI_SIMPLE ? nx * ny * nth : (t.kind == DDDMR_THEORY_DD_SIMPLE ? nx * nth : 2.0));
    const double vmax = std::max({std::fabs(t.max_vel_x), std::fabs(t.min_vel_x), std::fabs(t.max_vel_y), std::fabs(t.max_vel_trans)});
    double s = std::max(vmax * t.sim_time / std::max(t.sim_granularity, 1e-3), std::fabs(t.max_vel_theta) * t.sim_time / std::max(t.angular_sim_granularity, 1e-3));
    if (t.kind == DDDMR_THEORY_DD_ROTATE_INPLACE) {s = 6.28 / std::max(t.angular_sim_granularity, 1e-3);}
    steps = std::max(steps, std::ceil(s) + 2.0);
  }
  cfg.max_trajectories = static_cast<uint32_t>(readParam<int>(generators, "gpu_rollout.max_trajectories", static_cast<int>(std::min(samples * 1.25, 1048576.0))));
  cfg.max_steps = static_cast<uint32_t>(readParam<int>(generators, "gpu_rollout.max_steps", static_cast<int>(std::min(steps, 700.0))));
  cfg.max_plan_poses = static_cast<uint32_t>(readParam<int>(generators, "gpu_rollout.max_plan_poses", 512));
  cfg.max_points = static_cast<uint32_t>(readParam<int>(generators, "gpu_rollout.max_points", (1 << 20) - 1));
  cfg.device = readParam<int>(generators, "gpu_rollout.device", cfg.device);
}

// One context for all theories of the generators node with the critic stacks of the critics node.
inline dddmr_rollout_ctx * createContextFromNodes(rclcpp::Node & generators, rclcpp::Node & critics, int device = 0)
{
  std::vector<dddmr_theory_config> theories;
  for (const auto & name : readParam<std::vector<std::string>>(generators, "plugins", {})) {
    theories.push_back(theoryConfigFromParams(generators, name));
  }
  appendCritics(critics, theories);
  dddmr_rollout_config cfg{};
  cfg.abi_version = DDDMR_ROLLOUT_ABI_VERSION;
  cfg.device = device;
  cfg.world_size = 1;
  sizeContext(generators, theories, cfg);
  cfg.n_theories = static_cast<int32_t>(theories.size());
  cfg.theories = theories.data();
  dddmr_rollout_ctx * ctx = nullptr;
  const int rc = dddmr_rollout_create(&cfg, &ctx);
  if (rc != DDDMR_OK) {throw std::runtime_error("dddmr_rollout_create failed with " + std::to_string(rc) + " (no CPU fallback exists)");}
  return ctx;
}

}  // namespace dddmr_rollout_adapter
#endif
