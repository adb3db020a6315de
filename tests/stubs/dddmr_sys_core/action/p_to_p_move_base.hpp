// stand-in for the rosidl-generated <dddmr_sys_core/action/p_to_p_move_base.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
namespace dddmr_sys_core { namespace action {
struct PToPMoveBase {
  struct Goal { geometry_msgs::msg::PoseStamped target_pose; float target_value = 0; };
  struct Result { int32_t status = 0; std::string result; };
  struct Feedback { geometry_msgs::msg::TransformStamped base_position; std::string last_decision, current_decision; };
};
struct RecoveryBehaviors {
  struct Goal { std::string behavior_name; };
  struct Result { bool succeed = false; std::string info; };
  struct Feedback { bool undergoing = false; };
};
} }
