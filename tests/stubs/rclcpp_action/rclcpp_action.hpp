// stand-in for <rclcpp_action/rclcpp_action.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
