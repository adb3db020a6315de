// stand-in for <rclcpp/rclcpp.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
