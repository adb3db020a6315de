// stand-in for <geometry_msgs/msg/twist.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
