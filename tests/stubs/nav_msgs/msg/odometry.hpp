// stand-in for <nav_msgs/msg/odometry.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
