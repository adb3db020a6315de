// stand-in for <geometry_msgs/msg/point.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
