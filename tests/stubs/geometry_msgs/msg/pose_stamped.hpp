// stand-in for <geometry_msgs/msg/pose_stamped.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
