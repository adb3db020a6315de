// stand-in for <tf2_geometry_msgs/tf2_geometry_msgs.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
