// stand-in for <tf2_ros/transform_listener.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
