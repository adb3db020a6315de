// stand-in for <tf2_ros/buffer.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
