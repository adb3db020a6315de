// stand-in for <tf2_ros/create_timer_ros.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
