// stand-in for <sensor_msgs/msg/point_cloud2.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
