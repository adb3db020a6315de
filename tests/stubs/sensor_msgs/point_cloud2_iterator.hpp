// stand-in for <sensor_msgs/point_cloud2_iterator.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
