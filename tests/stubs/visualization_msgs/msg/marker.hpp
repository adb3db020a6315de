// stand-in for <visualization_msgs/msg/marker.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
