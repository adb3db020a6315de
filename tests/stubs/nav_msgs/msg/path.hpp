// stand-in for <nav_msgs/msg/path.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
