// stand-in for <pluginlib/class_list_macros.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
