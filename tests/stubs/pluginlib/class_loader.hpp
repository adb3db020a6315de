// stand-in for <pluginlib/class_loader.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
