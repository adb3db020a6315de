// stand-in for <tf2_eigen/tf2_eigen.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
