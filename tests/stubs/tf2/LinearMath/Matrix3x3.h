// stand-in for <tf2/LinearMath/Matrix3x3.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
