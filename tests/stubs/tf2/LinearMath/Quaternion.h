// stand-in for <tf2/LinearMath/Quaternion.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
