// stand-in for <tf2/LinearMath/Transform.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
