// stand-in for <pcl/common/geometry.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
