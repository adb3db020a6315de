// stand-in for <pcl/point_types.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
