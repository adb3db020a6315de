// stand-in for <pcl/features/normal_3d.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
