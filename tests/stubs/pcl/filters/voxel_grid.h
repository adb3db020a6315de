// stand-in for <pcl/filters/voxel_grid.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
