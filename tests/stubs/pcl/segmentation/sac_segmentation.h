// stand-in for <pcl/segmentation/sac_segmentation.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
