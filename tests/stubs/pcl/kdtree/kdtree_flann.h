// stand-in for <pcl/kdtree/kdtree_flann.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
