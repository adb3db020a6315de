// stand-in for <pcl/search/kdtree.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
