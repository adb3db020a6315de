// stand-in for <pcl/common/common.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
