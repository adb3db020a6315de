// stand-in for <pcl/common/transforms.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
