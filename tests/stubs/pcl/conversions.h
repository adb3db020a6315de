// stand-in for <pcl/conversions.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
