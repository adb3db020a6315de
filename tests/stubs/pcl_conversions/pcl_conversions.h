// stand-in for <pcl_conversions/pcl_conversions.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
