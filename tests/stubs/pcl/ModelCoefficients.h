// stand-in for <pcl/ModelCoefficients.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
