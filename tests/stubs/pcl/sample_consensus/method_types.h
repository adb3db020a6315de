// stand-in for <pcl/sample_consensus/method_types.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
