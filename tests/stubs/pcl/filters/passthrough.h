// stand-in for <pcl/filters/passthrough.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
