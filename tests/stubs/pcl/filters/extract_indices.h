// stand-in for <pcl/filters/extract_indices.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
