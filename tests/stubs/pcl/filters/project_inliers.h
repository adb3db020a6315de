// stand-in for <pcl/filters/project_inliers.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
