// stand-in for <pcl/segmentation/extract_clusters.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
