// stand-in for <tf2/utils.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
