// stand-in for <angles/angles.h>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
