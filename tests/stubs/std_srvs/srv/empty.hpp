// stand-in for <std_srvs/srv/empty.hpp>: see tests/stubs/README.md
#pragma once
#include "ros_stub_all.hpp"
