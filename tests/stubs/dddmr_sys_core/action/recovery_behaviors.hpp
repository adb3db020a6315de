// stand-in for the rosidl-generated <dddmr_sys_core/action/recovery_behaviors.hpp>: see tests/stubs/README.md
#pragma once
#include "dddmr_sys_core/action/p_to_p_move_base.hpp"
