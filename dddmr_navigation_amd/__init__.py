"""MI355X-native local-planner rollout engine for dddmr_navigation.

The product is the HIP library `csrc/libdddmr_rollout.so` behind the C-ABI of
`include/dddmr_rollout.h`; this package is the thin host-side mirror of the
reference's local-planner interface on top of it (no CPU fallback).
"""
from . import _capi, configs  # noqa: F401
from .local_planner import LocalPlanner, Trajectory, PlannerState  # noqa: F401

__all__ = ["LocalPlanner", "Trajectory", "PlannerState", "configs"]
