"""ctypes view of include/dddmr_rollout.h (the C-ABI of the rollout engine).

The struct layouts here mirror the header field by field; `tests/test_capi_cpu.py`
checks sizeof() of every struct against the compiled library
(`dddmr_rollout_sizeof`) so the two cannot drift apart silently.

There is deliberately no CPU fallback: if the HIP library is missing or no HIP
device is usable, loading / `dddmr_rollout_create` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_CRITICS = 8
NAME_LEN = 64
ABI_VERSION = 2

# dddmr_status
OK = 0
ERR_BAD_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_UNKNOWN_THEORY = -4
ERR_CAPACITY = -5
ERR_STATE = -6
OPINION_PASS = 0                 # perception_3d::PerceptionOpinion
OPINION_PATH_BLOCKED_WAIT = 1

# dddmr_planner_state (dddmr_sys_core/include/dddmr_sys_core/dddmr_enum_states.h:46-54)
TF_FAIL = 0
PRUNE_PLAN_FAIL = 1
ALL_TRAJECTORIES_FAIL = 2
PERCEPTION_MALFUNCTION = 3
TRAJECTORY_FOUND = 4
PATH_BLOCKED_WAIT = 5
PATH_BLOCKED_REPLANNING = 6

# dddmr_theory_kind
THEORY_DD_SIMPLE = 0
THEORY_OMNI_SIMPLE = 1
THEORY_DD_ROTATE_INPLACE = 2

# dddmr_critic_kind
CRITIC_COLLISION = 0
CRITIC_COLLISION_MIN_MAX = 1
CRITIC_STICK_PATH = 2
CRITIC_PURE_PURSUIT = 3
CRITIC_TOWARD_GLOBAL_PLAN = 4
CRITIC_SHORTEST_ANGLE = 5
CRITIC_TWIRLING = 6

COST_COLLISION = -1.0
COST_PURE_PURSUIT_GUARD = -4.0
COST_NN_FAIL = -12.0
COST_NOT_GENERATED = -100.0

KEY_NONE = (1 << 63) - 1
COMM_ID_BYTES = 128


class CriticConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("reserved", C.c_int32),
        ("weight", C.c_double),
        ("translation_weight", C.c_double),
        ("orientation_weight", C.c_double),
    ]


class TheoryConfig(C.Structure):
    _fields_ = [
        ("name", C.c_char * NAME_LEN),
        ("kind", C.c_int32),
        ("use_motor_constraint", C.c_int32),
        ("min_vel_x", C.c_double), ("max_vel_x", C.c_double),
        ("min_vel_y", C.c_double), ("max_vel_y", C.c_double),
        ("min_vel_trans", C.c_double), ("max_vel_trans", C.c_double),
        ("min_vel_theta", C.c_double), ("max_vel_theta", C.c_double),
        ("acc_lim_x", C.c_double), ("acc_lim_y", C.c_double), ("acc_lim_theta", C.c_double),
        ("deceleration_ratio", C.c_double),
        ("max_motor_shaft_rpm", C.c_double), ("wheel_diameter", C.c_double),
        ("gear_ratio", C.c_double), ("robot_radius", C.c_double),
        ("controller_frequency", C.c_double),
        ("sim_time", C.c_double),
        ("linear_x_sample", C.c_double), ("linear_y_sample", C.c_double),
        ("angular_z_sample", C.c_double),
        ("sim_granularity", C.c_double), ("angular_sim_granularity", C.c_double),
        ("rotation_speed", C.c_double),
        ("cuboid", (C.c_float * 3) * 8),
        ("bench_fixed_steps", C.c_int32),
        ("bench_no_zero_insert", C.c_int32),
        ("n_critics", C.c_int32),
        ("reserved", C.c_int32),
        ("critics", CriticConfig * MAX_CRITICS),
    ]


class RolloutConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("device", C.c_int32),
        ("rank", C.c_int32),
        ("world_size", C.c_int32),
        ("max_points", C.c_uint32),
        ("max_trajectories", C.c_uint32),
        ("max_steps", C.c_uint32),
        ("max_plan_poses", C.c_uint32),
        ("n_theories", C.c_int32),
        ("reserved", C.c_int32),
        ("theories", C.POINTER(TheoryConfig)),
    ]


class TickInput(C.Structure):
    _fields_ = [
        ("robot_pose", C.c_double * 7),
        ("robot_twist", C.c_double * 3),
        ("allowed_max_linear_speed", C.c_double),
        ("heading_deviation", C.c_double),
    ]


class RolloutResult(C.Structure):
    _fields_ = [
        ("planner_state", C.c_int32),
        ("best_index", C.c_int32),
        ("best_cost", C.c_double),
        ("vx", C.c_double), ("vy", C.c_double), ("wz", C.c_double),
        ("n_samples", C.c_uint32),
        ("n_local", C.c_uint32),
        ("local_begin", C.c_uint32),
        ("n_points_binned", C.c_uint32),
        ("key", C.c_int64),
        ("device_ms", C.c_float),
        ("score_ms", C.c_float),
    ]


class MarkingConfig(C.Structure):
    """dddmr_marking_config: the lidar plugin's global-mode parameters
    (dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:73-139) + the node's radii."""
    _fields_ = [
        ("xy_resolution", C.c_double), ("height_resolution", C.c_double),
        ("marking_height", C.c_double), ("perception_window_size", C.c_double),
        ("vertical_FOV_top", C.c_double), ("vertical_FOV_bottom", C.c_double),
        ("scan_effective_positive_start", C.c_double), ("scan_effective_positive_end", C.c_double),
        ("scan_effective_negative_start", C.c_double), ("scan_effective_negative_end", C.c_double),
        ("euclidean_cluster_extraction_tolerance", C.c_double),
        ("euclidean_cluster_extraction_min_cluster_size", C.c_int32),
        ("reserved", C.c_int32),
        ("segmentation_ignore_ratio", C.c_double),
        ("inscribed_radius", C.c_double), ("inflation_radius", C.c_double), ("max_obstacle_distance", C.c_double),
        ("max_markings", C.c_uint32), ("max_cluster_points", C.c_uint32),
    ]


class MarkingStats(C.Structure):
    _fields_ = [
        ("n_observation", C.c_uint32), ("n_clusters", C.c_uint32), ("n_marked", C.c_uint32),
        ("n_in_window", C.c_uint32), ("n_cleared", C.c_uint32), ("n_alive", C.c_uint32),
        ("clear_ms", C.c_float), ("mark_ms", C.c_float),
    ]


class RolloutDebug(C.Structure):
    _fields_ = [
        ("costs", C.POINTER(C.c_double)),
        ("steps", C.POINTER(C.c_int32)),
        ("samples", C.POINTER(C.c_float)),
    ]


# every symbol include/dddmr_rollout.h declares
EXPORTED_SYMBOLS = (
    "dddmr_rollout_create",
    "dddmr_rollout_destroy",
    "dddmr_rollout_set_cloud",
    "dddmr_rollout_set_scan",
    "dddmr_rollout_set_stitcher",
    "dddmr_rollout_get_cloud",
    "dddmr_rollout_set_prune_plan",
    "dddmr_rollout_samples",
    "dddmr_rollout_tick",
    "dddmr_rollout_tick_begin",
    "dddmr_rollout_tick_end",
    "dddmr_rollout_resolve",
    "dddmr_rollout_winner_words",
    "dddmr_rollout_resolve_words",
    "dddmr_rollout_get_debug",
    "dddmr_rollout_get_best_poses",
    "dddmr_rollout_get_best_cuboids",
    "dddmr_rollout_get_pose_arrays",
    "dddmr_rollout_path_blocked",
    "dddmr_rollout_pack_key",
    "dddmr_rollout_key_index",
    "dddmr_rollout_comm_unique_id",
    "dddmr_rollout_comm_init",
    "dddmr_rollout_comm_destroy",
    "dddmr_rollout_comm_ranks",
    "dddmr_rollout_device_count",
    "dddmr_rollout_comm_loopback",
    "dddmr_rollout_comm_loopback_set_peer",
    "dddmr_rollout_marking_create",
    "dddmr_rollout_marking_update",
    "dddmr_rollout_marking_reset",
    "dddmr_rollout_marking_get_voxels",
    "dddmr_rollout_marking_get_points",
    "dddmr_rollout_marking_get_dgraph",
    "dddmr_rollout_marking_get_lethal",
    "dddmr_rollout_marking_route_counts",
    "dddmr_rollout_set_scan_source",
    "dddmr_rollout_set_stitcher_source",
    "dddmr_rollout_stream_ceiling",
    "dddmr_rollout_selftest_sincos",
    "dddmr_rollout_last_error",
    "dddmr_rollout_version",
)

_LIB_NAME = "libdddmr_rollout.so"
_lib = None


def library_path() -> str:
    # DDDMR_LIB_NAME selects the diagnostic build (phase stamps) for tools/phase_stamps.py
    name = os.environ.get("DDDMR_LIB_NAME", _LIB_NAME)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", name)


def load_library() -> C.CDLL:
    """Load the HIP engine.  Raises (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the rollout engine."
        )
    lib = C.CDLL(path)
    ctx_p = C.c_void_p
    lib.dddmr_rollout_create.argtypes = [C.POINTER(RolloutConfig), C.POINTER(ctx_p)]
    lib.dddmr_rollout_create.restype = C.c_int
    lib.dddmr_rollout_destroy.argtypes = [ctx_p]
    lib.dddmr_rollout_destroy.restype = None
    lib.dddmr_rollout_set_cloud.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.c_size_t]
    lib.dddmr_rollout_set_cloud.restype = C.c_int
    lib.dddmr_rollout_set_scan.argtypes = [
        ctx_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double),
        C.c_double, C.c_double, C.POINTER(C.c_uint32)]
    lib.dddmr_rollout_set_scan.restype = C.c_int
    lib.dddmr_rollout_set_stitcher.argtypes = [ctx_p, C.c_int32]
    lib.dddmr_rollout_set_stitcher.restype = C.c_int
    lib.dddmr_rollout_get_cloud.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_get_cloud.restype = C.c_int
    lib.dddmr_rollout_set_prune_plan.argtypes = [ctx_p, C.c_void_p, C.c_size_t]
    lib.dddmr_rollout_set_prune_plan.restype = C.c_int
    lib.dddmr_rollout_samples.argtypes = [ctx_p, C.c_char_p, C.POINTER(TickInput), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_samples.restype = C.c_int
    lib.dddmr_rollout_tick.argtypes = [ctx_p, C.c_char_p, C.POINTER(TickInput), C.POINTER(RolloutResult)]
    lib.dddmr_rollout_tick.restype = C.c_int
    lib.dddmr_rollout_tick_begin.argtypes = [ctx_p, C.c_char_p, C.POINTER(TickInput)]
    lib.dddmr_rollout_tick_begin.restype = C.c_int
    lib.dddmr_rollout_tick_end.argtypes = [ctx_p, C.POINTER(RolloutResult)]
    lib.dddmr_rollout_tick_end.restype = C.c_int
    lib.dddmr_rollout_resolve.argtypes = [ctx_p, C.c_int64, C.POINTER(RolloutResult)]
    lib.dddmr_rollout_resolve.restype = C.c_int
    lib.dddmr_rollout_winner_words.argtypes = [C.POINTER(RolloutResult), C.POINTER(C.c_int64)]
    lib.dddmr_rollout_winner_words.restype = None
    lib.dddmr_rollout_resolve_words.argtypes = [ctx_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(RolloutResult)]
    lib.dddmr_rollout_resolve_words.restype = C.c_int
    lib.dddmr_rollout_get_debug.argtypes = [ctx_p, C.POINTER(RolloutDebug)]
    lib.dddmr_rollout_get_debug.restype = C.c_int
    lib.dddmr_rollout_get_best_poses.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_get_best_poses.restype = C.c_int
    lib.dddmr_rollout_get_best_cuboids.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_get_best_cuboids.restype = C.c_int
    lib.dddmr_rollout_get_pose_arrays.argtypes = [ctx_p, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_get_pose_arrays.restype = C.c_int
    lib.dddmr_rollout_path_blocked.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.c_double, C.POINTER(C.c_double),
                                               C.POINTER(C.c_int32), C.c_void_p]
    lib.dddmr_rollout_path_blocked.restype = C.c_int
    lib.dddmr_rollout_pack_key.argtypes = [C.c_double, C.c_uint32]
    lib.dddmr_rollout_pack_key.restype = C.c_int64
    lib.dddmr_rollout_key_index.argtypes = [C.c_int64]
    lib.dddmr_rollout_key_index.restype = C.c_int32
    lib.dddmr_rollout_comm_unique_id.argtypes = [C.c_void_p]
    lib.dddmr_rollout_comm_unique_id.restype = C.c_int
    lib.dddmr_rollout_comm_init.argtypes = [ctx_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.dddmr_rollout_comm_init.restype = C.c_int
    lib.dddmr_rollout_comm_destroy.argtypes = [ctx_p]
    lib.dddmr_rollout_comm_destroy.restype = C.c_int
    lib.dddmr_rollout_device_count.argtypes = [C.POINTER(C.c_int32)]
    lib.dddmr_rollout_device_count.restype = C.c_int
    lib.dddmr_rollout_comm_ranks.argtypes = [ctx_p, C.POINTER(C.c_int32)]
    lib.dddmr_rollout_comm_ranks.restype = C.c_int
    lib.dddmr_rollout_comm_loopback.argtypes = [ctx_p]
    lib.dddmr_rollout_comm_loopback.restype = C.c_int
    lib.dddmr_rollout_comm_loopback_set_peer.argtypes = [ctx_p, C.c_int32, C.POINTER(C.c_int64)]
    lib.dddmr_rollout_comm_loopback_set_peer.restype = C.c_int
    lib.dddmr_rollout_marking_create.argtypes = [ctx_p, C.POINTER(MarkingConfig), C.c_void_p, C.c_size_t, C.c_size_t,
                                                 C.c_void_p, C.c_size_t, C.c_size_t]
    lib.dddmr_rollout_marking_create.restype = C.c_int
    lib.dddmr_rollout_marking_update.argtypes = [ctx_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(MarkingStats)]
    lib.dddmr_rollout_marking_update.restype = C.c_int
    lib.dddmr_rollout_marking_reset.argtypes = [ctx_p]
    lib.dddmr_rollout_marking_reset.restype = C.c_int
    lib.dddmr_rollout_marking_get_voxels.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_marking_get_voxels.restype = C.c_int
    lib.dddmr_rollout_marking_get_points.argtypes = [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dddmr_rollout_marking_get_points.restype = C.c_int
    lib.dddmr_rollout_marking_get_dgraph.argtypes = [ctx_p, C.c_void_p, C.c_size_t]
    lib.dddmr_rollout_marking_get_dgraph.restype = C.c_int
    lib.dddmr_rollout_marking_get_lethal.argtypes = [ctx_p, C.c_void_p, C.c_size_t]
    lib.dddmr_rollout_marking_get_lethal.restype = C.c_int
    lib.dddmr_rollout_marking_route_counts.argtypes = [ctx_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.dddmr_rollout_marking_route_counts.restype = C.c_int
    lib.dddmr_rollout_set_scan_source.argtypes = [ctx_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.dddmr_rollout_set_scan_source.restype = C.c_int
    lib.dddmr_rollout_set_stitcher_source.argtypes = [ctx_p, C.c_int32, C.c_int32]
    lib.dddmr_rollout_set_stitcher_source.restype = C.c_int
    lib.dddmr_rollout_stream_ceiling.argtypes = [ctx_p, C.c_size_t, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dddmr_rollout_stream_ceiling.restype = C.c_int
    lib.dddmr_rollout_selftest_sincos.argtypes = [ctx_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.dddmr_rollout_selftest_sincos.restype = C.c_int
    lib.dddmr_rollout_last_error.argtypes = [ctx_p]
    lib.dddmr_rollout_last_error.restype = C.c_char_p
    lib.dddmr_rollout_version.argtypes = []
    lib.dddmr_rollout_version.restype = C.c_char_p
    # not part of the public header: layout self-check used by the CPU tests
    lib.dddmr_rollout_sizeof.argtypes = [C.c_int]
    lib.dddmr_rollout_sizeof.restype = C.c_size_t
    _lib = lib
    return lib
