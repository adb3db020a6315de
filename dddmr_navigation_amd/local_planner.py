"""Host-side mirror of the reference's local-planner surface on top of the C-ABI.

Names and argument meaning follow
/root/reference/src/dddmr_local_planner/local_planner/include/local_planner/local_planner.h:72-85
(`computeVelocityCommand(traj_gen_name, best_traj) -> PlannerState`, `setPlan`)
and base_trajectory/include/base_trajectory/trajectory.h:47-126 (`Trajectory`
with `xv_, yv_, thetav_, cost_`).  All compute happens in the HIP library; this
file only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Iterable, Optional

import numpy as np

from . import _capi as K
from . import configs


class PlannerState(enum.IntEnum):
    """dddmr_sys_core/include/dddmr_sys_core/dddmr_enum_states.h:46-54"""
    TF_FAIL = 0
    PRUNE_PLAN_FAIL = 1
    ALL_TRAJECTORIES_FAIL = 2
    PERCEPTION_MALFUNCTION = 3
    TRAJECTORY_FOUND = 4
    PATH_BLOCKED_WAIT = 5
    PATH_BLOCKED_REPLANNING = 6


@dataclass
class Trajectory:
    """The fields consumers read from best_traj (p2p_move_base.cpp:338,415,492).
    Default-constructed values are trajectory.cpp:34-37."""
    xv_: float = 0.0
    yv_: float = 0.0
    thetav_: float = 0.0
    cost_: float = -1.0
    index: int = -1


class RolloutError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dddmr_rollout error {code}: {msg}")
        self.code = code


def device_count() -> int:
    """HIP devices this process sees (0 without a GPU); through the library, so no torch import is needed."""
    n = C.c_int32(0)
    rc = K.load_library().dddmr_rollout_device_count(C.byref(n))
    return int(n.value) if rc == K.OK else 0


class LocalPlanner:
    """One rollout context = the trajectory generators + critics of one robot."""

    def __init__(self, theories: Iterable[K.TheoryConfig], device: int = 0, max_points: int = 600_000,
                 max_trajectories: int = 65_536, max_steps: int = 256, max_plan_poses: int = 256,
                 rank: int = 0, world_size: int = 1):
        self._lib = K.load_library()
        self._theories = configs.theory_array(theories)
        cfg = K.RolloutConfig()
        cfg.abi_version = K.ABI_VERSION
        cfg.device = device
        cfg.rank = rank
        cfg.world_size = world_size
        cfg.max_points = max_points
        cfg.max_trajectories = max_trajectories
        cfg.max_steps = max_steps
        cfg.max_plan_poses = max_plan_poses
        cfg.n_theories = len(self._theories)
        cfg.theories = C.cast(self._theories, C.POINTER(K.TheoryConfig))
        self._ctx = C.c_void_p()
        rc = self._lib.dddmr_rollout_create(C.byref(cfg), C.byref(self._ctx))
        if rc != K.OK:
            self._ctx = C.c_void_p()
            raise RolloutError(rc, "dddmr_rollout_create failed (no CPU fallback exists; "
                                   "a HIP device and the gfx950 build are required)")
        self.last_result: Optional[K.RolloutResult] = None

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.dddmr_rollout_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != K.OK:
            msg = self._lib.dddmr_rollout_last_error(self._ctx)
            raise RolloutError(rc, msg.decode() if msg else "")

    # -- inputs ------------------------------------------------------------
    def set_cloud(self, cloud: np.ndarray):
        """Aggregate observation, [P, >=3] float32 rows (x y z [intensity ...])."""
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        if cloud.ndim != 2 or (cloud.shape[0] and cloud.shape[1] < 3):
            raise ValueError("cloud must be [P, >=3] float32")
        stride = cloud.strides[0] if cloud.shape[0] else 16
        self._check(self._lib.dddmr_rollout_set_cloud(self._ctx, cloud.ctypes.data_as(C.c_void_p),
                                                      cloud.shape[0], stride))

    def set_scan(self, scan_xyz: np.ndarray, T_base_sensor, T_gbl_base, perception_window_size: float,
                 marking_height: float) -> int:
        """Fused local-mode perception feed (cbSensor); returns the number of
        downsampled points now forming the aggregate observation."""
        scan = np.ascontiguousarray(scan_xyz, dtype=np.float32)
        if scan.ndim != 2 or (scan.shape[0] and scan.shape[1] < 3):
            raise ValueError("scan must be [P, >=3] float32")
        tbs = (C.c_double * 7)(*[float(v) for v in T_base_sensor])
        tgb = (C.c_double * 7)(*[float(v) for v in T_gbl_base])
        n_out = C.c_uint32(0)
        stride = scan.strides[0] if scan.shape[0] else 12
        self._check(self._lib.dddmr_rollout_set_scan(self._ctx, scan.ctypes.data_as(C.c_void_p), scan.shape[0],
                                                     stride, tbs, tgb, perception_window_size, marking_height,
                                                     C.byref(n_out)))
        return int(n_out.value)

    def set_scan_source(self, source_id: int, scan_xyz: np.ndarray, T_base_sensor, T_gbl_base, perception_window_size: float,
                        marking_height: float):
        """One of several sensors (StackedPerception::aggregateObservations, stacked_perception.cpp:128-140): returns
        (points of this sensor's observation, points of the aggregate = all sensors' latest observations in source order)."""
        scan = np.ascontiguousarray(scan_xyz, dtype=np.float32)
        if scan.ndim != 2 or (scan.shape[0] and scan.shape[1] < 3):
            raise ValueError("scan must be [P, >=3] float32")
        tbs = (C.c_double * 7)(*[float(v) for v in T_base_sensor])
        tgb = (C.c_double * 7)(*[float(v) for v in T_gbl_base])
        n_src, n_all = C.c_uint32(0), C.c_uint32(0)
        stride = scan.strides[0] if scan.shape[0] else 12
        self._check(self._lib.dddmr_rollout_set_scan_source(self._ctx, int(source_id), scan.ctypes.data_as(C.c_void_p), scan.shape[0],
                                                            stride, tbs, tgb, perception_window_size, marking_height,
                                                            C.byref(n_src), C.byref(n_all)))
        return int(n_src.value), int(n_all.value)

    def set_stitcher_source(self, source_id: int, stitcher_num: int):
        self._check(self._lib.dddmr_rollout_set_stitcher_source(self._ctx, int(source_id), int(stitcher_num)))

    def set_stitcher(self, stitcher_num: int):
        """cbSensor's `stitcher_num` (multilayer_spinning_lidar.cpp:185-200): feed the last N raw scans together."""
        self._check(self._lib.dddmr_rollout_set_stitcher(self._ctx, int(stitcher_num)))

    def get_cloud(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(self._lib.dddmr_rollout_get_cloud(self._ctx, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self._check(self._lib.dddmr_rollout_get_cloud(self._ctx, out.ctypes.data_as(C.c_void_p), out.shape[0],
                                                      C.byref(n)))
        return out[: n.value]

    def setPlan(self, prune_plan: np.ndarray):
        """Prune plan poses [M,7] (x y z qx qy qz qw), the output of
        Local_Planner::prunePlan (local_planner.cpp:374-445)."""
        plan = np.ascontiguousarray(prune_plan, dtype=np.float64).reshape(-1, 7)
        self._check(self._lib.dddmr_rollout_set_prune_plan(self._ctx, plan.ctypes.data_as(C.c_void_p), plan.shape[0]))

    set_prune_plan = setPlan

    def path_blocked(self, pcl_prune_plan: np.ndarray, check_radius: float):
        """PathBlockedStrategy::selfMark (path_blocked_strategy.cpp:56-100) on the current
        aggregate observation.  pcl_prune_plan: [M,4] x y z intensity as prunePlan fills it
        (host_logic.prune_plan_cloud).  -> (blocked ratio in percent, opinion, flags[M])"""
        plan = np.ascontiguousarray(pcl_prune_plan, dtype=np.float32).reshape(-1, 4)
        flags = np.zeros(max(len(plan), 1), dtype=np.uint8)
        ratio, op = C.c_double(0.0), C.c_int32(0)
        self._check(self._lib.dddmr_rollout_path_blocked(self._ctx, plan.ctypes.data_as(C.c_void_p), len(plan),
                                                         float(check_radius), C.byref(ratio), C.byref(op),
                                                         flags.ctypes.data_as(C.c_void_p)))
        return ratio.value, op.value, flags[: len(plan)].astype(bool)

    def samples(self, traj_gen_name: str, tick_in: K.TickInput) -> np.ndarray:
        """The theory's initialise(): the sample list a tick would roll out, [N,3] vx vy wz (host-only)."""
        n = C.c_size_t(0)
        self._check(self._lib.dddmr_rollout_samples(self._ctx, traj_gen_name.encode(), C.byref(tick_in), None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), dtype=np.float32)
        self._check(self._lib.dddmr_rollout_samples(self._ctx, traj_gen_name.encode(), C.byref(tick_in),
                                                    out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(n)))
        return out[: n.value]

    # -- the tick ----------------------------------------------------------
    def tick(self, traj_gen_name: str, tick_in: K.TickInput) -> K.RolloutResult:
        res = K.RolloutResult()
        self._check(self._lib.dddmr_rollout_tick(self._ctx, traj_gen_name.encode(), C.byref(tick_in), C.byref(res)))
        self.last_result = res
        return res

    def tick_begin(self, traj_gen_name: str, tick_in: K.TickInput) -> None:
        """Enqueue a tick and return at once (pair with tick_end)."""
        self._check(self._lib.dddmr_rollout_tick_begin(self._ctx, traj_gen_name.encode(), C.byref(tick_in)))

    def tick_end(self) -> K.RolloutResult:
        res = K.RolloutResult()
        self._check(self._lib.dddmr_rollout_tick_end(self._ctx, C.byref(res)))
        self.last_result = res
        return res

    def computeVelocityCommand(self, traj_gen_name: str, best_traj: Trajectory, tick_in: K.TickInput) -> PlannerState:
        """Local_Planner::computeVelocityCommand (local_planner.cpp:482-621), the
        section :535-587; fills best_traj like the reference does."""
        res = self.tick(traj_gen_name, tick_in)
        best_traj.xv_, best_traj.yv_, best_traj.thetav_ = res.vx, res.vy, res.wz
        best_traj.cost_ = res.best_cost
        best_traj.index = res.best_index
        return PlannerState(res.planner_state)

    def resolve(self, reduced_key: int) -> K.RolloutResult:
        res = K.RolloutResult()
        if self.last_result is not None:
            C.memmove(C.byref(res), C.byref(self.last_result), C.sizeof(res))
        self._check(self._lib.dddmr_rollout_resolve(self._ctx, C.c_int64(reduced_key), C.byref(res)))
        return res

    def winner_words(self, res: Optional[K.RolloutResult] = None):
        """This rank's two int64 words of the exact multi-rank argmin (cost bits, -index)."""
        res = res if res is not None else self.last_result
        w = (C.c_int64 * 2)()
        self._lib.dddmr_rollout_winner_words(C.byref(res), w)
        return int(w[0]), int(w[1])

    def resolve_words(self, words) -> K.RolloutResult:
        """Resolve the global winner from the min-all-reduced slot vector [2 * n_ranks] int64."""
        words = [int(v) for v in words]
        arr = (C.c_int64 * len(words))(*words)
        res = K.RolloutResult()
        if self.last_result is not None:
            C.memmove(C.byref(res), C.byref(self.last_result), C.sizeof(res))
        self._check(self._lib.dddmr_rollout_resolve_words(self._ctx, arr, len(words) // 2, C.byref(res)))
        return res

    # -- in-library RCCL exchange (multi-rank contexts) ------------------------
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * K.COMM_ID_BYTES)()
        rc = self._lib.dddmr_rollout_comm_unique_id(buf)
        if rc != K.OK:
            raise RolloutError(rc, "dddmr_rollout_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        if len(unique_id) != K.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = (C.c_uint8 * K.COMM_ID_BYTES)(*unique_id)
        self._check(self._lib.dddmr_rollout_comm_init(self._ctx, buf, rank, n_ranks))

    def comm_destroy(self):
        self._check(self._lib.dddmr_rollout_comm_destroy(self._ctx))

    def comm_ranks(self) -> int:
        """Ranks the context's communicator reports (ncclCommCount); 0 without one."""
        n = C.c_int32(0)
        self._check(self._lib.dddmr_rollout_comm_ranks(self._ctx, C.byref(n)))
        return int(n.value)

    def comm_loopback(self):
        """Single-device rehearsal of the exchange: see dddmr_rollout_comm_loopback."""
        self._check(self._lib.dddmr_rollout_comm_loopback(self._ctx))

    def comm_loopback_set_peer(self, peer_rank: int, words):
        w = (C.c_int64 * 2)(int(words[0]), int(words[1]))
        self._check(self._lib.dddmr_rollout_comm_loopback_set_peer(self._ctx, int(peer_rank), w))

    def stream_ceiling(self, nbytes: int = 1 << 30, reps: int = 10):
        """Measured stream ceilings of this GPU -> (copy GB/s counting read + write, read-only GB/s)."""
        cp, rd = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.dddmr_rollout_stream_ceiling(self._ctx, nbytes, reps, C.byref(cp), C.byref(rd)))
        return cp.value, rd.value

    def selftest_sincos(self, angles):
        """sin / cos of heading angles from the rollout's own double routine -> (sin[n], cos[n]) f64"""
        a = np.ascontiguousarray(angles, dtype=np.float64)
        sn, cs = np.empty_like(a), np.empty_like(a)
        self._check(self._lib.dddmr_rollout_selftest_sincos(self._ctx, a.ctypes.data, a.size, sn.ctypes.data, cs.ctypes.data))
        return sn, cs

    # -- per-trajectory outputs of the last tick -----------------------------
    def debug(self):
        """-> (costs[n_local] f64, steps[n_local] i32, samples[n_local,3] f32)"""
        n = int(self.last_result.n_local) if self.last_result is not None else 0
        costs = np.zeros(max(n, 1), dtype=np.float64)
        steps = np.zeros(max(n, 1), dtype=np.int32)
        smp = np.zeros((max(n, 1), 3), dtype=np.float32)
        dbg = K.RolloutDebug()
        dbg.costs = costs.ctypes.data_as(C.POINTER(C.c_double))
        dbg.steps = steps.ctypes.data_as(C.POINTER(C.c_int32))
        dbg.samples = smp.ctypes.data_as(C.POINTER(C.c_float))
        self._check(self._lib.dddmr_rollout_get_debug(self._ctx, C.byref(dbg)))
        return costs[:n], steps[:n], smp[:n]

    def pose_arrays(self, accepted_only: bool = False) -> np.ndarray:
        """The `trajectory` / `accepted_trajectory` debug pose arrays of the last tick, [n,7]."""
        which = 1 if accepted_only else 0
        n = C.c_size_t(0)
        self._check(self._lib.dddmr_rollout_get_pose_arrays(self._ctx, which, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 7), dtype=np.float64)
        self._check(self._lib.dddmr_rollout_get_pose_arrays(self._ctx, which, out.ctypes.data_as(C.c_void_p), out.shape[0],
                                                            C.byref(n)))
        return out[: n.value]

    def best_poses(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(self._lib.dddmr_rollout_get_best_poses(self._ctx, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 7), dtype=np.float64)
        self._check(self._lib.dddmr_rollout_get_best_poses(self._ctx, out.ctypes.data_as(C.c_void_p), out.shape[0],
                                                           C.byref(n)))
        return out[: n.value]

    def best_cuboids(self) -> np.ndarray:
        """Cuboid vertices carried along the best trajectory, [n_poses, 8, 3] float32."""
        n = C.c_size_t(0)
        self._check(self._lib.dddmr_rollout_get_best_cuboids(self._ctx, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 8, 3), dtype=np.float32)
        self._check(self._lib.dddmr_rollout_get_best_cuboids(self._ctx, out.ctypes.data_as(C.c_void_p), out.shape[0],
                                                             C.byref(n)))
        return out[: n.value]
