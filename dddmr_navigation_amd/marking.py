"""Host-side mirror of the global-mode marking / clearing layer on top of the C-ABI
(dddmr_rollout_marking_*): MultiLayerSpinningLidar::selfClear / selfMark with is_local_planner =
false and its Marking store + dGraph
(/root/reference/src/dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:306-628,
plugins/cluster_marking.cpp:49-138).  All compute happens in the HIP library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


def shipped_config(**kw) -> K.MarkingConfig:
    """perception_3d_global / lidar block of
    dddmr_p2p_move_base/config/p2p_move_base_localization.yaml:286-334."""
    c = K.MarkingConfig()
    d = dict(xy_resolution=0.05, height_resolution=0.05, marking_height=2.0, perception_window_size=5.0,
             vertical_FOV_top=15.0, vertical_FOV_bottom=-15.0,
             scan_effective_positive_start=30.0, scan_effective_positive_end=180.0,
             scan_effective_negative_start=-30.0, scan_effective_negative_end=-180.0,
             euclidean_cluster_extraction_tolerance=0.1, euclidean_cluster_extraction_min_cluster_size=1,
             segmentation_ignore_ratio=1.1, inscribed_radius=0.5, inflation_radius=1.5,
             max_obstacle_distance=9999.0, max_markings=1 << 15, max_cluster_points=1 << 20)
    d.update(kw)
    for k, v in d.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class MarkingLayer:
    """The marking layer of one LocalPlanner context.  `ground` = pcl_ground_ (dGraph nodes),
    `static_map` = pcl_map_ (static layer cloud), both [N, >=3] float32."""

    def __init__(self, lp, cfg: K.MarkingConfig, ground: np.ndarray, static_map: np.ndarray):
        self._lp = lp
        self.cfg = cfg
        g = np.ascontiguousarray(ground, dtype=np.float32)
        m = np.ascontiguousarray(static_map, dtype=np.float32).reshape(-1, max(3, static_map.shape[1] if static_map.ndim == 2 else 3))
        self.n_ground = len(g)
        lp._check(lp._lib.dddmr_rollout_marking_create(
            lp._ctx, C.byref(cfg), g.ctypes.data_as(C.c_void_p), len(g), g.strides[0] if len(g) else 12,
            m.ctypes.data_as(C.c_void_p), len(m), m.strides[0] if len(m) else 12))
        self.last = None
        self.totals = dict(updates=0, clusters=0, marked=0, cleared=0, clear_ms=0.0, mark_ms=0.0)

    def update(self, T_base_sensor, T_gbl_base) -> K.MarkingStats:
        """One doClear_then_Mark pass on the context's current aggregate observation."""
        tbs = (C.c_double * 7)(*[float(v) for v in T_base_sensor])
        tgb = (C.c_double * 7)(*[float(v) for v in T_gbl_base])
        st = K.MarkingStats()
        self._lp._check(self._lp._lib.dddmr_rollout_marking_update(self._lp._ctx, tbs, tgb, C.byref(st)))
        self.last = st
        t = self.totals
        t["updates"] += 1; t["clusters"] += st.n_clusters; t["marked"] += st.n_marked; t["cleared"] += st.n_cleared
        t["clear_ms"] += st.clear_ms; t["mark_ms"] += st.mark_ms
        return st

    def reset(self):
        self._lp._check(self._lp._lib.dddmr_rollout_marking_reset(self._lp._ctx))

    def route_counts(self) -> dict:
        """Updates that ran fused (four launches) / on the general route, launches of the last update."""
        a, b, n = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_route_counts(self._lp._ctx, C.byref(a), C.byref(b), C.byref(n)))
        return {"fused": int(a.value), "general": int(b.value), "launches_last_update": int(n.value)}

    def points(self, with_voxels: bool = False):
        """Generator points of the alive markings (projected on the ground plane, 0.1 m VoxelGrid), [n, 3] float32;
        with_voxels: also the voxel key of the marking each point belongs to, [n, 3] int32."""
        n = C.c_size_t(0)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_points(self._lp._ctx, None, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), dtype=np.float32)
        vox = np.zeros((max(n.value, 1), 3), dtype=np.int32)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_points(self._lp._ctx, out.ctypes.data_as(C.c_void_p), vox.ctypes.data_as(C.c_void_p),
                                                                       out.shape[0], C.byref(n)))
        return (out[: n.value], vox[: n.value]) if with_voxels else out[: n.value]

    def voxels(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_voxels(self._lp._ctx, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), dtype=np.int32)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_voxels(self._lp._ctx, out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(n)))
        return out[: n.value]

    def dgraph(self) -> np.ndarray:
        out = np.zeros(self.n_ground + 1, dtype=np.float64)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_dgraph(self._lp._ctx, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def lethal(self) -> np.ndarray:
        out = np.zeros(self.n_ground + 1, dtype=np.uint8)
        self._lp._check(self._lp._lib.dddmr_rollout_marking_get_lethal(self._lp._ctx, out.ctypes.data_as(C.c_void_p), out.size))
        return out.astype(bool)

    def summary(self) -> dict:
        t = self.totals
        n = max(t["updates"], 1)
        return {"updates": t["updates"], "clusters_per_update": round(t["clusters"] / n, 1),
                "marked_per_update": round(t["marked"] / n, 1), "cleared_per_update": round(t["cleared"] / n, 1),
                "alive_markings": int(self.last.n_alive) if self.last is not None else 0,
                "clear_ms": round(t["clear_ms"] / n, 4), "mark_ms": round(t["mark_ms"] / n, 4),
                "route": self.route_counts()}

    def close(self):
        pass      # the context owns the device state


def ground_lattice(half: float = 10.0, spacing: float = 0.25, seed: int = 7) -> np.ndarray:
    """A synthetic mapground: floor nodes on a jittered lattice at z ~ 0 (the reference's ground cloud is a
    voxel-downsampled floor scan of about this density)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xs = np.arange(-half, half + 1e-6, spacing)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    pts += rng.uniform(-0.02, 0.02, size=pts.shape) * np.array([1, 1, 0.25])
    return pts.astype(np.float32)


def bench_layer(lp, sc) -> "BenchMarking":
    return BenchMarking(lp, sc)


class BenchMarking:
    """bench.py --workload C5M: the marking layer fed by the same scans as the local feed (sensor
    0.5 m above base_link at the origin), static map = the scene's walls."""

    def __init__(self, lp, sc):
        cloud = sc.cloud
        walls = cloud[(np.abs(np.abs(cloud[:, 1]) - 9.9) < 0.05)]
        self.layer = MarkingLayer(lp, shipped_config(perception_window_size=10.0), ground_lattice(), walls)

    def update(self, scan, T_base_sensor, T_gbl_base):
        return self.layer.update(T_base_sensor, T_gbl_base)

    def summary(self):
        return self.layer.summary()

    def close(self):
        self.layer.close()
