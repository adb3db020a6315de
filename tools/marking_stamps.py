"""Phase breakdown of the fused marking update's two single-workgroup blocks (diagnostic build:
make -C dddmr_navigation_amd/csrc diag).
usage: DDDMR_LIB_NAME=libdddmr_rollout_diag.so python tools/marking_stamps.py [window]"""
import ctypes as C, sys
import numpy as np
from dddmr_navigation_amd import scenes, marking, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner

window = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
sc = scenes.bench_scene("C2")
lib = K.load_library()
lib.dddmr_rollout_diag_mkstamps.argtypes = [C.c_void_p, C.c_size_t]
scans = [scenes.lidar_scan(sc.cloud, seed=100 + i) for i in range(10)]
t_bs, t_gb = (0.0, 0.0, 0.5, 0, 0, 0, 1), (0.0, 0.0, 0.0, 0, 0, 0, 1)
names = {0: "start", 1: "P0 select + local cids", 2: "P1 sort 1 + starts", 3: "P2 centroids", 4: "P3 sort 2", 5: "P4 0.2 m voxels",
         6: "P5 static / FOV", 7: "P6 sort 3", 8: "P7 generator points", 9: "P8 slots"}
gn = {32: "start", 33: "zero + count", 34: "scan", 35: "scatter"}
with LocalPlanner([sc.theory], max_points=1 << 16) as lp:
    layer = marking.BenchMarking(lp, sc).layer if window == 10.0 else marking.MarkingLayer(
        lp, marking.shipped_config(perception_window_size=window), marking.ground_lattice(), np.zeros((0, 3), np.float32))
    acc = np.zeros(64); last = None
    n = 0
    for i in range(30):
        lp.set_scan(scans[i % 10], t_bs, t_gb, window, 2.0)
        st = layer.update(t_bs, t_gb)
        buf = np.zeros(192, dtype=np.uint64)
        assert lib.dddmr_rollout_diag_mkstamps(buf.ctypes.data_as(C.c_void_p), 192) == 0
        if i >= 10:
            acc += buf[:64].astype(np.float64); n += 1; last = buf[64:].copy()
    acc /= n
    print("observation", st.n_observation, "clusters", st.n_clusters, "alive", st.n_alive, layer.route_counts())
    print("partition 0 (kilo-ticks of s_memtime):")
    for i in range(1, 10):
        print(f"  {names[i]:24s} {(acc[i] - acc[i-1]) / 1000.0:9.2f}")
    print(f"  total                    {(acc[9] - acc[0]) / 1000.0:9.2f}")
    pm = (last >> np.uint64(40)).astype(np.int64); pt = (last & np.uint64((1 << 40) - 1)).astype(np.float64) / 1000.0
    order = np.argsort(-pt)
    print("partitions of the last update (points, kilo-ticks), slowest first:", [(int(pm[i]), round(float(pt[i]), 1)) for i in order[:12]],
          "... median", (int(np.median(pm[pm > 0])), round(float(np.median(pt[pm > 0])), 1)))
    print("grid block:")
    for i in range(33, 36):
        print(f"  {gn[i]:24s} {(acc[i] - acc[i-1]) / 1000.0:9.2f}")
    cyc = np.zeros(8, dtype=np.uint64)
    lib.dddmr_rollout_diag_mkclear.argtypes = [C.c_void_p]
    assert lib.dddmr_rollout_diag_mkclear(cyc.ctypes.data_as(C.c_void_p)) == 0
    c = cyc.astype(np.float64)
    r = max(c[0], 1)
    print(f"ray tests over 30 updates: {int(c[0])} rays, {c[1] / r:.2f} chunks probed per ray; kilo-ticks per ray: lay-out {c[2] / r / 1e3:.2f}, "
          f"probes {c[3] / r / 1e3:.2f}, near test {c[4] / r / 1e3:.2f}, removal {c[5] / r / 1e3:.2f}")
