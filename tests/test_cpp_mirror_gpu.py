"""The C++ host mirror (include/dddmr_rollout.hpp) over the C-ABI, driven through
the playground scenario by a plain g++ program, against the oracle."""
import os
import subprocess

import pytest

from dddmr_navigation_amd import scenes, _capi as K
import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("goal", [(3.0, 1.0), (3.0, -1.0)])
def test_cpp_playground_matches_oracle(tmp_path, goal):
    exe = str(tmp_path / "playground")
    libdir = os.path.join(ROOT, "dddmr_navigation_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "playground_main.cpp"), "-o", exe,
                           "-L", libdir, "-ldddmr_rollout", f"-Wl,-rpath,{libdir}"])
    out = subprocess.check_output([exe, str(goal[0]), str(goal[1])], text=True).split("\n")
    st, idx, cost, vx, vy, wz, n, nposes = out[0].split()
    sc = scenes.playground_scene(goal, 5.0)
    o = oracle.tick(sc.theory, sc.cloud, sc.plan, sc.tick)
    r = o.result
    assert int(st) == K.TRAJECTORY_FOUND == r.planner_state
    assert int(idx) == r.best_index and int(n) == 55
    assert abs(float(cost) - r.best_cost) <= 1e-4
    assert abs(float(vx) - r.vx) <= 1e-4 and abs(float(vy) - r.vy) <= 1e-4 and abs(float(wz) - r.wz) <= 1e-4
    assert int(nposes) == int(o.steps[r.best_index])
    tag, ratio, op = out[1].split()
    assert tag == "blocked" and float(ratio) == 25.0 and int(op) == K.OPINION_PATH_BLOCKED_WAIT
    assert out[2].strip() == f"error {K.ERR_UNKNOWN_THEORY}"
    # in-library RCCL exchange from plain C++ (1-rank communicator): same winner, exact cost, host-side words agree
    # (RCCL prints a version banner on stdout when its first communicator comes up: pick the lines by tag)
    tagged = {l.split()[0]: l.split() for l in out if l.strip()}
    tag, cidx, ccost, ridx, rcost, nsmp = tagged["comm"]
    assert tag == "comm" and int(cidx) == int(ridx) == r.best_index and int(nsmp) == 55
    assert float(ccost) == float(rcost) == float(cost)
    # marking / clearing layer from plain C++ against the oracle on the same inputs
    from dddmr_navigation_amd import marking
    import numpy as np
    g = np.array([[-5 + 0.25 * i, -5 + 0.25 * j, 0.0] for i in range(41) for j in range(41)], dtype=np.float32)
    blob = lambda cx, cy: np.array([[np.float32(cx) + np.float32(0.03) * a, np.float32(cy) + np.float32(0.03) * b, np.float32(0.1) * z]
                                    for z in range(2, 10) for a in (-1, 1) for b in (-1, 1)], dtype=np.float32)
    mo = oracle.MarkingOracle(marking.shipped_config(euclidean_cluster_extraction_tolerance=0.25, max_markings=1024,
                                                     max_cluster_points=1 << 16), g, np.zeros((0, 3), np.float32))
    tbs, tgb = (0, 0, 0.5, 0, 0, 0, 1), (0, 0, 0, 0, 0, 0, 1)
    s1 = mo.update(np.concatenate([blob(0, 2), blob(-3, 3)]), tbs, tgb)
    touched, lethal = int((mo.dgraph() < 9999.0).sum()), int(mo.lethal().sum())
    mo.update(blob(-3, 3), tbs, tgb)
    s3 = mo.update(blob(-3, 3), tbs, tgb)
    tag, *vals = tagged["marking"]
    assert tag == "marking"
    assert [int(v) for v in vals] == [s1.n_clusters, s1.n_marked, s1.n_alive, touched, lethal, s3.n_cleared, s3.n_alive,
                                      len(mo.voxels())]
    assert s1.n_marked == 2 and s3.n_cleared == 1 and s3.n_alive == 1
