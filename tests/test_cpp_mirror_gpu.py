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
