"""The ROS 2 adapter sources (adapters/ros2/) cannot be compiled here (no ROS 2 / PCL / pluginlib in the
image), so this checks what can be checked: every C-ABI symbol, enum and struct field they use exists in
include/dddmr_rollout.h, the plugin manifest names the classes the sources export, and -- where the
reference checkout is present -- the variant-(ii) patches apply cleanly to it."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

AD = os.path.join(ROOT, "adapters", "ros2")
REF = "/root/reference"


def _sources():
    out = {}
    for d, _, files in os.walk(AD):
        for f in files:
            if f.endswith((".h", ".cpp", ".patch")):
                out[os.path.join(d, f)] = open(os.path.join(d, f)).read()
    return out


def test_adapter_sources_only_use_declared_abi():
    header = open(os.path.join(ROOT, "include", "dddmr_rollout.h")).read()
    declared = set(re.findall(r"\b(dddmr_rollout_[a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    types = set(re.findall(r"\b(dddmr_[a-z_]+)\b", header))
    macros = set(re.findall(r"\b(DDDMR_[A-Z_0-9]+)\b", header))
    srcs = _sources()
    assert len(srcs) >= 8
    for path, text in srcs.items():
        added = "\n".join(l[1:] for l in text.split("\n") if l.startswith("+")) if path.endswith(".patch") else text
        for call in set(re.findall(r"\b(dddmr_rollout_[a-z_]+)\s*\(", added)):
            assert call in declared, (path, call)
        for name in set(re.findall(r"\b(dddmr_[a-z_]+)\b", added)):
            if name.startswith("dddmr_rollout_adapter") or name in ("dddmr_sys_core", "dddmr_navigation", "dddmr_local_planner",
                                                                     "dddmr_navigation_amd", "dddmr_p", "dddmr_rollout"):
                continue
            assert name in types or name in declared, (path, name)
        for m in set(re.findall(r"\b(DDDMR_[A-Z_0-9]+)\b", added)):
            if m.startswith("DDDMR_ROLLOUT_ADAPTER") or m in ("DDDMR_ROLLOUT_ROOT", "DDDMR_ROLLOUT_LIB"):
                continue
            assert m in macros, (path, m)
    # struct fields the adapters fill exist in the header
    bridge = srcs[os.path.join(AD, "dddmr_rollout_adapter", "src", "gpu_rollout_theory.cpp")]
    for field in set(re.findall(r"config_\.([a-z_]+)", bridge)):
        assert re.search(r"\b%s\b" % field, header), field


def test_plugin_manifest_matches_exported_classes():
    xml = open(os.path.join(AD, "dddmr_rollout_adapter", "plugins.xml")).read()
    srcs = _sources()
    exported = set()
    for path, text in srcs.items():
        if not path.endswith(".patch"):       # (patches quote the reference's own export lines as context)
            exported |= set(re.findall(r"PLUGINLIB_EXPORT_CLASS\((\S+?),", text))
    assert exported == set(re.findall(r'class type="([^"]+)"', xml)) and len(exported) == 2
    assert "trajectory_generators::TrajectoryGeneratorTheory" in xml and "mpc_critics::ScoringModel" in xml


def _patched_tree(tmp):
    """A copy of the reference packages the patches touch, with every patch applied."""
    for pkg, subs in (("dddmr_local_planner", ("local_planner", "recovery_behaviors")), ("dddmr_perception_3d", ("include", "plugins", "src"))):
        for sub in subs:
            shutil.copytree(os.path.join(REF, "src", pkg, sub), os.path.join(tmp, "src", pkg, sub))
    out = []
    for p in sorted(os.listdir(os.path.join(AD, "patches"))):
        if not p.endswith(".patch"):
            continue
        r = subprocess.run(["patch", "-p1", "-i", os.path.join(AD, "patches", p)], cwd=tmp, capture_output=True, text=True)
        out.append((p, r))
    return out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "dddmr_local_planner")) or shutil.which("patch") is None,
                    reason="needs the reference checkout and patch(1)")
def test_variant_ii_patches_apply_to_the_reference():
    """Planner (local_planner.cpp:535-587), second caller (rotate_inplace_behavior.cpp:224-259) and the perception side
    (multilayer_spinning_lidar.cpp cbSensor / selfClear / selfMark / resetdGraph / get_dGraphValue / updateLethalPointCloud,
    path_blocked_strategy.cpp selfMark): all four patches apply without fuzz, and the committed files are what
    make_patches.py generates from this checkout."""
    with tempfile.TemporaryDirectory() as tmp:
        res = _patched_tree(tmp)
        assert len(res) == 4
        for p, r in res:
            assert r.returncode == 0, (p, r.stdout, r.stderr)
            assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
        lidar = open(os.path.join(tmp, "src/dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp")).read()
        for call in ("feedScan(", "gpu_marking_.clearThenMark(", "gpu_marking_.create(", "gpu_marking_.dGraphValue(", "gpu_marking_.lethalPointCloud("):
            assert call in lidar, call
        assert "dddmr_rollout_adapter::pathBlocked(" in open(os.path.join(tmp, "src/dddmr_perception_3d/plugins/path_blocked_strategy.cpp")).read()
        planner = open(os.path.join(tmp, "src/dddmr_local_planner/local_planner/src/local_planner.cpp")).read()
        assert "dddmr_sys_core::PERCEPTION_MALFUNCTION" in planner.split("rolloutTick(")[1].split("auto t_diff")[0]
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(AD, "patches"), os.path.join(tmp, "patches"))
        r = subprocess.run([sys.executable, os.path.join(tmp, "patches", "make_patches.py"), REF], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for p in os.listdir(os.path.join(AD, "patches")):
            if p.endswith(".patch"):
                assert open(os.path.join(tmp, "patches", p)).read() == open(os.path.join(AD, "patches", p)).read(), p + " is stale"


def test_bridges_compile_and_behave_without_ros():
    """planner_bridge.h / perception_bridge.h / shared_context.h are templates over the ROS / PCL types: compiled here
    with stand-in types against a fake C-ABI (tests/cpp/adapter_bridge_test.cpp) and run -- a rejected observation never
    lets the tick run, a failed tick is an engine error (not "all trajectories fail"), a device feed is not overwritten."""
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "adapter_bridge_test")
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                            "-I", os.path.join(AD, "dddmr_rollout_adapter", "include"),
                            os.path.join(ROOT, "tests", "cpp", "adapter_bridge_test.cpp"), "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and "adapter bridges OK" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "dddmr_local_planner")) or shutil.which("patch") is None or shutil.which("g++") is None,
                    reason="needs the reference checkout, patch(1) and g++")
def test_adapter_sources_and_patched_files_pass_a_syntax_check():
    """g++ -fsyntax-only against the stand-in headers of tests/stubs/ (SURVEY.md 7): the three adapter sources and the
    four reference files as the patches leave them."""
    stubs = os.path.join(ROOT, "tests", "stubs")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(REF, "src", "dddmr_sys_core", "include"), os.path.join(tmp, "src", "dddmr_sys_core", "include"))
        for sub in ("trajectory_generators", "mpc_critics", "base_trajectory"):
            shutil.copytree(os.path.join(REF, "src", "dddmr_local_planner", sub, "include"), os.path.join(tmp, "src", "dddmr_local_planner", sub, "include"))
        for p, r in _patched_tree(tmp):
            assert r.returncode == 0, (p, r.stdout, r.stderr)
        src = os.path.join(tmp, "src")
        inc = ["-I", stubs, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(AD, "dddmr_rollout_adapter", "include")]
        for d in ("dddmr_local_planner/trajectory_generators", "dddmr_local_planner/mpc_critics", "dddmr_local_planner/base_trajectory",
                  "dddmr_sys_core", "dddmr_perception_3d", "dddmr_local_planner/local_planner", "dddmr_local_planner/recovery_behaviors"):
            inc += ["-I", os.path.join(src, d, "include")]
        files = [os.path.join(AD, "dddmr_rollout_adapter", "src", f) for f in sorted(os.listdir(os.path.join(AD, "dddmr_rollout_adapter", "src")))]
        files += [os.path.join(src, f) for f in ("dddmr_local_planner/local_planner/src/local_planner.cpp",
                                                  "dddmr_local_planner/recovery_behaviors/behaviors/rotate_inplace_behavior.cpp",
                                                  "dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp",
                                                  "dddmr_perception_3d/plugins/path_blocked_strategy.cpp")]
        assert len(files) == 7
        procs = [(f, subprocess.Popen(["g++", "-std=c++17", "-fsyntax-only", "-w"] + inc + [f], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
                 for f in files]
        for f, pr in procs:
            _, err = pr.communicate()
            assert pr.returncode == 0, (f, err[-3000:])
        # the check can fail: a call the patched planner does not have is caught
        bad = os.path.join(tmp, "bad.cpp")
        open(bad, "w").write(open(files[3]).read().replace("dddmr_rollout_adapter::rolloutTick(", "dddmr_rollout_adapter::rolloutTock(", 1))
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w"] + inc + [bad], capture_output=True, text=True)
        assert r.returncode != 0 and "rolloutTock" in r.stderr
