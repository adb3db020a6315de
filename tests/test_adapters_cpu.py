"""The ROS 2 adapter sources (adapters/ros2/) cannot be compiled here (no ROS 2 / PCL / pluginlib in the
image), so this checks what can be checked: every C-ABI symbol, enum and struct field they use exists in
include/dddmr_rollout.h, the plugin manifest names the classes the sources export, and -- where the
reference checkout is present -- the variant-(ii) patches apply cleanly to it."""
import os
import re
import shutil
import subprocess
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


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "dddmr_local_planner")) or shutil.which("patch") is None,
                    reason="needs the reference checkout and patch(1)")
def test_variant_ii_patches_apply_to_the_reference():
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "src", "dddmr_local_planner")
        for sub in ("local_planner", "recovery_behaviors"):
            shutil.copytree(os.path.join(REF, "src", "dddmr_local_planner", sub), os.path.join(dst, sub))
        for p in sorted(os.listdir(os.path.join(AD, "patches"))):
            r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(AD, "patches", p)], cwd=tmp,
                               capture_output=True, text=True)
            assert r.returncode == 0, (p, r.stdout, r.stderr)
            assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
