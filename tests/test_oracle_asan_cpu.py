"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): the CPU suites that exercise it
(tests/test_oracle_cpu.py, test_oracle_marking_cpu.py, test_sharding_cpu.py) run in a child interpreter with libasan
preloaded and DDDMR_ORACLE_LIB=liboracle_asan.so; any report aborts the child (-fno-sanitize-recover, halt_on_error).
GPU sanitizers are not available on the pool (and not needed: the oracle is the checker)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def test_oracle_suites_are_clean_under_asan_and_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, DDDMR_ORACLE_LIB="liboracle_asan.so", PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_cpu.py",
                        "tests/test_oracle_marking_cpu.py", "tests/test_sharding_cpu.py", "-m", "not gpu"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout
