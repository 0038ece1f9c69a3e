"""(not gpu) Host code under AddressSanitizer + UndefinedBehaviorSanitizer ("Race detection /
sanitizers": build host code with -fsanitize=address,undefined).  Three targets, none of them device code:
  * the CPU emulation of the sweep kernels (tests/cpp/sweep_emulation.cc): every ring / spill / refill path,
  * the C++ host glue (csrc/host/*.cc) through tests/cpp/test_hip_host --no-device,
  * the oracle (oracle/vgt_oracle.c) under its own known-answer tests, with libasan preloaded into python."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def _libasan():
    path = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip("libasan not available")
    return os.path.realpath(path)


def test_sweep_emulation_under_sanitizers():
    _libasan()
    subprocess.check_call(["make", "-s", "-C", CPP, "sweep_emulation_asan"])
    run = subprocess.run([os.path.join(CPP, "sweep_emulation_asan"), "1"], capture_output=True, text=True, timeout=900,
                         env=ENV)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert " 0 mismatches" in run.stdout


def test_host_glue_under_sanitizers():
    _libasan()
    if not os.path.exists(os.path.join(ROOT, "voxelized_geometry_tools_amd", "libvgt_hip.so")):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "voxelized_geometry_tools_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", CPP, "test_hip_host_asan"])
    run = subprocess.run([os.path.join(CPP, "test_hip_host_asan"), "--no-device"], capture_output=True, text=True,
                         timeout=300, env=ENV)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "PASSED" in run.stdout


def test_oracle_under_sanitizers():
    asan = _libasan()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(ENV, LD_PRELOAD=asan, VGT_ORACLE_LIB=os.path.join(ROOT, "oracle", "libvgt_oracle_asan.so"),
               OMP_NUM_THREADS="4")
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_sdf.py"),
                          os.path.join(ROOT, "tests", "test_oracle_voxelization.py")],
                         capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
