"""(not gpu) The lane-per-line sweep kernels (csrc/edt_sweep_kernels.hip, the default EDT line passes) compiled by g++ against a host
stand-in for the HIP runtime (tests/cpp/hip_shim) and run one lane at a time on random lines, against a brute-force line
transform: tests/cpp/sweep_emulation.cc.  Covers every ring / band size the kernels can be built with, so that a change
of the tuning constants cannot silently break the ring / spill bookkeeping (the GPU tests pin the shipped build only)."""
import os
import subprocess

import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")
CONFIGS = [
    ("default", ""),
    ("b16_r16_c4", "-DVGT_SWEEP_BAND=16 -DVGT_SWEEP_RING=16 -DVGT_SWEEP_CHUNK=4"),
    ("b8_r16_c4", "-DVGT_SWEEP_BAND=8 -DVGT_SWEEP_RING=16 -DVGT_SWEEP_CHUNK=4"),
    ("b16_r32_c8", "-DVGT_SWEEP_BAND=16 -DVGT_SWEEP_RING=32 -DVGT_SWEEP_CHUNK=8"),
    ("b8_r32_c8", "-DVGT_SWEEP_BAND=8 -DVGT_SWEEP_RING=32 -DVGT_SWEEP_CHUNK=8"),
    ("b32_wide_r32_c8", "-DVGT_SWEEP_BAND=32 -DVGT_SWEEP_RING_WIDE=32 -DVGT_SWEEP_CHUNK_WIDE=8"),
]


@pytest.mark.parametrize("name,flags", CONFIGS)
def test_sweep_kernels_on_cpu(name, flags):
    out = "sweep_emulation_" + name
    subprocess.check_call(["make", "-s", "-C", CPP, out, "SWEEP_OUT=" + out, "SWEEP_FLAGS=" + flags])
    run = subprocess.run([os.path.join(CPP, out), "2"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    assert " 0 mismatches" in run.stdout
