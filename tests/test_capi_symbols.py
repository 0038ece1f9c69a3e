"""(not gpu) The C-ABI library loads and exports every symbol include/vgt_hip.h declares.
No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from voxelized_geometry_tools_amd import capi


def _declared_symbols(testing=False):
    """Functions include/vgt_hip.h declares: outside (product) or inside (testing=True) its VGT_HIP_TESTING section."""
    text = open(os.path.join(ROOT, "include", "vgt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    section = re.search(r"#ifdef VGT_HIP_TESTING(.*?)#endif", text, flags=re.S)
    assert section, "the header has a VGT_HIP_TESTING section"
    text = section.group(1) if testing else text.replace(section.group(0), "")
    return sorted(set(re.findall(r"\b(vgt_hipx?_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi.load()


def test_header_and_binding_agree(lib):
    declared = _declared_symbols()
    assert len(declared) >= 30
    assert sorted(capi.SIGNATURES) == declared


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), name


def test_testing_hooks_live_in_the_testing_library_only(lib):
    """The cross-check EDT variant and the debug / tuning hooks are declared under VGT_HIP_TESTING, bound separately and
    exported by libvgt_hip_testing.so -- which also exports the whole product ABI -- and by nothing else."""
    hooks = _declared_symbols(testing=True)
    assert sorted(capi.TESTING_SIGNATURES) == hooks and len(hooks) >= 3
    product = ctypes.CDLL(capi.LIB_PATH)
    testing = ctypes.CDLL(capi.TESTING_LIB_PATH)
    for name in hooks:
        assert hasattr(testing, name), name
        assert not hasattr(product, name), name
    for name in _declared_symbols():
        assert hasattr(testing, name), name
    # no environment knobs in the product library
    with open(capi.LIB_PATH, "rb") as fh:
        assert b"VGT_HIP_HOST_PIPELINE" not in fh.read()


def _exported(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_dynamic_symbol_table_is_the_c_abi_and_nothing_else(lib):
    """-fvisibility=hidden + csrc/exports.map: a drop-in library for someone else's process exports the functions of
    include/vgt_hip.h and nothing else -- no vgt:: internals, no bare helper names, no libstdc++ instantiations."""
    product = _exported(capi.LIB_PATH)
    assert product == _declared_symbols(), sorted(set(product) ^ set(_declared_symbols()))
    testing = _exported(capi.TESTING_LIB_PATH)
    assert testing == sorted(_declared_symbols() + _declared_symbols(testing=True))


def test_abi_version_and_workspace_size(lib):
    assert lib.vgt_hip_abi_version() == 2
    # class records (16 bytes per 64-voxel word of a Z line, plus 256 records of padding) + the int32 intermediate, a
    # small min/max block and the line passes' scratch (8 work counters + per workgroup in flight -- at most 4096 -- the
    # spill area of a full-depth stack per lane, chunks of 8 x 4-byte entries where the extents allow packed entries,
    # and one 8-byte word record per 32 rows and lane), 256-byte aligned pieces; the cross-check variant keeps an int16
    # distance field instead of the records
    n = 64 * 64 * 64
    slots, words = 64, 2
    narrow = ((64 + 4 + 7) // 8 + 1) * 64 * 8 * 4
    # (the scratch's head: 8 work counters, a cache line each; per slot one spare row of 64 x 8 bytes behind the sign words)
    head = 8 * 128
    scratch = head + slots * (narrow + (words + 1) * 64 * 8) + 256
    records = (64 * 1 * 64 + 256) * 16
    assert capi.sdf_workspace_bytes((64, 64, 64)) == records + n * 4 + 256 + scratch
    assert capi.sdf_workspace_bytes((64, 64, 64), 1) == n * 2 + n * 4 + 256 + scratch
    assert capi.sdf_workspace_bytes((64, 64, 64), 2) == 0  # (earlier rounds' other cross-check pipelines are gone)
    # the scratch grows with the axis lengths, not with the volume: at most 5120 workgroups are in flight
    big = capi.sdf_workspace_bytes((1024, 1024, 1024))
    assert big - 4 * 2 ** 30 - 2 ** 28 - 256 < 1.25 * 2 ** 30
    assert capi.sdf_workspace_bytes((0, 4, 4)) == 0


def test_argument_errors_without_device(lib):
    """Null / invalid arguments are rejected before any HIP call."""
    assert lib.vgt_hip_create(0, -1, None) == 1
    assert b"null" in lib.vgt_hip_last_error()
    assert lib.vgt_hip_device_count(None) == 1
    assert lib.vgt_hip_tracking_grids_create(None, 10, 1, None) == 1
