"""CPU: the C-ABI library loads, exports every symbol include/hrnet_hip.h declares, and its host-only entry points
(sizes, argument validation) behave; no kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from hrnet_hip import binding, build
    if not os.path.exists(binding.LIB_PATH):
        build.build_library(verbose=False)
    return binding.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hrnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hrn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from hrnet_hip import binding
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hrnet_hip.h but not exported"
    assert sorted(binding.SIGNATURES) == syms, "binding.SIGNATURES and the header disagree"


def test_version_and_sizes(lib):
    assert lib.hrn_version() == 1
    # packed HRNet parameters: every conv weight once, in the storage dtype (+ small f32 tensors, 256-B aligned)
    w_elems = 4 * 64 * 64 * 9 + 64 * 64 * 9 + 2 * 128 * 128 * 9 + 128 * 64 * 9 + 64 * 64 * 9
    for dt, es in ((0, 4), (1, 2), (2, 4)):       # f32, bf16, bf16x3 (two bf16 planes; fp32 decoder weights)
        n = lib.hrn_hrnet_packed_bytes(dt, 2)
        assert w_elems * es < n < w_elems * es + 64 * 1024
    assert lib.hrn_hrnet_packed_bytes(3, 2) == 0 and lib.hrn_hrnet_packed_bytes(0, 99) == 0
    # workspace: reference frame + 3 view stacks + fused state
    B, V, H = 32, 32, 128
    stack = B * V * H * H * 64 * 2
    ws = lib.hrn_hrnet_workspace_bytes(1, B, V, H, H)
    assert 3 * stack < ws < 3 * stack + B * H * H * (4 + 128) + 4096
    assert lib.hrn_hrnet_workspace_bytes(1, 0, V, H, H) == 0
    conv_bytes = 4 * 9 * (2 * 64 + 3 * 64 * 64 + 64 * 128 + 3 * 128 * 128)
    assert conv_bytes < lib.hrn_shiftnet_packed_bytes() < conv_bytes + 64 * 1024       # fc1.weight (134 MB) is read in place, not packed
    assert lib.hrn_shiftnet_workspace_bytes(4) > 2 * 4 * 128 * 128 * 64 * 4


def test_bad_arguments_fail_before_any_launch(lib):
    null = ctypes.c_void_p(0)
    rc = lib.hrn_hrnet_forward(null, 0, 2, 1, null, null, 1, 2, 8, 8, null, null, 0, null)
    assert rc == -2 and b"null" in lib.hrn_last_error()
    rc = lib.hrn_hrnet_forward(null, 7, 2, 1, null, null, 1, 2, 8, 8, null, null, 0, null)
    assert rc == -2 and b"dtype" in lib.hrn_last_error()
    rc = lib.hrn_lanczos_shift(null, null, 1, 1, 2, 2, null, null)
    assert rc == -2
    assert lib.hrn_profile_enable(0) == 0 and lib.hrn_profile_count() == 0
