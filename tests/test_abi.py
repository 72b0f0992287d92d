"""CPU tests of the boundary: the C-ABI library loads and exports every symbol that
include/pantax_hip.h declares; without a GPU init fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "pantax_amd", "lib", "libpantax_hip.so")):
        ge.build()
    from pantax_amd import _ffi
    return _ffi.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pantax_hip.h")).read()
    declared = set(re.findall(r"\b(pantax_hip_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 15
    from pantax_amd import _ffi
    assert declared == set(_ffi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_init_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = lib.pantax_hip_init(C.byref(ctx), (C.c_int * 1)(0), 1)
    assert rc < 0 and not ctx.value
    msg = lib.pantax_hip_last_error(None).decode()
    assert "no CPU path" in msg or "HIP" in msg or "device" in msg


def test_product_does_not_touch_oracle():
    """The product path may not import/link the oracle (only tests, smoke, bench cpu_baseline may)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pantax_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "synth.py" and "oracle" not in txt, (dirpath, f)
