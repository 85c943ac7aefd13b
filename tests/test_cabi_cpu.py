"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
include/drn_wsod.h declares (no compute calls without a GPU); the product path refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

import golden_util as G
from __graft_entry__ import build, load_package


@pytest.fixture(scope="module")
def pkg():
    return build()


def test_header_symbols_exported(pkg):
    hdr = open(os.path.join(G.ROOT, "include", "drn_wsod.h")).read()
    declared = sorted(set(re.findall(r"\b(drn_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 19
    lib = ctypes.CDLL(pkg._cabi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert sorted(pkg._cabi.exported_symbols()) == declared
    # exports == header, both ways (cross-TU helpers of drn_tune used to leak into the ABI)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", pkg._cabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("drn_")})
    assert exported == declared, (set(exported) ^ set(declared))


def test_no_cpu_fallback(pkg):
    import importlib

    ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
    with pytest.raises((AssertionError, pkg._cabi.DrnError)):
        ops.maxpool2x2_nhwc(torch.zeros(1, 4, 4, 8), 2)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    c = pkg._cabi
    monkeypatch.setattr(c, "_lib", None)
    monkeypatch.setattr(c, "LIB_PATH", "/nonexistent/libdrn_wsod_hip.so")
    with pytest.raises(c.DrnError):
        c.lib()
