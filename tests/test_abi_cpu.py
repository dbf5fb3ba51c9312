"""CPU-side checks: the C-ABI library loads and exports every symbol include/vlm_hip.h declares."""
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vlm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vlm_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    import __graft_entry__ as ge
    L = importlib.import_module("vl_merging_amd._lib")
    if not os.path.exists(L.LIB_PATH):
        ge.build()
    lib = L.get_lib()
    syms = declared_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(lib, s), "library does not export " + s
        assert s in L.SIGNATURES, "ctypes binding lacks " + s
    assert lib.vlm_abi_version() == 10


def test_cpu_tensors_are_rejected(pkg):
    import torch
    merge = importlib.import_module("vl_merging_amd.merge")
    L = importlib.import_module("vl_merging_amd._lib")
    with pytest.raises(L.VlmError):
        merge.MergePlan("cpu")
    with pytest.raises(L.VlmError):
        L.require_cuda(torch.zeros(1))
