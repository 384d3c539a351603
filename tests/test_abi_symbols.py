"""The C-ABI library loads and exports every symbol include/mp2g.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest


def declared_symbols(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp2g_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(mp2):
    assert os.path.exists(mp2.LIB_PATH), "libmp2gpu.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(mp2.LIB_PATH)
    names = declared_symbols(mp2.HEADER_PATH)
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/mp2g.h but not exported: {missing}"


def test_no_cpu_fallback(mp2):
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mp2.Mp2gError):
        mp2.Context(0)


def test_product_does_not_reference_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "mapreduce-plonky2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".cuh", ".py", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text, f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', text), f
                assert not re.search(r'^\s*(import|from)\s+oracle', text, flags=re.M), f
                assert "orc_" not in text, f
