"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/sppm_abi.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import pytest

from raytrace3_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sppm_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sppm_[a-z_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    assert _declared_symbols() == sorted(abi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(abi.LIB_PATH):
        pytest.fail(f"{abi.LIB_PATH} not built: run __graft_entry__.build()")
    lib = C.CDLL(abi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_struct_sizes_match_the_header():
    # sizes implied by include/sppm_abi.h on LP64
    assert C.sizeof(abi.Material) == 11 * 8
    assert C.sizeof(abi.Object) == 8 + 24 + 88 + 24 + 24 + 8 + 8
    assert C.sizeof(abi.Camera) == 12 * 8 + 8
    assert C.sizeof(abi.HitPoint) == 16 * 8 + 6 * 4
    assert C.sizeof(abi.Collision) == 10 * 8 + 8
    assert C.sizeof(abi.Counters) == 8 * 8


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.SppmError):
        abi.SppmContext(0)
