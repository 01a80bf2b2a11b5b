"""The C-ABI library loads without a GPU and exports every symbol include/splat_one_amd.h declares;
argument validation fails loudly with a status code and a message before anything is launched."""
import ctypes
import os
import re

import pytest

from splat_one_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "splat_one_amd.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(so_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 16, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # and the Python binding table covers exactly the declared entry points
    assert sorted(_lib.exported_symbols()) == names


def test_abi_version_and_error_channel():
    lib = _lib.load()
    assert lib.so_abi_version() == 1
    # invalid arguments are rejected before any HIP call (works without a GPU)
    with pytest.raises(RuntimeError, match="degrees_to_use"):
        _lib.call("so_sh_fwd", 1, 4, 16, 9, 0, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="camera_model"):
        _lib.call("so_projection_fwd", 1, 4, 0, 0, 0, 0, 0, 0, 16, 16, 0.3, 0.01, 1e8, 0.0, 7, 0, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="tile_size"):
        _lib.call("so_rasterize_fwd", 1, 4, 3, 16, 16, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="channel count"):
        _lib.call("so_rasterize_fwd", 1, 0, 7, 16, 16, 16, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0)
    with pytest.raises(RuntimeError, match="n_groups"):
        _lib.call("so_adam_step", 99, None, 0.9, 0.999, 1e-15, 0, 0)
    assert b"n_groups" in lib.so_last_error()


def test_product_has_no_cpu_path():
    """Tensors on the CPU are refused: the HIP path is the only path."""
    import torch
    from splat_one_amd.ops import fully_fused_projection
    with pytest.raises(RuntimeError, match="HIP device"):
        fully_fused_projection(torch.zeros(2, 3), None, torch.ones(2, 4), torch.ones(2, 3),
                               torch.eye(4)[None], torch.eye(3)[None], 16, 16)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "splat_one_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt or f.endswith((".md",)), f
