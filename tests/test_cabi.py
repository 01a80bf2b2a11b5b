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
    assert lib.so_abi_version() == _lib.ABI_VERSION == 2
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


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """sizeof / offsetof of so_step_desc, so_adam_group and so_attr_shadow as gcc sees the header == the ctypes mirrors
    (a silent mismatch would hand the kernels shifted pointers)."""
    import subprocess
    fields = {"so_step_desc": ["means", "viewmats", "radii", "key_buf", "rec", "v_means", "grad2d", "isect_capacity", "abi_size",
                               "raster_impl", "eps2d", "scale_reg", "pixels_indirect", "inputs_staged", "tile_cull",
                               "overflow_flag_out", "attr_rows_f16", "tile_slots", "bin_capacity", "fuse_adam", "n_dev", "tile_order", "sort_in_rasteriser",
                               "bin_replicas", "bin_sub_counts", "bwd_seg_len", "bwd_seg_count", "bwd_seg_state", "tile_order_ready"],
              "so_adam_group": ["param", "visibility", "numel", "row_len", "lr_step_size", "bc2_sqrt"],
              "so_attr_shadow": ["arec", "stride_bytes", "offset_bytes"],
              "so_model_set": ["p", "m", "v"],
              "so_raster_desc": ["abi_size", "seq", "raster_impl", "eps2d", "radius_clip", "bin_capacity", "means", "shN", "backgrounds", "counters",
                                 "key_buf", "vrec", "status_out", "render_colors", "last_ids", "v_render_colors", "v_means",
                                 "v_shN", "v_means2d", "v_means2d_abs", "bin_sub_counts", "bin_replicas", "tile_order"],
              "so_refine_params": ["grow_grad2d", "grow_scale3d", "prune_opa", "prune_scale3d", "prune_big", "revised_opacity",
                                   "seed", "step"]}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for st, fs in fields.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            src.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src += ['  printf("SO_TILE_SLOTS %d\\nSO_ADAM_MAX_GROUPS %d\\nSO_CAM_PER_VIEW %d\\n", SO_TILE_SLOTS, SO_ADAM_MAX_GROUPS, SO_CAM_PER_VIEW);',
            "  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(c)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirrors = {"so_step_desc": _lib.StepDesc, "so_adam_group": _lib.AdamGroup, "so_attr_shadow": _lib.AttrShadow,
               "so_model_set": _lib.ModelSet, "so_refine_params": _lib.RefineParams, "so_raster_desc": _lib.RasterDesc}
    for st, fs in fields.items():
        assert int(out[st]) == ctypes.sizeof(mirrors[st]), st
        for f in fs:
            assert int(out[f"{st}.{f}"]) == getattr(mirrors[st], f).offset, (st, f)
    from splat_one_amd import ops
    assert int(out["SO_TILE_SLOTS"]) == _lib.SO_TILE_SLOTS and int(out["SO_ADAM_MAX_GROUPS"]) == _lib.SO_ADAM_MAX_GROUPS
    assert int(out["SO_CAM_PER_VIEW"]) == ops.SO_CAM_PER_VIEW


def test_per_view_camera_models_and_f16_rows_validate_without_a_gpu():
    from splat_one_amd.ops import SO_CAM_PER_VIEW, camera_model_code
    assert camera_model_code("pinhole", 3) == 0 and camera_model_code(["fisheye"] * 4, 4) == 2
    assert camera_model_code(["pinhole", "fisheye", "ortho"], 3) == SO_CAM_PER_VIEW | (0 << 0) | (2 << 2) | (1 << 4)
    assert camera_model_code("spherical", 1) == 3 and camera_model_code(["spherical", "pinhole"], 2) == SO_CAM_PER_VIEW | 3
    for bad, nv in ((["pinhole"], 2), (["pinhole", "cylindrical"], 2), (["pinhole", "fisheye"] * 8, 16)):
        with pytest.raises(AssertionError):
            camera_model_code(bad, nv)
    lib = _lib.load()
    assert [lib.so_attr_rec_stride(k) for k in (0, 1, 4, 9, 16, 25)] == [0, 32, 48, 80, 112, 176]
    # a model id that does not exist (the four 2-bit per-view values are all taken), and more views than the code can
    # carry: refused
    with pytest.raises(RuntimeError, match="camera_model"):
        _lib.call("so_preprocess_fwd", 2, 4, 16, 3, *([1] * 8), 16, 16, 0.3, 0.01, 1e8, 0.0, 4, 0, 16,
                  *([1] * 7), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="camera_model"):
        _lib.call("so_preprocess_fwd", 16, 4, 16, 3, *([1] * 8), 16, 16, 0.3, 0.01, 1e8, 0.0, SO_CAM_PER_VIEW, 0, 16,
                  *([1] * 7), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        _lib.call("so_preprocess_bwd_f16", 1, 4, 16, 3, 1, 1, 8, 1, 1, 16, 16, 0.3, 0, 0, 1, 1, 1, 0.0, 0.0, *([1] * 6), 0, 0, 64, 0, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="tile_slots"):
        _lib.call("so_preprocess_fwd", 1, 4, 16, 3, *([1] * 8), 16, 16, 0.3, 0.01, 1e8, 0.0, 0, 0, 16,
                  *([1] * 7), 0, 0, 0, 0, 1, 0, 0, 0, 0, 0)


def test_bin_counter_placement_is_a_bijection_that_separates_neighbouring_runs():
    """so_bin_counter_index (include/splat_one_amd.h): where tile t of M keeps its binned count.  A bijection of [0, M) for
    every M (any tile grid, any number of views; odd M leaves its last tile in place; a tile count that 1031 divides takes
    the other multiplier), runs of two neighbouring tiles stay together, neighbouring runs land >= 1 KB apart."""
    f = _lib.load().so_bin_counter_index
    for M in (1, 2, 3, 4, 16, 17, 510, 8160, 8161, 2 * 1031, 2 * 1031 * 1033 // 1033 * 3, 14400, 65280, 4 * 8160 + 1):
        idx = [f(t, M) for t in range(M)]
        assert sorted(idx) == list(range(M)), M
        for t in range(0, M - 1, 2):
            assert idx[t + 1] == idx[t] + 1 and idx[t] % 2 == 0, (M, t)
    idx = [f(t, 8160) for t in range(8160)]
    gaps = [abs(idx[t + 2] - idx[t]) * 4 for t in range(0, 8158, 2)]
    assert min(gaps) >= 1024, min(gaps)
    assert f(5, 1 << 23) == 5                     # beyond 2^21 runs: plain order (the 32-bit product would wrap)
