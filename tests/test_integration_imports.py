"""The import swap of INTEGRATION.md section 2 is complete: the replacement lines, executed EXACTLY as that
file prints them (this repo's own text), resolve every name the reference imports from the CUDA-only
packages -- `gsplat` at /root/reference/utils/gsplat_utils/gsplat_trainer.py:42-46 and
utils/gsplat_utils/utils.py:91, `fused_ssim` at gsplat_trainer.py:30 -- on a box that has neither."""
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the reference's import lines bind (names only; see the file:line list above)
GSPLAT_NAMES = ("PngCompression", "cli", "rasterization", "DefaultStrategy", "MCMCStrategy", "SelectiveAdam",
                "_eval_sh_bases_fast", "fused_ssim")
DATA_NAMES = ("Dataset", "Parser", "generate_interpolated_path", "generate_ellipse_path_z", "generate_spiral_path")


def _python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec2 = text[text.index("## 2."):text.index("## 3.")]
    return re.findall(r"```python\n(.*?)```", sec2, flags=re.S)


def test_replacement_imports_resolve_every_name():
    blocks = [b for b in _python_blocks() if "from splat_one_amd" in b]
    assert len(blocks) >= 3, "INTEGRATION.md section 2 lost its replacement-import blocks"
    ns = {}
    for b in blocks:
        # the commented lines are the reference's originals, kept for orientation; the rest must be importable as is
        assert "gsplat" not in re.sub(r"^#.*$", "", b, flags=re.M), "a live line still imports gsplat"
        exec(compile(b, "INTEGRATION.md", "exec"), ns)
    for name in GSPLAT_NAMES + DATA_NAMES:
        assert name in ns and ns[name] is not None, f"{name} is not bound by the documented swap"
    assert callable(ns["rasterization"]) and callable(ns["cli"]) and callable(ns["fused_ssim"])
    # the keyword surface the reference passes at gsplat_trainer.py:478-493
    params = inspect.signature(ns["rasterization"]).parameters
    for kw in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "width", "height", "packed", "absgrad",
               "sparse_grad", "rasterize_mode", "distributed", "camera_model", "sh_degree", "near_plane", "far_plane",
               "render_mode", "radius_clip"):
        assert kw in params, kw
    # strategy surface used at :129-131, 345-352, 616-622, 744-763
    for cls in (ns["DefaultStrategy"], ns["MCMCStrategy"]):
        for meth in ("check_sanity", "initialize_state", "step_pre_backward", "step_post_backward"):
            assert callable(getattr(cls, meth)), (cls, meth)
    assert "visibility" in inspect.signature(ns["SelectiveAdam"].step).parameters     # .step(visibility_mask), :728


def test_png_compression_imports_but_refuses_use():
    from splat_one_amd.compression import PngCompression
    with pytest.raises(NotImplementedError, match="compression"):
        PngCompression()
    with pytest.raises(NotImplementedError):
        PngCompression(use_sort=False, verbose=False)


def test_header_documents_the_spherical_model_consistently():
    h = open(os.path.join(ROOT, "include", "splat_one_amd.h")).read()
    assert "SO_CAM_SPHERICAL = 3" in h
    assert "unspecified -> SO_ERR_UNSUPPORTED" not in h
