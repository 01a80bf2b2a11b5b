"""ctypes binding + autograd wrapper for oracle/c/raster_oracle.c.  TEST INFRASTRUCTURE ONLY.

`raster_fn(dtype)` returns a drop-in for torch_oracle.rasterize_to_pixels that runs the C
restatement (forward and its hand-written backward), so the float64 torch oracle can cover
1920x1080 with millions of intersections: projection / SH / binning stay torch+autograd, only
the per-pixel loops move to C.  tests/test_oracle_c.py pins the C code against the pure-torch
path (whose gradients come from autograd) on small cases.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


# SPLAT_ONE_AMD_SANITIZE=1 (tests/test_sanitizers.py, in a child process started with LD_PRELOAD=libasan): load the
# -fsanitize=address,undefined builds of the same source
_SAN = os.environ.get("SPLAT_ONE_AMD_SANITIZE") == "1"


def build(quiet: bool = True) -> None:
    subprocess.run(["make", "-C", _HERE] + (["sanitize"] if _SAN else []) + (["-s"] if quiet else []), check=True)


def _lib(dtype: torch.dtype):
    name = {torch.float64: "liboracle_f64.so", torch.float32: "liboracle_f32.so"}[dtype]
    if _SAN:
        name = name.replace(".so", "_san.so")
    if name not in _LIBS:
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "c", "raster_oracle.c")):
            build()
        lib = ctypes.CDLL(path)
        assert lib.oracle_real_size() == torch.empty(0, dtype=dtype).element_size()
        _LIBS[name] = lib
    return _LIBS[name]


def max_threads() -> int:
    return int(_lib(torch.float64).oracle_max_threads())


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class _RasterC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height, tile_size,
                isect_offsets, flatten_ids, want_abs, periodic=False):
        dtype = means2d.dtype
        lib = _lib(dtype)
        C, N = opacities.shape
        D = colors.shape[-1]
        th, tw = isect_offsets.shape[1:]
        means2d, conics, colors, opacities = (t.contiguous() for t in (means2d, conics, colors, opacities))
        bg = None if backgrounds is None else backgrounds.to(dtype).contiguous()
        off = isect_offsets.to(torch.int32).contiguous()
        fid = flatten_ids.to(torch.int32).contiguous()
        rc = torch.empty(C, height, width, D, dtype=dtype)
        ra = torch.empty(C, height, width, 1, dtype=dtype)
        last = torch.empty(C, height, width, dtype=torch.int32)
        r = lib.oracle_rasterize_fwd(C, N, D, width, height, tile_size, tw, th, int(periodic), _p(means2d), _p(conics),
                                     _p(colors), _p(opacities), _p(bg), _p(off), _p(fid),
                                     ctypes.c_int64(fid.numel()), _p(rc), _p(ra), _p(last))
        assert r == 0
        ctx.save_for_backward(means2d, conics, colors, opacities, bg if bg is not None else torch.empty(0),
                              off, fid, ra, last)
        ctx.dims = (C, N, D, width, height, tile_size, tw, th, bg is not None, want_abs, int(periodic))
        ctx.absgrad = None
        return rc, ra, last

    @staticmethod
    def backward(ctx, v_rc, v_ra, _v_last):
        means2d, conics, colors, opacities, bg, off, fid, ra, last = ctx.saved_tensors
        C, N, D, width, height, tile_size, tw, th, has_bg, want_abs, periodic = ctx.dims
        dtype = means2d.dtype
        lib = _lib(dtype)
        v_rc = v_rc.to(dtype).contiguous()
        v_ra = v_ra.to(dtype).contiguous()
        v_m = torch.zeros_like(means2d)
        v_abs = torch.zeros_like(means2d) if want_abs is not None else None
        v_cn = torch.zeros_like(conics)
        v_col = torch.zeros_like(colors)
        v_op = torch.zeros_like(opacities)
        r = lib.oracle_rasterize_bwd(C, N, D, width, height, tile_size, tw, th, periodic, _p(means2d), _p(conics),
                                     _p(colors), _p(opacities), _p(bg if has_bg else None), _p(off), _p(fid),
                                     ctypes.c_int64(fid.numel()), _p(ra), _p(last), _p(v_rc), _p(v_ra),
                                     _p(v_m), _p(v_abs), _p(v_cn), _p(v_col), _p(v_op))
        assert r == 0
        if want_abs is not None:
            want_abs.append(v_abs)
        v_bg = None
        if has_bg and ctx.needs_input_grad[4]:
            v_bg = ((1.0 - ra) * v_rc).sum(dim=(1, 2))
        return v_m, v_cn, v_col, v_op, v_bg, None, None, None, None, None, None, None


def rasterize_to_pixels(means2d, conics, colors, opacities, width, height, tile_size, isect_offsets,
                        flatten_ids, backgrounds=None, absgrad_out: Optional[list] = None,
                        return_last_ids: bool = False, periodic: bool = False):
    """Same contract as torch_oracle.rasterize_to_pixels.  If `absgrad_out` is a list, the
    backward appends v_means2d_abs[C,N,2] to it."""
    rc, ra, last = _RasterC.apply(means2d, conics, colors, opacities, backgrounds, width, height,
                                  tile_size, isect_offsets, flatten_ids, absgrad_out, periodic)
    if return_last_ids:
        return rc, ra, last
    return rc, ra


def raster_fn(absgrad_out: Optional[list] = None):
    def fn(means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids, bg, periodic=False):
        return rasterize_to_pixels(means2d, conics, cols, opac, width, height, tile_size, isect_offsets,
                                   flatten_ids, backgrounds=bg, absgrad_out=absgrad_out, periodic=periodic)
    return fn
