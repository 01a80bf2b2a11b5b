"""ctypes binding of libsplat_one_amd.so (the C ABI declared in include/splat_one_amd.h).

The library is the product; there is NO fallback.  If it is missing or an entry point reports
an error this module raises -- the HIP path is the only path (see DESIGN.md "boundary").
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPLAT_ONE_AMD_LIB: an alternative build of the same library (kernel experiments under tools/); never a fallback
LIB_PATH = os.environ.get("SPLAT_ONE_AMD_LIB") or os.path.join(_HERE, "lib", "libsplat_one_amd.so")

c_int, c_i64, c_f32, c_ptr = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

SO_ADAM_MAX_GROUPS = 8
SO_TILE_SLOTS = 12


class AdamGroup(ctypes.Structure):
    _fields_ = [("param", c_ptr), ("grad", c_ptr), ("exp_avg", c_ptr), ("exp_avg_sq", c_ptr),
                ("visibility", c_ptr), ("numel", c_i64), ("row_len", ctypes.c_int32),
                ("lr_step_size", c_f32), ("bc2_sqrt", c_f32)]


class StepDesc(ctypes.Structure):
    """Mirror of `so_step_desc` (include/splat_one_amd.h) -- same field order and types."""
    _fields_ = (
        [(n, c_ptr) for n in ("means", "log_scales", "quats", "logit_opacities", "sh0", "shN",
                              "viewmats", "Ks", "pixels", "backgrounds",
                              "radii", "means2d", "depths", "conics", "opacities", "colors",
                              "tiles_per_gauss", "counters", "isect_offsets", "key_buf", "flatten_ids",
                              "render_colors", "render_alphas", "last_ids", "loss_sums", "dmaps",
                              "v_render_colors", "zero_v_alphas", "rec", "vrec",
                              "v_means", "v_log_scales", "v_quats", "v_logit_opacities", "v_sh0", "v_shN",
                              "grad2d", "count")]
        + [("isect_capacity", c_i64)]
        + [(n, ctypes.c_int32) for n in ("abi_size", "C", "N", "K", "width", "height", "tile_size", "sh_degree",
                                         "camera_model", "antialiased", "absgrad", "raster_impl")]
        + [(n, c_f32) for n in ("eps2d", "near_plane", "far_plane", "radius_clip", "ssim_lambda", "opacity_reg",
                                "scale_reg")]
        + [("pixels_indirect", c_ptr), ("inputs_staged", ctypes.c_int32), ("tile_cull", ctypes.c_int32),
           ("overflow_flag_out", c_ptr), ("attr_rows_f16", c_ptr), ("tile_slots", c_ptr), ("bin_capacity", c_i64), ("fuse_adam", c_ptr),
           ("n_dev", c_ptr), ("tile_order", c_ptr), ("sort_in_rasteriser", ctypes.c_int32),
           ("bin_replicas", ctypes.c_int32), ("bin_sub_counts", c_ptr),
           ("bwd_seg_len", ctypes.c_int32), ("bwd_seg_count", ctypes.c_int32), ("bwd_seg_state", c_ptr),
           ("tile_order_ready", ctypes.c_int32)])


class RasterDesc(ctypes.Structure):
    """Mirror of `so_raster_desc` (so_rasterization_fwd / _bwd: gsplat's `rasterization` whole, one call each way)."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("abi_size", "C", "N", "K", "width", "height", "tile_size", "sh_degree",
                                               "camera_model", "antialiased", "absgrad", "tile_cull", "activated", "seq", "raster_impl")]
                + [(n, c_f32) for n in ("eps2d", "near_plane", "far_plane", "radius_clip")]
                + [("bin_capacity", c_i64)]
                + [(n, c_ptr) for n in ("means", "quats", "scales", "opacities", "sh0", "shN", "viewmats", "Ks", "backgrounds",
                                        "counters", "key_buf", "flatten_ids", "rec", "vrec", "status_out",
                                        "render_colors", "render_alphas", "last_ids",
                                        "v_render_colors", "v_render_alphas",
                                        "v_means", "v_quats", "v_scales", "v_opacities", "v_sh0", "v_shN", "v_means2d",
                                        "v_means2d_abs", "bin_sub_counts")]
                + [("bin_replicas", ctypes.c_int32), ("tile_order", c_ptr)])


class AdamFuse(ctypes.Structure):
    """Mirror of `so_adam_fuse` (the optimiser fused into the backward kernel)."""
    _fields_ = [("groups", AdamGroup * 6), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("step_counter", c_ptr)]


class ModelSet(ctypes.Structure):
    """Mirror of `so_model_set`: parameters, exp_avg, exp_avg_sq of the six tensors (capacity rows each)."""
    _fields_ = [("p", c_ptr * 6), ("m", c_ptr * 6), ("v", c_ptr * 6)]


class RefineParams(ctypes.Structure):
    """Mirror of `so_refine_params` (device-side DefaultStrategy refinement)."""
    _fields_ = [("grow_grad2d", c_f32), ("grow_scale3d", c_f32), ("prune_opa", c_f32), ("prune_scale3d", c_f32),
                ("prune_big", ctypes.c_int32), ("revised_opacity", ctypes.c_int32), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class McmcParams(ctypes.Structure):
    """Mirror of `so_mcmc_params` (device-side MCMCStrategy refinement)."""
    _fields_ = [("min_opacity", c_f32), ("cap_max", ctypes.c_int32), ("seed", ctypes.c_uint64), ("step", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class AttrShadow(ctypes.Structure):
    """Mirror of `so_attr_shadow`: where so_adam_step_dev_shadow keeps the float16 attribute rows current."""
    _fields_ = [("arec", c_ptr), ("stride_bytes", ctypes.c_int32), ("offset_bytes", ctypes.c_int32 * SO_ADAM_MAX_GROUPS)]


# name -> argtypes, exactly the prototypes of include/splat_one_amd.h
_SIGS = {
    "so_projection_fwd": [c_int, c_int] + [c_ptr] * 6 + [c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_int] + [c_ptr] * 6,
    "so_projection_bwd": [c_int, c_int] + [c_ptr] * 6 + [c_int, c_int, c_f32, c_int] + [c_ptr] * 11,
    "so_projection_packed": [c_int, c_int] + [c_ptr] * 6 + [c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_int] + [c_ptr] * 12,
    "so_projection_bwd_packed": [c_int, c_int, c_i64] + [c_ptr] * 6 + [c_int, c_int, c_f32, c_int] + [c_ptr] * 12,
    "so_sh_fwd": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr],
    "so_sh_bwd": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_isect_count": [c_int, c_int, c_ptr, c_ptr, c_int, c_int, c_int] + [c_ptr] * 6,
    "so_isect_fill": [c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_i64] + [c_ptr] * 7,
    "so_isect_emit_unsorted": [c_int, c_int] + [c_ptr] * 4 + [c_int, c_int, c_int] + [c_ptr] * 3,
    "so_isect_offset_encode": [c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr],
    "so_rasterize_fwd": [c_int] * 6 + [c_ptr] * 9 + [c_i64] + [c_ptr] * 4,
    "so_rasterize_bwd": [c_int] * 6 + [c_ptr] * 9 + [c_i64] + [c_ptr] * 10,
    "so_ssim_l1_fwd": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr],
    "so_ssim_l1_bwd": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_f32, c_ptr],
    "so_ssim_l1_fused": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_int,
                         c_ptr],
    "so_camera_inverse": [c_int, c_ptr, c_ptr, c_ptr],
    "so_debug_wave_reduce": [c_int, c_ptr, c_ptr, c_ptr],
    "so_debug_wave_reduce9": [c_int, c_ptr, c_ptr, c_ptr],
    "so_debug_cull": [c_i64, c_ptr, c_ptr, c_ptr],
    "so_debug_bin_counter_index": [c_i64, c_ptr, c_ptr],
    "so_isect_scan": [c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_preprocess_fwd": [c_int] * 4 + [c_ptr] * 8 + [c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_int] + [c_ptr] * 10 + [c_i64, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_ptr],
    "so_isect_sort_bins": [c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_rec_unpack": [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_shard_flag_put": [c_int, c_i64, c_ptr, c_ptr, c_ptr],
    "so_shard_flag_get": [c_int, c_i64, c_ptr, c_ptr, c_ptr],
    "so_preprocess_bwd": [c_int] * 4 + [c_ptr] * 8 + [c_int, c_int, c_f32, c_int, c_int] + [c_ptr] * 9 + [c_f32, c_f32] + [c_ptr] * 9 + [c_int, c_i64, c_ptr, c_ptr, c_ptr],
    "so_sh_view_colors_fwd": [c_int] * 4 + [c_ptr] * 6,
    "so_sh_view_colors_bwd": [c_int] * 4 + [c_ptr] * 9,
    "so_strategy_update_state": [c_int, c_i64, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_rec_pack": [c_i64] + [c_ptr] * 7,
    "so_rec_unpack_grads": [c_i64] + [c_ptr] * 7,
    "so_rasterize_fwd_packed": [c_int] * 5 + [c_ptr] * 5 + [c_i64] + [c_ptr] * 4,
    "so_rasterize_bwd_packed": [c_int] * 5 + [c_ptr] * 5 + [c_i64] + [c_ptr] * 5 + [c_int, c_ptr],
    "so_train_step_fwd_bwd": [ctypes.POINTER(StepDesc), c_ptr],
    "so_train_step_head": [ctypes.POINTER(StepDesc), c_ptr],
    "so_train_step_bwd_rows": [ctypes.POINTER(StepDesc), ctypes.c_int64, ctypes.c_int64, c_ptr],
    "so_render_forward": [ctypes.POINTER(StepDesc), c_ptr],
    "so_rasterization_fwd": [ctypes.POINTER(RasterDesc), c_ptr],
    "so_rasterization_bwd": [ctypes.POINTER(RasterDesc), c_ptr],
    "so_profile_enable": [c_int],
    "so_profile_read": [ctypes.POINTER(c_f32), ctypes.POINTER(c_int)],
    "so_adam_step_dev": [c_int, ctypes.POINTER(AdamGroup), ctypes.POINTER(c_f32), ctypes.POINTER(c_f32),
                         ctypes.c_double, ctypes.c_double, ctypes.c_double, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "so_adam_step_dev_shadow": [c_int, ctypes.POINTER(AdamGroup), ctypes.POINTER(c_f32), ctypes.POINTER(c_f32),
                                ctypes.c_double, ctypes.c_double, ctypes.c_double, c_ptr, c_int, c_int, c_ptr, c_ptr,
                                ctypes.POINTER(AttrShadow), c_ptr],
    "so_adam_step_dev_n": [c_int, ctypes.POINTER(AdamGroup), ctypes.POINTER(c_f32), ctypes.POINTER(c_f32),
                           ctypes.c_double, ctypes.c_double, ctypes.c_double, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_refine_default": [c_i64, c_int, ctypes.POINTER(ModelSet), c_ptr, ctypes.POINTER(ModelSet), c_ptr, c_ptr, c_ptr,
                          ctypes.POINTER(RefineParams), c_ptr, c_ptr, c_ptr],
    "so_reset_opacity": [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr],
    "so_attr_pack_f16": [c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_attr_pack_f16_n": [c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "so_preprocess_fwd_f16": [c_int] * 4 + [c_ptr] * 5 + [c_int, c_int, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_int] + [c_ptr] * 10 + [c_i64, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_ptr],
    "so_preprocess_bwd_f16": [c_int] * 4 + [c_ptr] * 5 + [c_int, c_int, c_f32, c_int, c_int] + [c_ptr] * 3 + [c_f32, c_f32] + [c_ptr] * 9 + [c_int, c_i64, c_ptr, c_ptr, c_ptr],
    "so_step_inputs": [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, ctypes.POINTER(c_f32),
                       ctypes.POINTER(c_f32), ctypes.c_double, ctypes.c_double, c_ptr, c_ptr, c_i64, c_int, c_i64, c_ptr,
                       c_ptr, c_ptr, c_i64, c_ptr],
    "so_compute_relocation": [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr],
    "so_inject_noise": [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr],
    "so_mcmc_refine": [c_i64, c_int, ctypes.POINTER(ModelSet), c_ptr, c_ptr, c_int, ctypes.POINTER(McmcParams), c_ptr, c_ptr, c_ptr],
    "so_inject_noise_dev": [c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_uint64, c_ptr, c_f32, c_f32, c_f32, c_ptr, c_ptr],
    "so_adam_step": [c_int, ctypes.POINTER(AdamGroup), ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int, c_ptr],
    "so_adam_step_scaled": [c_int, ctypes.POINTER(AdamGroup), ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int, c_ptr,
                            c_f32, c_ptr],
}

# include/splat_one_amd.h SO_ABI_VERSION: raised whenever an entry point's signature changes (2: so_step_inputs gained
# n_lists / lists_stat), so that a stale library against newer bindings fails at load instead of shifting arguments
ABI_VERSION = 2

_lib: Optional[ctypes.CDLL] = None


def exported_symbols():
    return ["so_abi_version", "so_last_error", "so_device_cu_count", "so_profile_num_stages",
            "so_profile_stage_name", "so_profile_stage_begin_end", "so_attr_rec_stride", "so_bin_counter_index", "so_refine_scratch_words", "so_mcmc_scratch_words", "so_projection_packed_blocks"] + list(_SIGS)


def load() -> ctypes.CDLL:
    """Load the shared library (built by `__graft_entry__.build()` / splat_one_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU or PyTorch fallback for this path)")
        lib = ctypes.CDLL(LIB_PATH)
        lib.so_abi_version.restype = c_int
        lib.so_last_error.restype = ctypes.c_char_p
        lib.so_device_cu_count.restype = c_int
        lib.so_profile_num_stages.restype = c_int
        lib.so_profile_stage_name.restype = ctypes.c_char_p
        lib.so_profile_stage_name.argtypes = [c_int]
        lib.so_attr_rec_stride.restype = c_i64
        lib.so_attr_rec_stride.argtypes = [c_int]
        lib.so_bin_counter_index.restype = c_i64
        lib.so_bin_counter_index.argtypes = [c_i64, c_i64]
        lib.so_projection_packed_blocks.restype = c_i64
        lib.so_projection_packed_blocks.argtypes = [c_int, c_int]
        lib.so_refine_scratch_words.restype = c_i64
        lib.so_refine_scratch_words.argtypes = [c_i64]
        lib.so_mcmc_scratch_words.restype = c_i64
        lib.so_mcmc_scratch_words.argtypes = [c_i64]
        for name, argtypes in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = c_int
        assert lib.so_abi_version() == ABI_VERSION, (f"{LIB_PATH} has ABI version {lib.so_abi_version()}, these bindings are written for "
                                                      f"{ABI_VERSION}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = lib
    return _lib


def ptr(t: Optional[torch.Tensor]) -> int:
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError("splat_one_amd: tensors must live on a HIP device (no CPU path exists)")
    if not t.is_contiguous():
        raise RuntimeError("splat_one_amd: tensor must be contiguous")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """Handle of torch's current HIP stream on the current device (the raw getter: 0.3 us against 10 us for
    `torch.cuda.current_stream().cuda_stream`, ten launches per operator-level step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# Optional per-entry-point timing with HIP events on the launch stream (bench.py / profiling only).
# PROFILE = None (off) | set of names to time | "all".  Results accumulate in PROFILE_EVENTS.
PROFILE = None
PROFILE_EVENTS: dict = {}


def call(name: str, *args) -> None:
    lib = load()
    timed = PROFILE is not None and (PROFILE == "all" or name in PROFILE)
    if timed:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(lib, name)(*args)
    if timed:
        e1.record()
        PROFILE_EVENTS.setdefault(name, []).append((e0, e1))
    if rc != 0:
        msg = lib.so_last_error().decode()
        raise RuntimeError(f"{name} failed with status {rc}: {msg}")


def stage_profile() -> dict:
    """{stage name: (n_calls, mean_ms)} of the C-side stage timers (so_profile_enable); synchronises."""
    lib = load()
    n = lib.so_profile_num_stages()
    ms = (c_f32 * n)()
    calls = (c_int * n)()
    call("so_profile_read", ms, calls)
    return {lib.so_profile_stage_name(i).decode(): (calls[i], ms[i] / max(1, calls[i])) for i in range(n) if calls[i]}


def profile_summary(reset: bool = True) -> dict:
    """{name: (n_calls, mean_ms)} from the recorded events (synchronises the device)."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in PROFILE_EVENTS.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[name] = (len(ms), sum(ms) / max(1, len(ms)))
    if reset:
        PROFILE_EVENTS.clear()
    return out
