"""camera_models.json handling (host side): the per-workdir table of camera models the reference's GUI
edits and its SfM / training front ends read.

Restates `CameraModelManager.load_camera_models` of the reference's app/camera_models.py:240-292 without Qt:
base table `camera_models.json` (created with one default perspective 1920x1080 model when missing or
unreadable), `camera_models_overrides.json` merged key-wise on top (an override of an unknown camera adds
it), and the merged table written back to `camera_models.json`.  Pinned by tests/golden/g9_camera_models.json
(outputs of the reference class itself on the fixture of its own tests/test_camera_models.py:13-41).
`intrinsics` turns one entry into the pinhole K of SURVEY.md 8d (fx = fy = focal_ratio * max(w, h),
principal point at the centre) for the projection types the rasteriser supports.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Dict

import torch

DEFAULT_MODEL = {"Perspective": {"projection_type": "perspective", "width": 1920, "height": 1080, "focal_ratio": 1.0}}


def _read_json(path: str):
    try:
        with open(path, "r") as f:
            return json.load(f)
    except Exception:   # noqa: BLE001  (missing, unreadable or malformed: the reference falls back the same way)
        return None


def load_camera_models(workdir: str, write_back: bool = True) -> Dict[str, dict]:
    base_path = os.path.join(workdir, "camera_models.json")
    base = _read_json(base_path) if os.path.exists(base_path) else None
    if base is None:
        base = copy.deepcopy(DEFAULT_MODEL)
    overrides = _read_json(os.path.join(workdir, "camera_models_overrides.json")) or {}
    merged = dict(base)
    for name, params in overrides.items():
        if name in merged:
            merged[name].update(params)
        else:
            merged[name] = params
    if write_back:
        try:
            with open(base_path, "w") as f:
                json.dump(merged, f, indent=4)
        except OSError:
            pass
    return merged


def intrinsics(model: dict) -> torch.Tensor:
    """[3,3] K of a perspective / fisheye entry (the spherical model has no pinhole K: ValueError)."""
    ptype = model.get("projection_type", "perspective")
    if ptype not in ("perspective", "fisheye", "brown"):
        raise ValueError(f"projection_type {ptype!r} has no pinhole intrinsics")
    w, h = int(model["width"]), int(model["height"])
    f = float(model.get("focal_ratio", 1.0)) * max(w, h)
    return torch.tensor([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]])
