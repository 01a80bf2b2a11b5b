"""Measured parity numbers of the GPU tests, written next to pass/fail ("commit the measured errors").

Every `-m gpu` parity test at a BASELINE.json size calls `record(...)`; the entries are merged into ONE JSON file --
`$SPLAT_ONE_AMD_PARITY_JSON`, default `gpurun_out/parity_r04.json` (gpurun copies it back; the tracked copy is
`profiles/parity_r04.json`).  Helpers here only measure; the bars are asserted in the tests."""
import json
import os
import time
from typing import Dict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("SPLAT_ONE_AMD_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity_r04.json")


def record(section: str, **metrics) -> None:
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    data = {}
    if os.path.exists(PATH):
        try:
            with open(PATH) as f:
                data = json.load(f)
        except (OSError, ValueError):
            data = {}
    entry = data.setdefault(section, {})
    entry.update(metrics)
    entry["recorded_at"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
    with open(PATH, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def quats_unfloored(g_h: Dict[str, torch.Tensor], g_o: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """The quaternion gradient WITHOUT the floor of `grad_errors`, with both norms next to it: ||g - g*|| / ||g*_quats||.
    Where the scene is isotropic (reference init: equal scales) ||g*_quats|| is rounding noise itself -- 1e-8 of
    ||g*_scales|| -- and the ratio says nothing; `quats_over_scales` tells the reader which case this is."""
    ref = g_o["quats"].detach().cpu().double()
    d = g_h["quats"].detach().cpu().double() - ref
    sc = g_o["scales"].detach().cpu().double().norm().item()
    return {"quats_nofloor": d.norm().item() / max(ref.norm().item(), 1e-300), "quats_abs_err": d.norm().item(),
            "quats_ref_norm": ref.norm().item(), "scales_ref_norm": sc, "quats_over_scales": ref.norm().item() / max(sc, 1e-300)}


def grad_errors(g_h: Dict[str, torch.Tensor], g_o: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Per-tensor ||g - g*|| / ||g*|| over ALL rows (no trimming).  `quats`: for (nearly) isotropic Gaussians the true
    gradient is zero and only rounding remains, so its denominator is ||g*_quats|| + 0.01 ||g*_scales|| (at the 1e-3
    bar: an absolute floor of 1e-5 ||g*_scales||, as in every parity test of this repo)."""
    out = {}
    for k in g_o:
        ref = g_o[k].detach().cpu().double()
        d = g_h[k].detach().cpu().double() - ref
        floor = 1e-2 * g_o["scales"].detach().cpu().double().norm().item() if k == "quats" and "scales" in g_o else 0.0
        out[k] = d.norm().item() / max(ref.norm().item() + floor, 1e-300)
    return out
