"""Optimisers of the path.

`FusedAdam` is state-compatible with `torch.optim.Adam` (state[p] = {"step", "exp_avg", "exp_avg_sq"}),
so the densification strategy can rewrite moments exactly as it does for the reference's six
`torch.optim.Adam` instances (/root/reference/utils/gsplat_utils/gsplat_trainer.py:266-280); its
arithmetic runs in the `so_adam_step` HIP kernel.  `step_all` updates every tensor of every
optimiser in ONE launch and zeroes the gradients in the same pass (replaces :726-731).

`SelectiveAdam` mirrors `gsplat.optimizers.SelectiveAdam` (:269-270, :719-728): `step(visibility)`
updates only the rows whose visibility flag is set.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def _init_state(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)          # host scalar, as torch.optim.Adam(capturable=False)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _collect(self, visibility: Optional[torch.Tensor] = None) -> List[tuple]:
        """(param, grad, state, lr, betas, eps, visibility) for every parameter that has a gradient."""
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                assert not p.grad.is_sparse, "FusedAdam does not support sparse gradients"
                out.append((p, p.grad, self._init_state(p), group["lr"], group["betas"], group["eps"], visibility))
        return out

    @torch.no_grad()
    def step(self, closure=None, visibility: Optional[torch.Tensor] = None, zero_grad: bool = False):
        assert closure is None
        _launch(self._collect(visibility), zero_grad)


class SelectiveAdam(FusedAdam):
    """Adam restricted to visible rows: `step(visibility)` with visibility[N] bool."""

    @torch.no_grad()
    def step(self, visibility: torch.Tensor, zero_grad: bool = False):  # type: ignore[override]
        _launch(self._collect(visibility), zero_grad)


def _launch(items: List[tuple], zero_grad: bool) -> None:
    """One `so_adam_step` launch per (betas, eps) class and per SO_ADAM_MAX_GROUPS tensors."""
    if not items:
        return
    by_hyper: Dict[tuple, List[tuple]] = {}
    for it in items:
        by_hyper.setdefault((it[4], it[5]), []).append(it)
    for (betas, eps), its in by_hyper.items():
        for i0 in range(0, len(its), _lib.SO_ADAM_MAX_GROUPS):
            chunk = its[i0:i0 + _lib.SO_ADAM_MAX_GROUPS]
            arr = (_lib.AdamGroup * len(chunk))()
            keep = []
            torch._foreach_add_([it[2]["step"] for it in chunk], 1)      # (one dispatch for the six host-side step counters)
            for i, (p, g, st, lr, _b, _e, vis) in enumerate(chunk):
                t = float(st["step"])
                assert p.is_contiguous() and g.is_contiguous(), "parameters and gradients must be contiguous"
                vptr, row_len = 0, 1
                if vis is not None:
                    v8 = vis.to(torch.uint8).contiguous()
                    assert v8.numel() == p.shape[0], "visibility must have one flag per row"
                    keep.append(v8)
                    vptr, row_len = _lib.ptr(v8), max(1, p.numel() // max(1, p.shape[0]))
                arr[i] = _lib.AdamGroup(_lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]),
                                        vptr, p.numel(), row_len, lr / (1.0 - betas[0] ** t),
                                        math.sqrt(1.0 - betas[1] ** t))
            _lib.call("so_adam_step", len(chunk), arr, float(betas[0]), float(betas[1]), float(eps),
                      1 if zero_grad else 0, _lib.stream())


@torch.no_grad()
def step_all(optimizers: Iterable[FusedAdam], set_to_none: bool = True, zero_grad_in_place: bool = False,
             visibility: Optional[torch.Tensor] = None) -> None:
    """All tensors of all optimisers in one launch (per hyper-parameter class), then
    `zero_grad(set_to_none=True)` as the reference does (gsplat_trainer.py:726-731).  With
    `zero_grad_in_place` the gradients are zeroed inside the Adam pass instead (static buffers
    for hipGraph replay)."""
    items: List[tuple] = []
    opts = list(optimizers)
    for opt in opts:
        items.extend(opt._collect(visibility))
    _launch(items, zero_grad_in_place)
    if set_to_none and not zero_grad_in_place:
        for opt in opts:       # == opt.zero_grad(set_to_none=True) without torch's per-call profiler / dynamo wrappers
            for group in opt.param_groups:     # (20 us each: 0.12 ms per step for the six 3DGS tensors)
                for p in group["params"]:
                    p.grad = None
