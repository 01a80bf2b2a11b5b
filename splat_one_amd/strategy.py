"""Densification strategies -- drop-in for `gsplat.strategy.DefaultStrategy` as the reference drives
it (/root/reference/utils/gsplat_utils/gsplat_trainer.py:129-131, 345-350, 616-622, 744-752).

Behaviour restates the published ADC schedule of Kerbl et al. 2023 with gsplat's defaults
([upstream-memory]): accumulate screen-space gradient norms of visible
Gaussians, every `refine_every` steps duplicate small / split large high-gradient Gaussians, prune
transparent (and, after the first opacity reset, oversized) ones, reset opacities every
`reset_every` steps.  Every structural edit rewrites the parameters AND the Adam moments.

MI355X notes: the per-step statistics update has no host synchronisation (dense masked
reduction instead of `torch.where`); the structural edits (every 100th step) read the three
counts back to size the new tensors -- the only device->host traffic of the strategy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor


# ---------------------------------------------------------------------------------------------
# parameter / optimiser surgery (gsplat.strategy.ops)
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def _update_param_with_optimizer(param_fn: Callable[[str, Tensor], Tensor],
                                 optimizer_fn: Callable[[str, Tensor], Tensor],
                                 params: Union[Dict[str, torch.nn.Parameter], torch.nn.ParameterDict],
                                 optimizers: Dict[str, torch.optim.Optimizer],
                                 names: Optional[list] = None) -> None:
    if names is None:
        names = list(params.keys())   # every parameter, as gsplat does
    for name in names:
        param = params[name]
        new_param = param_fn(name, param)
        params[name] = new_param
        if name not in optimizers:
            assert not param.requires_grad, f"optimizer for {name} is missing"
            continue
        optimizer = optimizers[name]
        for i in range(len(optimizer.param_groups)):
            param_state = optimizer.state[param]
            del optimizer.state[param]
            for key in param_state.keys():
                if key != "step":
                    param_state[key] = optimizer_fn(key, param_state[key])
            optimizer.param_groups[i]["params"] = [new_param]
            optimizer.state[new_param] = param_state


def normalized_quat_to_rotmat(quat: Tensor) -> Tensor:
    w, x, y, z = torch.unbind(quat, dim=-1)
    mat = torch.stack([
        1 - 2 * (y ** 2 + z ** 2), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x ** 2 + z ** 2), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x ** 2 + y ** 2)], dim=-1)
    return mat.reshape(quat.shape[:-1] + (3, 3))


@torch.no_grad()
def duplicate(params, optimizers, state: Dict[str, Tensor], mask: Tensor) -> None:
    sel = torch.where(mask)[0]

    def param_fn(name: str, p: Tensor) -> Tensor:
        return torch.nn.Parameter(torch.cat([p, p[sel]]), requires_grad=p.requires_grad)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        return torch.cat([v, torch.zeros((len(sel), *v.shape[1:]), device=v.device, dtype=v.dtype)])

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            state[k] = torch.cat((v, v[sel]))


@torch.no_grad()
def split(params, optimizers, state: Dict[str, Tensor], mask: Tensor, revised_opacity: bool = False,
          generator: Optional[torch.Generator] = None) -> None:
    device = mask.device
    sel = torch.where(mask)[0]
    rest = torch.where(~mask)[0]
    scales = torch.exp(params["scales"][sel])
    quats = F.normalize(params["quats"][sel], dim=-1)
    rotmats = normalized_quat_to_rotmat(quats)
    noise = torch.randn(2, len(scales), 3, device=device, generator=generator)
    samples = torch.einsum("nij,nj,bnj->bni", rotmats, scales, noise)  # [2, n, 3]

    def param_fn(name: str, p: Tensor) -> Tensor:
        repeats = [2] + [1] * (p.dim() - 1)
        if name == "means":
            p_split = (p[sel] + samples).reshape(-1, 3)
        elif name == "scales":
            p_split = torch.log(scales / 1.6).repeat(2, 1)
        elif name == "opacities" and revised_opacity:
            new_opacities = 1.0 - torch.sqrt(1.0 - torch.sigmoid(p[sel]))
            p_split = torch.logit(new_opacities).repeat(repeats)
        else:
            p_split = p[sel].repeat(repeats)
        return torch.nn.Parameter(torch.cat([p[rest], p_split]), requires_grad=p.requires_grad)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        v_split = torch.zeros((2 * len(sel), *v.shape[1:]), device=device, dtype=v.dtype)
        return torch.cat([v[rest], v_split])

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            repeats = [2] + [1] * (v.dim() - 1)
            state[k] = torch.cat((v[rest], v[sel].repeat(repeats)))


@torch.no_grad()
def remove(params, optimizers, state: Dict[str, Tensor], mask: Tensor) -> None:
    sel = torch.where(~mask)[0]

    def param_fn(name: str, p: Tensor) -> Tensor:
        return torch.nn.Parameter(p[sel], requires_grad=p.requires_grad)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        return v[sel]

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            state[k] = v[sel]


@torch.no_grad()
def reset_opa(params, optimizers, state: Dict[str, Tensor], value: float) -> None:
    def param_fn(name: str, p: Tensor) -> Tensor:
        if name == "opacities":
            return torch.nn.Parameter(torch.clamp(p, max=torch.logit(torch.tensor(value)).item()),
                                      requires_grad=p.requires_grad)
        raise ValueError(f"Unexpected parameter name: {name}")

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        return torch.zeros_like(v)

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers, names=["opacities"])


# ---------------------------------------------------------------------------------------------
# MCMC operations (gsplat.strategy.ops: relocate / sample_add / inject_noise_to_position)
# ---------------------------------------------------------------------------------------------
def _multinomial_sample(weights: Tensor, n: int, replacement: bool = True,
                        generator: Optional[torch.Generator] = None) -> Tensor:
    num_elements = weights.size(0)
    if num_elements <= 2 ** 24:
        return torch.multinomial(weights, n, replacement=replacement, generator=generator)
    weights = weights / weights.sum()          # torch.multinomial is limited to 2^24 categories
    cdf = torch.cumsum(weights, dim=0)
    u = torch.rand(n, device=weights.device, generator=generator)
    return torch.searchsorted(cdf, u).clamp_max(num_elements - 1)


@torch.no_grad()
def compute_relocation(opacities: Tensor, scales: Tensor, ratios: Tensor, binoms: Tensor) -> Tuple[Tensor, Tensor]:
    """Equation (9) of 3DGS-MCMC: opacities[N], scales[N,3], ratios[N] int, binoms[n_max,n_max] ->
    (new_opacities[N], new_scales[N,3]).  HIP kernel `so_compute_relocation` (gsplat K13)."""
    from . import _lib
    N = opacities.shape[0]
    n_max = binoms.shape[0]
    assert scales.shape == (N, 3) and ratios.shape == (N,) and binoms.shape == (n_max, n_max)
    opacities = opacities.contiguous().float()
    scales = scales.contiguous().float()
    ratios = ratios.clamp(min=1, max=n_max).int().contiguous()
    binoms = binoms.contiguous().float()
    new_opacities = torch.empty_like(opacities)
    new_scales = torch.empty_like(scales)
    _lib.call("so_compute_relocation", N, _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(ratios), _lib.ptr(binoms),
              n_max, _lib.ptr(new_opacities), _lib.ptr(new_scales), _lib.stream())
    return new_opacities, new_scales


@torch.no_grad()
def relocate(params, optimizers, state: Dict[str, Tensor], mask: Tensor, binoms: Tensor, min_opacity: float = 0.005,
             generator: Optional[torch.Generator] = None) -> None:
    """Teleport the dead Gaussians (mask) onto alive ones sampled proportionally to opacity."""
    opacities = torch.sigmoid(params["opacities"])
    dead_indices = mask.nonzero(as_tuple=True)[0]
    alive_indices = (~mask).nonzero(as_tuple=True)[0]
    n = len(dead_indices)
    eps = torch.finfo(torch.float32).eps
    probs = opacities[alive_indices].flatten()
    sampled_idxs = alive_indices[_multinomial_sample(probs, n, replacement=True, generator=generator)]
    new_opacities, new_scales = compute_relocation(
        opacities=opacities[sampled_idxs], scales=torch.exp(params["scales"])[sampled_idxs],
        ratios=torch.bincount(sampled_idxs)[sampled_idxs] + 1, binoms=binoms)
    new_opacities = torch.clamp(new_opacities, max=1.0 - eps, min=min_opacity)

    def param_fn(name: str, p: Tensor) -> Tensor:
        p = p.detach().clone()
        if name == "opacities":
            p[sampled_idxs] = torch.logit(new_opacities)
        elif name == "scales":
            p[sampled_idxs] = torch.log(new_scales)
        p[dead_indices] = p[sampled_idxs]
        return torch.nn.Parameter(p)

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        v[sampled_idxs] = 0
        return v

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == mask.shape[0]:
            v[sampled_idxs] = 0


@torch.no_grad()
def sample_add(params, optimizers, state: Dict[str, Tensor], n: int, binoms: Tensor, min_opacity: float = 0.005,
               generator: Optional[torch.Generator] = None) -> None:
    """Add n Gaussians as copies of ones sampled proportionally to opacity (both get the relocated
    opacity / scale)."""
    opacities = torch.sigmoid(params["opacities"])
    n_before = opacities.shape[0]
    eps = torch.finfo(torch.float32).eps
    probs = opacities.flatten()
    sampled_idxs = _multinomial_sample(probs, n, replacement=True, generator=generator)
    new_opacities, new_scales = compute_relocation(
        opacities=opacities[sampled_idxs], scales=torch.exp(params["scales"])[sampled_idxs],
        ratios=torch.bincount(sampled_idxs)[sampled_idxs] + 1, binoms=binoms)
    new_opacities = torch.clamp(new_opacities, max=1.0 - eps, min=min_opacity)

    def param_fn(name: str, p: Tensor) -> Tensor:
        p = p.detach().clone()
        if name == "opacities":
            p[sampled_idxs] = torch.logit(new_opacities)
        elif name == "scales":
            p[sampled_idxs] = torch.log(new_scales)
        return torch.nn.Parameter(torch.cat([p, p[sampled_idxs]]))

    def optimizer_fn(key: str, v: Tensor) -> Tensor:
        v_new = torch.zeros((len(sampled_idxs), *v.shape[1:]), device=v.device, dtype=v.dtype)
        return torch.cat([v, v_new])

    _update_param_with_optimizer(param_fn, optimizer_fn, params, optimizers)
    for k, v in state.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n_before:
            v_new = torch.zeros((len(sampled_idxs), *v.shape[1:]), device=v.device, dtype=v.dtype)
            state[k] = torch.cat((v, v_new))


@torch.no_grad()
def inject_noise_to_position(params, optimizers, state: Dict[str, Tensor], scaler: float,
                             generator: Optional[torch.Generator] = None) -> None:
    """means += Sigma (N(0,I) * sigmoid_100(0.005 - opacity) * scaler): one HIP kernel on the raw parameters."""
    from . import _lib
    means = params["means"]
    noise = torch.randn(means.shape, device=means.device, dtype=means.dtype, generator=generator)
    _lib.call("so_inject_noise", means.shape[0], _lib.ptr(means.data), _lib.ptr(params["scales"].data.contiguous()),
              _lib.ptr(params["quats"].data.contiguous()), _lib.ptr(params["opacities"].data.flatten().contiguous()),
              _lib.ptr(noise), float(scaler), _lib.stream())


# ---------------------------------------------------------------------------------------------
# strategies
# ---------------------------------------------------------------------------------------------
@dataclass
class Strategy:
    def check_sanity(self, params, optimizers) -> None:
        trainable = set(name for name, p in params.items() if p.requires_grad)
        assert trainable == set(optimizers.keys()), (
            f"trainable parameters and optimizers must have the same keys, got {trainable} and {optimizers.keys()}")
        for optimizer in optimizers.values():
            assert len(optimizer.param_groups) == 1, "each optimizer must have exactly one param_group"

    def step_pre_backward(self, *args, **kwargs):
        pass

    def step_post_backward(self, *args, **kwargs):
        pass


@dataclass
class DefaultStrategy(Strategy):
    prune_opa: float = 0.005
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    prune_scale2d: float = 0.15
    refine_scale2d_stop_iter: int = 0
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    reset_every: int = 3000
    refine_every: int = 100
    pause_refine_after_reset: int = 0
    absgrad: bool = False
    revised_opacity: bool = False
    verbose: bool = False
    key_for_gradient: str = "means2d"

    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        state = {"grad2d": None, "count": None, "scene_scale": scene_scale}
        if self.refine_scale2d_stop_iter > 0:
            state["radii"] = None
        return state

    def check_sanity(self, params, optimizers) -> None:
        super().check_sanity(params, optimizers)
        for key in ["means", "scales", "quats", "opacities"]:
            assert key in params, f"{key} is required in params but missing."

    def step_pre_backward(self, params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any]):
        assert self.key_for_gradient in info, "The 2D means of the Gaussians is required but missing."
        info[self.key_for_gradient].retain_grad()

    def step_post_backward(self, params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           packed: bool = False, generator: Optional[torch.Generator] = None):
        if step >= self.refine_stop_iter:
            return
        self._update_state(params, state, info, packed=packed)
        if (step > self.refine_start_iter and step % self.refine_every == 0
                and step % self.reset_every >= self.pause_refine_after_reset):
            n_dupli, n_split = self._grow_gs(params, optimizers, state, step, generator)
            n_prune = self._prune_gs(params, optimizers, state, step)
            if self.verbose:
                print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                      f"Now having {len(params['means'])} GSs.")
            state["grad2d"].zero_()
            state["count"].zero_()
            if self.refine_scale2d_stop_iter > 0:
                state["radii"].zero_()
        if step % self.reset_every == 0 and step > 0:
            # (`step > 0`: the schedule of Kerbl et al. starts counting at 1, so the first reset is at
            # reset_every, not at the very first iteration)
            reset_opa(params=params, optimizers=optimizers, state=state, value=self.prune_opa * 2.0)

    @torch.no_grad()
    def _update_state(self, params, state: Dict[str, Any], info: Dict[str, Any], packed: bool = False):
        for key in ["width", "height", "n_cameras", "radii", self.key_for_gradient]:
            assert key in info, f"{key} is required but missing."
        m2 = info[self.key_for_gradient]
        grads = m2.absgrad if self.absgrad else m2.grad                     # [C, N, 2] | packed [nnz, 2]
        n_gaussian = len(list(params.values())[0])
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n_gaussian, device=grads.device)
        if state["count"] is None:
            state["count"] = torch.zeros(n_gaussian, device=grads.device)
        if self.refine_scale2d_stop_iter > 0 and state["radii"] is None:
            state["radii"] = torch.zeros(n_gaussian, device=grads.device)
        sx = info["width"] / 2.0 * info["n_cameras"]
        sy = info["height"] / 2.0 * info["n_cameras"]
        radii = info["radii"]
        # radii: a dense [C,N] array, or the radius slot of the 64-byte records (`rasterization`'s whole-operator path: a
        # view with one element every `stride` words)
        r_stride = radii.stride(-1) if radii.dim() == 2 else 0
        r_ok = (radii.dtype == torch.int32 and radii.dim() == 2
                and (radii.is_contiguous() or (r_stride > 1 and radii.stride(0) == radii.shape[1] * r_stride)))
        if (not packed and grads.is_cuda and grads.dim() == 3 and grads.dtype == torch.float32 and grads.is_contiguous() and r_ok
                and all(state[k].is_contiguous() and state[k].dtype == torch.float32 for k in ("grad2d", "count"))):
            # dense layout on the device: ONE launch instead of a dozen torch kernels (csrc/refine.hip k_strategy_update)
            from ._lib import call, stream
            ptr = lambda t: 0 if t is None else t.data_ptr()
            C, N = radii.shape
            call("so_strategy_update_state", C, N, ptr(grads), ptr(radii), int(r_stride), float(sx), float(sy),
                 1.0 / float(max(info["width"], info["height"])), ptr(state["grad2d"]), ptr(state["count"]),
                 ptr(state["radii"]) if self.refine_scale2d_stop_iter > 0 else 0, stream())
            return
        norms = torch.sqrt((grads[..., 0] * sx) ** 2 + (grads[..., 1] * sy) ** 2)
        if packed:       # one row per visible (camera, Gaussian) pair, named by info["gaussian_ids"] (radii > 0 on all of them)
            gs_ids = info["gaussian_ids"]
            state["grad2d"].index_add_(0, gs_ids, norms)
            state["count"].index_add_(0, gs_ids, torch.ones_like(norms))
            if self.refine_scale2d_stop_iter > 0:
                r = info["radii"].to(torch.float32) / float(max(info["width"], info["height"]))
                state["radii"].scatter_reduce_(0, gs_ids, r, reduce="amax", include_self=True)
            return
        sel = info["radii"] > 0                                            # [C, N]
        # dense masked sums == index_add_ over the visible ids, without torch.where's host sync
        state["grad2d"] += (norms * sel).sum(dim=0)
        state["count"] += sel.sum(dim=0).to(torch.float32)
        if self.refine_scale2d_stop_iter > 0:
            r = (info["radii"].to(torch.float32) / float(max(info["width"], info["height"]))) * sel
            state["radii"] = torch.maximum(state["radii"], r.max(dim=0).values)

    @torch.no_grad()
    def _grow_gs(self, params, optimizers, state, step, generator=None) -> Tuple[int, int]:
        count = state["count"]
        grads = state["grad2d"] / count.clamp_min(1)
        device = grads.device
        is_grad_high = grads > self.grow_grad2d
        is_small = torch.exp(params["scales"]).max(dim=-1).values <= self.grow_scale3d * state["scene_scale"]
        is_dupli = is_grad_high & is_small
        is_split = is_grad_high & ~is_small
        if step < self.refine_scale2d_stop_iter:
            is_split |= state["radii"] > self.grow_scale2d
        n_dupli, n_split = (int(v) for v in torch.stack([is_dupli.sum(), is_split.sum()]).tolist())
        if n_dupli > 0:
            duplicate(params=params, optimizers=optimizers, state=state, mask=is_dupli)
        # Gaussians added by duplication are not split
        is_split = torch.cat([is_split, torch.zeros(n_dupli, dtype=torch.bool, device=device)])
        if n_split > 0:
            split(params=params, optimizers=optimizers, state=state, mask=is_split,
                  revised_opacity=self.revised_opacity, generator=generator)
        return n_dupli, n_split

    @torch.no_grad()
    def _prune_gs(self, params, optimizers, state, step) -> int:
        is_prune = torch.sigmoid(params["opacities"].flatten()) < self.prune_opa
        if step > self.reset_every:
            is_too_big = torch.exp(params["scales"]).max(dim=-1).values > self.prune_scale3d * state["scene_scale"]
            if step < self.refine_scale2d_stop_iter:
                is_too_big |= state["radii"] > self.prune_scale2d
            is_prune = is_prune | is_too_big
        n_prune = int(is_prune.sum().item())
        if n_prune > 0:
            remove(params=params, optimizers=optimizers, state=state, mask=is_prune)
        return n_prune


@dataclass
class MCMCStrategy(Strategy):
    """3D Gaussian Splatting as Markov Chain Monte Carlo (Kheradmand et al. 2024) with gsplat's
    defaults -- the strategy of the reference's `mcmc` preset (gsplat_trainer.py:975-983), driven with
    `lr=` the current means learning rate (:753-761).  Use with opacity_reg / scale_reg = 0.01."""
    cap_max: int = 1_000_000
    noise_lr: float = 5e5
    refine_start_iter: int = 500
    refine_stop_iter: int = 25_000
    refine_every: int = 100
    min_opacity: float = 0.005
    verbose: bool = False

    def initialize_state(self) -> Dict[str, Any]:
        import math
        n_max = 51
        binoms = torch.zeros((n_max, n_max))
        for n in range(n_max):
            for k in range(n + 1):
                binoms[n, k] = math.comb(n, k)
        return {"binoms": binoms}

    def check_sanity(self, params, optimizers) -> None:
        super().check_sanity(params, optimizers)
        for key in ["means", "scales", "quats", "opacities"]:
            assert key in params, f"{key} is required in params but missing."

    def step_post_backward(self, params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           lr: float, generator: Optional[torch.Generator] = None) -> Tuple[int, int]:
        state["binoms"] = state["binoms"].to(params["means"].device)
        binoms = state["binoms"]
        n_relocated = n_new = 0
        if step < self.refine_stop_iter and step > self.refine_start_iter and step % self.refine_every == 0:
            n_relocated = self._relocate_gs(params, optimizers, binoms, generator)
            n_new = self._add_new_gs(params, optimizers, binoms, generator)
            if self.verbose:
                print(f"Step {step}: Relocated {n_relocated} GSs. Added {n_new} GSs. "
                      f"Now having {len(params['means'])} GSs.")
        inject_noise_to_position(params=params, optimizers=optimizers, state={}, scaler=lr * self.noise_lr,
                                 generator=generator)
        return n_relocated, n_new

    @torch.no_grad()
    def _relocate_gs(self, params, optimizers, binoms: Tensor, generator=None) -> int:
        opacities = torch.sigmoid(params["opacities"].flatten())
        dead_mask = opacities <= self.min_opacity
        n_gs = int(dead_mask.sum().item())
        if n_gs > 0:
            relocate(params=params, optimizers=optimizers, state={}, mask=dead_mask, binoms=binoms,
                     min_opacity=self.min_opacity, generator=generator)
        return n_gs

    @torch.no_grad()
    def _add_new_gs(self, params, optimizers, binoms: Tensor, generator=None) -> int:
        current_n_points = len(params["means"])
        n_target = min(self.cap_max, int(1.05 * current_n_points))
        n_gs = max(0, n_target - current_n_points)
        if n_gs > 0:
            sample_add(params=params, optimizers=optimizers, state={}, n=n_gs, binoms=binoms,
                       min_opacity=self.min_opacity, generator=generator)
        return n_gs
