"""Photometric loss of the training step: 0.8 * L1 + 0.2 * (1 - SSIM)
(/root/reference/utils/gsplat_utils/gsplat_trainer.py:624-628), as ONE forward and ONE backward HIP
kernel (`so_ssim_l1_fwd/bwd`) on the rasteriser's channel-last output -- or, where the gradient is wanted at once
(`photometric_loss_and_grad`, the fused training step), as a single kernel (`so_ssim_l1_fused`).

`fused_ssim(img1, img2, padding="valid")` mirrors the call the reference makes into the CUDA-only
`fused_ssim` package (Dockerfile:55-60): 11x11 Gaussian window, sigma 1.5, C1=0.01^2, C2=0.03^2,
NCHW, mean of the SSIM map; "valid" crops the 5-pixel border where the window leaves the image.
"""
from __future__ import annotations

import torch
from torch import Tensor

from ._lib import call, ptr, stream


class _L1SSIM(torch.autograd.Function):
    """a * mean|x-y| + b * mean SSIM(x,y) + c   on channel-last [B,H,W,CH] images.

    When img1 needs a gradient the forward runs `so_ssim_l1_fused`: loss scalars AND d loss / d img1 in one launch (the
    derivative values of the SSIM map pass through LDS instead of three maps in HBM); the backward then only scales that
    gradient by the incoming v_loss.  Forward-only calls (eval) run `so_ssim_l1_fwd` without derivative maps."""

    @staticmethod
    def forward(ctx, img1: Tensor, img2: Tensor, a: float, b: float, c: float, valid: bool, one_minus: bool = False):
        """-> (loss, mean|x-y|, mean SSIM -- or 1 - mean SSIM with `one_minus`, which is what the kernel leaves behind)"""
        B, H, W, CH = img1.shape
        need_grad = ctx.needs_input_grad[0]
        n_l1 = float(B * H * W * CH)
        n_ss = float(B * CH * ((H - 10) * (W - 10) if valid else H * W))
        ctx.set_materialize_grads(False)
        if need_grad:
            work = torch.zeros(6, dtype=torch.float32, device=img1.device)     # sums[2] | loss, l1, 1 - ssim | ticket
            grad = torch.empty_like(img1)
            call("so_ssim_l1_fused", B, H, W, CH, ptr(img1), ptr(img2), 1 if valid else 0, a / n_l1, b / n_ss, 0, ptr(work), ptr(grad),
                 ptr(work[2:]), ptr(work[5:]), c, 0, stream())
            ctx.save_for_backward(grad)
            loss, l1, ssim = work[2], work[3], (work[4] if one_minus else 1.0 - work[4])
        else:
            sums = torch.zeros(2, dtype=torch.float32, device=img1.device)
            call("so_ssim_l1_fwd", B, H, W, CH, ptr(img1), ptr(img2), 1 if valid else 0, ptr(sums), 0, stream())
            l1 = sums[0] / n_l1
            ssim = sums[1] / n_ss
            loss = a * l1 + b * ssim + c
            if one_minus:
                ssim = 1.0 - ssim
        ctx.mark_non_differentiable(l1, ssim)
        return loss, l1, ssim

    @staticmethod
    def backward(ctx, v_loss, _v_l1=None, _v_ssim=None):
        (grad,) = ctx.saved_tensors
        if v_loss is None:
            return None, None, None, None, None, None, None
        return grad * v_loss.to(torch.float32), None, None, None, None, None, None


def _prep(x: Tensor) -> Tensor:
    assert x.dtype == torch.float32, x.dtype
    return x.contiguous()


def photometric_loss(colors: Tensor, pixels: Tensor, ssim_lambda: float = 0.2):
    """colors, pixels: [B,H,W,3] in 0..1.  Returns (loss, l1loss, ssimloss) as at gsplat_trainer.py:624-628."""
    assert colors.shape == pixels.shape and colors.dim() == 4, (colors.shape, pixels.shape)
    assert colors.shape[1] > 10 and colors.shape[2] > 10, "image smaller than the 11x11 SSIM window"
    return _L1SSIM.apply(_prep(colors), _prep(pixels.detach()), 1.0 - ssim_lambda, -ssim_lambda, ssim_lambda, True, True)


def photometric_loss_and_grad(colors: Tensor, pixels: Tensor, ssim_lambda: float = 0.2, *, rows: int = 0):
    """The same loss AND d loss / d colors in one launch (`so_ssim_l1_fused`: the derivative values of the SSIM map pass
    through LDS instead of three maps in HBM) -- what the fused training step runs.  No autograd graph is built.
    Returns (loss, l1loss, ssimloss, grad[B,H,W,CH]).  rows: output rows per workgroup, 0 = chosen by the launcher."""
    assert colors.shape == pixels.shape and colors.dim() == 4, (colors.shape, pixels.shape)
    B, H, W, CH = colors.shape
    assert H > 10 and W > 10, "image smaller than the 11x11 SSIM window"
    x, y = _prep(colors.detach()), _prep(pixels.detach())
    work = torch.zeros(6, dtype=torch.float32, device=x.device)        # sums[2] | loss, l1, ssimloss | ticket
    grad = torch.empty_like(x)
    n_l1, n_ss = float(B * H * W * CH), float(B * CH * (H - 10) * (W - 10))
    call("so_ssim_l1_fused", B, H, W, CH, ptr(x), ptr(y), 1, (1.0 - ssim_lambda) / n_l1, -ssim_lambda / n_ss, 0, ptr(work), ptr(grad),
         ptr(work[2:]), ptr(work[5:]), ssim_lambda, rows, stream())
    return work[2], work[3], work[4], grad


def fused_ssim(img1: Tensor, img2: Tensor, padding: str = "same", train: bool = True) -> Tensor:
    """Mean SSIM of NCHW images; the gradient flows to img1 only (img2 is the target)."""
    assert padding in ("same", "valid"), padding
    assert img1.shape == img2.shape and img1.dim() == 4, (img1.shape, img2.shape)
    a = img1.permute(0, 2, 3, 1)
    b = img2.detach().permute(0, 2, 3, 1)
    out, _, _ = _L1SSIM.apply(_prep(a), _prep(b), 0.0, 1.0, 0.0, padding == "valid")
    return out
