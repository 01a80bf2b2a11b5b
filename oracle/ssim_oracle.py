"""Float64 torch restatement of the photometric loss (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/utils/gsplat_utils/gsplat_trainer.py:624-628.  `fused_ssim` is a CUDA-only
dependency of the reference (Dockerfile:55-60), absent here; its published algorithm is the
standard SSIM of Wang et al. 2004 with an 11x11 Gaussian window (sigma 1.5), C1=0.01^2, C2=0.03^2,
zero 'same' padding, the mean taken over the 'valid' interior when padding="valid".
"""
import torch
import torch.nn.functional as F


def gaussian_window(size=11, sigma=1.5, dtype=torch.float64):
    x = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).to(dtype)


def ssim_map(img1, img2, dtype=torch.float64):
    """NCHW -> SSIM map [B,C,H,W] with zero 'same' padding."""
    img1, img2 = img1.to(dtype), img2.to(dtype)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    CH = img1.shape[1]
    g = gaussian_window(dtype=dtype)
    k2 = (g[:, None] * g[None, :]).expand(CH, 1, 11, 11)

    def blur(x):
        return F.conv2d(x, k2, padding=5, groups=CH)

    mu1, mu2 = blur(img1), blur(img2)
    s11 = blur(img1 * img1) - mu1 * mu1
    s22 = blur(img2 * img2) - mu2 * mu2
    s12 = blur(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))


def fused_ssim(img1, img2, padding="same", dtype=torch.float64):
    m = ssim_map(img1, img2, dtype)
    if padding == "valid":
        m = m[:, :, 5:-5, 5:-5]
    return m.mean()


def photometric_loss(colors, pixels, ssim_lambda=0.2, dtype=torch.float64):
    """colors, pixels [B,H,W,3] -> (loss, l1, ssimloss)."""
    colors, pixels = colors.to(dtype), pixels.to(dtype)
    l1 = (colors - pixels).abs().mean()
    ssimloss = 1.0 - fused_ssim(colors.permute(0, 3, 1, 2), pixels.permute(0, 3, 1, 2), padding="valid", dtype=dtype)
    return l1 * (1.0 - ssim_lambda) + ssimloss * ssim_lambda, l1, ssimloss
