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
    g = gaussian_window(dtype=dtype)

    def blur(x):
        # the 11x11 window is the outer product g (x) g: rows then columns, each as a sum of shifted slices over the zero
        # 'same' padding.  Same sum as F.conv2d(x, g[:, None] * g[None, :], padding=5) regrouped (differs by 1e-15 in
        # float64) -- torch's float64 grouped conv2d on the CPU took 10 s forward + backward per blur at 1080p, this 0.6 s
        H, W = x.shape[-2:]
        xp = F.pad(x, (0, 0, 5, 5))
        y = sum(g[k] * xp[..., k:k + H, :] for k in range(11))
        yp = F.pad(y, (5, 5, 0, 0))
        return sum(g[k] * yp[..., k:k + W] for k in range(11))

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


def photometric_loss(colors, pixels, ssim_lambda=0.2, dtype=torch.float64, l1_signs=None):
    """colors, pixels [B,H,W,3] -> (loss, l1, ssimloss).

    l1_signs [B,H,W,3] (optional, values in {-1, 0, +1}): the sign pattern sign(colors - pixels) to DIFFERENTIATE the L1
    term with, instead of the one this function's own `colors` give.  |x - y| is not differentiable at x = y: of the
    6.2 M pixel channels of a 1080p image, a dozen have |x - y| below the float32 error of the rendered value (~1e-6),
    and there a float32 renderer and a float64 one legitimately take opposite signs -- each such pixel moves the
    gradient of the few Gaussians covering it by percent (measured in round 2).  A parity test
    that passes the device's own signs compares the arithmetic on identical discrete decisions; the VALUE of the loss is
    the true |x - y| either way, and `l1_sign_report` counts the disagreements and checks that they are such ties."""
    colors, pixels = colors.to(dtype), pixels.to(dtype)
    l1 = (colors - pixels).abs().mean()
    if l1_signs is not None:
        lin = (l1_signs.to(dtype) * (colors - pixels)).mean()
        l1 = lin + (l1 - lin).detach()          # value of |x - y|, gradient through the given signs
    ssimloss = 1.0 - fused_ssim(colors.permute(0, 3, 1, 2), pixels.permute(0, 3, 1, 2), padding="valid", dtype=dtype)
    return l1 * (1.0 - ssim_lambda) + ssimloss * ssim_lambda, l1, ssimloss


def l1_sign_report(colors_oracle, colors_device, pixels):
    """Where do sign(x - y) of the oracle's render and of the device's render differ, and by how little do those pixels
    miss x = y?  -> {"flips": n, "max_abs_diff_at_flips": max |x_oracle - y| over them}"""
    xo, xd, y = colors_oracle.detach().double().cpu(), colors_device.detach().double().cpu(), pixels.detach().double().cpu()
    flip = torch.sign(xo - y) != torch.sign(xd - y)
    return {"flips": int(flip.sum()), "max_abs_diff_at_flips": float((xo - y).abs()[flip].max()) if flip.any() else 0.0,
            "pixel_channels": int(flip.numel())}


def disparity_loss(depths, points, depths_gt, width, height, dtype=torch.float64):
    """The depth-supervision term of /root/reference/utils/gsplat_utils/gsplat_trainer.py:629-644, without the scene scale
    and depth_lambda factors the caller applies (:644-645): the rendered expected depth [B,H,W,1] (render_mode "RGB+ED",
    :595) bilinearly sampled at the SfM points [B,M,2] in pixel coordinates (grid_sample, align_corners=True), disparity
    1/d where d > 0 and 0 elsewhere, mean |disp - 1/depths_gt| over all B x M points.  Restated with explicit bilinear
    weights (no grid_sample): x = px (W-1)/(W-1) = px, so the sample is the 2x2 neighbourhood of (px, py), zero outside."""
    d, p, g = depths.to(dtype)[..., 0], points.to(dtype), depths_gt.to(dtype)
    B, H, W = d.shape
    assert (W, H) == (width, height)
    x, y = p[..., 0], p[..., 1]
    x0, y0 = torch.floor(x), torch.floor(y)
    out = torch.zeros_like(x)
    bi = torch.arange(B)[:, None].expand_as(x)
    for dx in (0, 1):
        for dy in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            w = (1 - (x - xi).abs()) * (1 - (y - yi).abs())
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            v = d[bi, yi.clamp(0, H - 1).long(), xi.clamp(0, W - 1).long()]
            out = out + torch.where(ok, w * v, torch.zeros_like(v))
    # value of where(d > 0, 1 / d, 0) (:641); the inner where only keeps autograd from forming 0 * inf at d = 0 exactly (a
    # point over an empty pixel), where the reference's own expression differentiates to NaN
    pos = out > 0
    disp = torch.where(pos, 1.0 / torch.where(pos, out, torch.ones_like(out)), torch.zeros_like(out))
    return (disp - 1.0 / g).abs().mean()
