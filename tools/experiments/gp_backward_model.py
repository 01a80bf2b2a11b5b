"""Numerical model of a Gaussian-parallel backward for long per-tile lists: lanes own
Gaussians, pixels are walked front to back, and the colour BEHIND a Gaussian -- which the usual backward accumulates
back to front -- is obtained as a difference,

    S_i = (C_final - T_final * background) - sum_{j <= i} c_j alpha_j T_j ,

so that only front-to-back state (T_i and the running colour) has to travel from lane to lane.  The question answered
here, on the CPU and before any kernel is written: what does the cancellation in that difference cost in float32?
Compared: the standard back-to-front float32 backward and the difference formulation in float32, both against float64,
for the gradients the rasteriser hands to the projection backward (v_opacity, v_colour, v_sigma per (Gaussian, pixel),
summed over the pixels of a tile), on synthetic tiles with the dense regime's statistics (hundreds of faint, wide splats).

    python tools/experiments/gp_backward_model.py [list_length] [tiles]
"""
import sys

import torch


def make_tile(L, gen, dtype):
    px = torch.stack(torch.meshgrid(torch.arange(16.) + 0.5, torch.arange(16.) + 0.5, indexing="ij"), -1).reshape(256, 2).to(dtype)
    mu = (torch.rand(L, 2, generator=gen) * 48 - 16).to(dtype)                 # centres in and around the tile
    s = (torch.rand(L, generator=gen) * 20 + 3).to(dtype)                      # footprints of 3..23 pixels
    op = (torch.rand(L, generator=gen) * 0.25 + 0.02).to(dtype)                # faint: the reference's init_opa = 0.1
    col = torch.rand(L, 3, generator=gen).to(dtype)
    d = px[None] - mu[:, None]                                                 # [L,256,2]
    sigma = 0.5 * (d * d).sum(-1) / (s * s)[:, None]
    return sigma, op, col


def forward(sigma, op, col, bg):
    alpha = torch.clamp(op[:, None] * torch.exp(-sigma), max=0.999)
    alpha = torch.where(alpha < 1.0 / 255.0, torch.zeros_like(alpha), alpha)
    L = alpha.shape[0]
    T = torch.ones(256, dtype=alpha.dtype)
    Ts, live = [], torch.ones(256, dtype=torch.bool)
    for i in range(L):                                                         # gsplat: stop a pixel when T would drop below 1e-4
        nxt = T * (1 - alpha[i])
        live = live & (nxt > 1e-4)
        a = torch.where(live, alpha[i], torch.zeros_like(alpha[i]))
        alpha[i] = a
        Ts.append(T.clone())
        T = T * (1 - a)
    Ts = torch.stack(Ts)
    C = (alpha * Ts)[:, :, None] * col[:, None, :]
    return alpha, Ts, T, C.sum(0) + T[:, None] * bg


def backward_standard(alpha, Ts, Tf, col, bg, vC):
    """back to front, as k_rasterize_bwd: S accumulates the colour behind"""
    L = alpha.shape[0]
    S = torch.zeros(256, 3, dtype=alpha.dtype)
    v_alpha = torch.zeros_like(alpha)
    for i in range(L - 1, -1, -1):
        ra = 1.0 / (1.0 - alpha[i])
        v_alpha[i] = ((col[i][None] * Ts[i][:, None] - S * ra[:, None]) * vC).sum(-1) - (Tf * ra) * (bg[None] * vC).sum(-1)
        S = S + (alpha[i] * Ts[i])[:, None] * col[i][None]
    return v_alpha


def backward_difference(alpha, Ts, Tf, C, col, bg, vC):
    """front to back: S_i = (C - Tf bg) - running colour including i"""
    L = alpha.shape[0]
    base = C - Tf[:, None] * bg[None]
    run = torch.zeros(256, 3, dtype=alpha.dtype)
    v_alpha = torch.zeros_like(alpha)
    for i in range(L):
        run = run + (alpha[i] * Ts[i])[:, None] * col[i][None]
        S = base - run
        ra = 1.0 / (1.0 - alpha[i])
        v_alpha[i] = ((col[i][None] * Ts[i][:, None] - S * ra[:, None]) * vC).sum(-1) - (Tf * ra) * (bg[None] * vC).sum(-1)
    return v_alpha


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 417
    tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    worst = {"standard": 0.0, "difference": 0.0}
    for t in range(tiles):
        gen = torch.Generator().manual_seed(100 + t)
        out = {}
        for dtype in (torch.float64, torch.float32):
            g = torch.Generator().manual_seed(100 + t)
            sigma, op, col = make_tile(L, g, dtype)
            bg = torch.tensor([0.3, 0.5, 0.7], dtype=dtype)
            vC = (torch.rand(256, 3, generator=torch.Generator().manual_seed(7 + t)) - 0.5).to(dtype)
            alpha, Ts, Tf, C = forward(sigma, op, col, bg)
            va_s = backward_standard(alpha, Ts, Tf, col, bg, vC)
            va_d = backward_difference(alpha, Ts, Tf, C, col, bg, vC)
            mask = alpha > 0
            # what leaves the rasteriser per Gaussian: sums over the tile's pixels of v_alpha-weighted terms
            res = {}
            for name, va in (("standard", va_s), ("difference", va_d)):
                va = torch.where(mask, va, torch.zeros_like(va))
                v_op = (va * torch.exp(-sigma)).sum(1)                          # d alpha / d opacity
                v_sig = (-va * alpha).sum(1)                                    # d alpha / d sigma  (-> conic, mean)
                res[name] = torch.stack([v_op, v_sig]).double()
            out[dtype] = (res, int(mask.sum()), float(Tf.mean()))
        ref = out[torch.float64][0]["standard"]
        assert (ref - out[torch.float64][0]["difference"]).norm() <= 1e-12 * ref.norm()      # same mathematics
        line = [f"tile {t}: {out[torch.float32][1]} live (Gaussian, pixel) pairs, mean final T {out[torch.float32][2]:.4f}"]
        for name in ("standard", "difference"):
            e = ((out[torch.float32][0][name] - ref).norm() / ref.norm()).item()
            worst[name] = max(worst[name], e)
            line.append(f"{name} {e:.2e}")
        print("  ".join(line))
    print(f"list length {L}: worst relative error of the per-Gaussian gradient sums in float32 -- back-to-front {worst['standard']:.2e}, "
          f"difference formulation {worst['difference']:.2e}  (the parity bar is 1e-3)")


if __name__ == "__main__":
    main()
