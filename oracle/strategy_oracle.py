"""Loop-level restatement of the default densification strategy (TEST INFRASTRUCTURE; see
oracle/__init__.py).  Follows the published ADC procedure (Kerbl et al. 2023) with the defaults the
reference relies on through `gsplat.strategy.DefaultStrategy`
(/root/reference/utils/gsplat_utils/gsplat_trainer.py:129-131, 345-350, 616-622, 744-752;
SURVEY.md B.3).  Plain Python loops over small N -- deliberately nothing shared with the product."""
import math

import torch


def update_state(grad2d, count, means2d_grad, radii, width, height, n_cameras):
    """grad2d[N], count[N] += over visible (camera, Gaussian) pairs."""
    C, N = radii.shape
    g2, cn = grad2d.clone(), count.clone()
    for c in range(C):
        for n in range(N):
            if radii[c, n] > 0:
                gx = means2d_grad[c, n, 0].item() * width / 2.0 * n_cameras
                gy = means2d_grad[c, n, 1].item() * height / 2.0 * n_cameras
                g2[n] += math.sqrt(gx * gx + gy * gy)
                cn[n] += 1
    return g2, cn


def refine_masks(grad2d, count, log_scales, opac_logits, step, scene_scale, grow_grad2d=0.0002, grow_scale3d=0.01,
                 prune_opa=0.005, prune_scale3d=0.1, reset_every=3000):
    """-> (is_dupli[N], is_split[N]) and a function computing is_prune on the grown set."""
    N = grad2d.shape[0]
    dup, spl = [], []
    for n in range(N):
        avg = grad2d[n].item() / max(count[n].item(), 1.0)
        high = avg > grow_grad2d
        small = math.exp(log_scales[n].max().item()) <= grow_scale3d * scene_scale
        dup.append(high and small)
        spl.append(high and not small)

    def prune(log_scales2, opac_logits2):
        out = []
        for n in range(log_scales2.shape[0]):
            p = 1.0 / (1.0 + math.exp(-opac_logits2[n].item())) < prune_opa
            if step > reset_every:
                p = p or math.exp(log_scales2[n].max().item()) > prune_scale3d * scene_scale
            out.append(p)
        return torch.tensor(out)

    return torch.tensor(dup), torch.tensor(spl), prune


# ------------------------------------------------------------------------------------------------
# MCMC strategy (Kheradmand et al. 2024, "3D Gaussian Splatting as Markov Chain Monte Carlo", eq. 9;
# reached through the reference's `mcmc` preset, gsplat_trainer.py:975-983 and :753-761)
# ------------------------------------------------------------------------------------------------
def compute_relocation(opacities, scales, ratios, n_max=51):
    """Plain loops in float64: opacity and scale of a Gaussian split `ratio` ways."""
    N = opacities.shape[0]
    new_op = torch.zeros(N, dtype=torch.float64)
    new_sc = torch.zeros(N, 3, dtype=torch.float64)
    for i in range(N):
        n = int(min(max(int(ratios[i]), 1), n_max))
        o = float(opacities[i])
        no = 1.0 - (1.0 - o) ** (1.0 / n)
        denom = 0.0
        for a in range(1, n + 1):
            for k in range(a):
                denom += math.comb(a - 1, k) * ((-1.0) ** k) / math.sqrt(k + 1) * no ** (k + 1)
        new_op[i] = no
        new_sc[i] = scales[i].double() * (o / denom)
    return new_op, new_sc


def inject_noise(means, log_scales, quats, logit_opac, noise, scaler):
    """means + Sigma (noise * sigmoid_100((1-opacity) - 0.995) * scaler) in float64."""
    from oracle.torch_oracle import quat_scale_to_covar
    cov = quat_scale_to_covar(quats.double(), torch.exp(log_scales.double()))
    op = torch.sigmoid(logit_opac.double())
    g = 1.0 / (1.0 + torch.exp(-100.0 * ((1.0 - op) - 0.995)))
    return means.double() + torch.einsum("bij,bj->bi", cov, noise.double() * g[:, None] * scaler)
