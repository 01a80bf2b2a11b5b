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
