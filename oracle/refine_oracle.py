"""CPU restatement (numpy, float64) of one DefaultStrategy refinement -- TEST INFRASTRUCTURE (see oracle/__init__.py).

What it restates: the densification step the reference drives through `gsplat.strategy.DefaultStrategy`
(/root/reference/utils/gsplat_utils/gsplat_trainer.py:129-131, 345-350, 744-763; defaults and procedure SURVEY.md
section 8 a11 / B.3, [upstream-memory] of gsplat ~v1.4 -- the source of that dependency is absent, parity unpinned):

    grow:   avg = grad2d / max(count, 1);  high = avg > grow_grad2d;  small = max exp(log s) <= grow_scale3d scene_scale
            duplicate high & small (copies appended, Adam moments zero);
            split high & ~small   (each replaced by two samples mu + R (s * z), z ~ N(0, I); scales / 1.6; moments zero)
    prune:  on the grown set: sigmoid(opacity logit) < prune_opa, and once step > reset_every also
            max exp(log s) > prune_scale3d scene_scale
    order:  gsplat's torch.cat order -- [originals that were not split | duplicates | first children | second children]

`refine_masks_np` is the vectorised form of `oracle.strategy_oracle.refine_masks` (plain Python loops, small N);
tests/test_refine_oracle.py checks one against the other.

The split noise: the device has no host generator in the loop, so z is DEFINED as a counter-based function of
(seed, step, source row, child): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11) with key = seed, counter = (row, child, step, 0x53504C54), then Box-Muller on 24-bit uniforms.
`philox4x32_10` here is written from the paper (its known-answer vectors are checked in the same test file); nothing
is shared with the product's csrc/so_rng.hpp.
"""
from __future__ import annotations

import numpy as np

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
SPLIT_STREAM = 0x53504C54
MASK32 = 0xFFFFFFFF


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter [...,4], key [...,2] (any integer dtype, values < 2^32) -> [...,4] uint64 holding 32-bit words."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) & MASK32 for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64) & MASK32
    k1 = np.asarray(key[..., 1], dtype=np.uint64) & MASK32
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c[0]
        p1 = np.uint64(PHILOX_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(PHILOX_W0)) & MASK32
        k1 = (k1 + np.uint64(PHILOX_W1)) & MASK32
    return np.stack(c, axis=-1)


def _u24(x: np.ndarray) -> np.ndarray:
    return ((x >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0


def split_normals(seed: int, step: int, ids: np.ndarray, child: int) -> np.ndarray:
    """[n,3] standard normals of child `child` (0 / 1) of the source rows `ids` at training step `step`."""
    ids = np.asarray(ids, dtype=np.uint64)
    n = ids.shape[0]
    ctr = np.stack([ids, np.full(n, child, np.uint64), np.full(n, step, np.uint64), np.full(n, SPLIT_STREAM, np.uint64)], -1)
    key = np.stack([np.full(n, seed & MASK32, np.uint64), np.full(n, (seed >> 32) & MASK32, np.uint64)], -1)
    r = philox4x32_10(ctr, key)
    u0, u1, u2, u3 = (_u24(r[:, i]) for i in range(4))
    r0, r1 = np.sqrt(-2.0 * np.log(u0)), np.sqrt(-2.0 * np.log(u2))
    return np.stack([r0 * np.cos(2 * np.pi * u1), r0 * np.sin(2 * np.pi * u1), r1 * np.cos(2 * np.pi * u3)], -1)


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


def refine_masks_np(grad2d, count, log_scales, opac_logits, step, scene_scale, grow_grad2d=0.0002, grow_scale3d=0.01,
                    prune_opa=0.005, prune_scale3d=0.1, reset_every=3000):
    """-> (is_dupli[N], is_split[N], prune_self[N], smax[N]) -- the decisions of `strategy_oracle.refine_masks`, vectorised."""
    avg = grad2d / np.maximum(count, 1.0)
    high = avg > grow_grad2d
    smax = np.exp(log_scales.max(axis=-1))
    small = smax <= grow_scale3d * scene_scale
    prune = 1.0 / (1.0 + np.exp(-opac_logits)) < prune_opa
    if step > reset_every:
        prune = prune | (smax > prune_scale3d * scene_scale)
    return high & small, high & ~small, prune, smax


def refine_default(params: dict, exp_avg: dict, exp_avg_sq: dict, grad2d, count, *, step: int, scene_scale: float, seed: int,
                   grow_grad2d=0.0002, grow_scale3d=0.01, prune_opa=0.005, prune_scale3d=0.1, reset_every=3000,
                   revised_opacity=False):
    """One refinement on float64 copies of the model.  params / exp_avg / exp_avg_sq: dicts over
    means[N,3], scales[N,3] (log), quats[N,4], opacities[N] (logit), sh0[N,1,3], shN[N,K-1,3].
    Returns (params', exp_avg', exp_avg_sq', report) in gsplat's row order."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    M = {k: np.asarray(v, dtype=np.float64) for k, v in exp_avg.items()}
    V = {k: np.asarray(v, dtype=np.float64) for k, v in exp_avg_sq.items()}
    g2, cn = np.asarray(grad2d, np.float64), np.asarray(count, np.float64)
    N = P["means"].shape[0]
    dup, spl, prune_self, smax = refine_masks_np(g2, cn, P["scales"], P["opacities"], step, scene_scale, grow_grad2d,
                                                  grow_scale3d, prune_opa, prune_scale3d, reset_every)
    # the children, as the grown set holds them: scales / 1.6, opacity unchanged unless revised
    op = 1.0 / (1.0 + np.exp(-P["opacities"]))
    child_op = 1.0 - np.sqrt(1.0 - op) if revised_opacity else op
    prune_child = child_op < prune_opa
    if step > reset_every:
        prune_child = prune_child | (smax / 1.6 > prune_scale3d * scene_scale)
    keepA = ~spl & ~prune_self
    keepB = dup & ~prune_self
    keepC = spl & ~prune_child
    iA, iB, iC = np.nonzero(keepA)[0], np.nonzero(keepB)[0], np.nonzero(keepC)[0]
    R = quat_to_rotmat(P["quats"][iC])
    s = np.exp(P["scales"][iC])
    out_p, out_m, out_v = {}, {}, {}
    for k in P:
        src = P[k]
        kids = [src[iC].copy(), src[iC].copy()]
        for child in (0, 1):
            if k == "means":
                z = split_normals(seed, step, iC, child)
                kids[child] = src[iC] + np.einsum("nij,nj->ni", R, s * z)
            elif k == "scales":
                kids[child] = np.log(np.exp(src[iC]) / 1.6)
            elif k == "opacities" and revised_opacity:
                o = 1.0 - np.sqrt(1.0 - 1.0 / (1.0 + np.exp(-src[iC])))
                kids[child] = np.log(o / (1.0 - o))
        out_p[k] = np.concatenate([src[iA], src[iB], kids[0], kids[1]])
        zeros = np.zeros((len(iB) + 2 * len(iC),) + src.shape[1:])
        out_m[k] = np.concatenate([M[k][iA], zeros])
        out_v[k] = np.concatenate([V[k][iA], zeros])
    n_new = len(iA) + len(iB) + 2 * len(iC)
    report = {"n_dupli": int(dup.sum()), "n_split": int(spl.sum()),
              "n_prune": int(N + dup.sum() + spl.sum() - n_new), "n_new": int(n_new), "n_old": int(N)}
    return out_p, out_m, out_v, report
