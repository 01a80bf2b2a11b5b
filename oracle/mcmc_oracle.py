"""CPU restatement (numpy) of the device-side MCMCStrategy refinement -- TEST INFRASTRUCTURE (see oracle/__init__.py).

What it follows: gsplat's `relocate` / `sample_add` / `inject_noise_to_position` as the reference drives them from its
`mcmc` preset (/root/reference/utils/gsplat_utils/gsplat_trainer.py:975-983, :753-761) [upstream-memory: gsplat ~1.4
strategy/ops.py -- the gsplat source is not in /root/reference, SURVEY.md section 0], with the ONE deliberate difference
of the device path (include/splat_one_amd.h, so_mcmc_refine): a draw is a counter-based function of
(seed, step, phase, sample number) -- Philox4x32-10 -> 64-bit uniform -> inverse CDF over the float64 running sum of the
opacities -- instead of torch.multinomial on a host-seeded generator.  Parity unpinned (no fixture of the reference
covers it); the relocation formula itself is pinned by tests/test_gpu_mcmc.py::test_compute_relocation_matches_oracle
against oracle/strategy_oracle.py's loops.
"""
import math

import numpy as np

from oracle.refine_oracle import MASK32, philox4x32_10, _u24

SAMPLE_STREAM = 0x53414D50
NOISE_STREAM = 0x4E4F4953
ORDER = ("means", "scales", "quats", "opacities", "sh0", "shN")


def sigmoid32(x):
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def uniforms(seed: int, step: int, phase: int, n: int) -> np.ndarray:
    """u_j in [0, 1), float64, exactly as the kernel forms them: (x0 * 2^32 + x1) * 2^-64 in double arithmetic."""
    j = np.arange(n, dtype=np.uint64)
    ctr = np.stack([j, np.full(n, phase, np.uint64), np.full(n, step, np.uint64), np.full(n, SAMPLE_STREAM, np.uint64)], -1)
    key = np.stack([np.full(n, seed & MASK32, np.uint64), np.full(n, (seed >> 32) & MASK32, np.uint64)], -1)
    r = philox4x32_10(ctr, key)
    return (r[:, 0].astype(np.float64) * 4294967296.0 + r[:, 1].astype(np.float64)) * (1.0 / 18446744073709551616.0)


def draw(weights: np.ndarray, n: int, seed: int, step: int, phase: int) -> np.ndarray:
    """n rows drawn in proportion to `weights` (float32, zero = never drawn): smallest i with cdf[i] > u W."""
    cdf = np.cumsum(weights.astype(np.float64))
    W = cdf[-1] if cdf.size else 0.0
    t = uniforms(seed, step, phase, n) * W
    idx = np.searchsorted(cdf, t, side="right")
    idx = np.minimum(idx, cdf.size - 1)
    prev = np.concatenate([[0.0], cdf[:-1]])
    for k in range(idx.size):                 # t rounded up to W: back to the last row that carries weight
        while idx[k] > 0 and not cdf[idx[k]] > prev[idx[k]]:
            idx[k] -= 1
    return idx.astype(np.int64)


def relocation(op, scales, ratios, n_max=51):
    """float64 loops of 3DGS-MCMC eq. 9 (same as oracle/strategy_oracle.compute_relocation, numpy)"""
    new_op = np.zeros(len(op))
    coeff = np.zeros(len(op))
    for i in range(len(op)):
        n = int(min(max(int(ratios[i]), 1), n_max))
        o = float(op[i])
        no = 1.0 - (1.0 - o) ** (1.0 / n)
        denom = 0.0
        for a in range(1, n + 1):
            for k in range(a):
                denom += math.comb(a - 1, k) * ((-1.0) ** k) / math.sqrt(k + 1) * no ** (k + 1)
        new_op[i], coeff[i] = no, o / denom
    return new_op, scales.astype(np.float64) * coeff[:, None]


def _new_values(P, src, min_opacity):
    op = sigmoid32(P["opacities"][src]).astype(np.float64)
    ratios = np.bincount(src, minlength=len(P["opacities"]))[src] + 1
    new_op, new_sc = relocation(op, np.exp(P["scales"][src].astype(np.float64)), ratios)
    new_op = np.clip(new_op, min_opacity, 1.0 - np.finfo(np.float32).eps)
    return np.log(new_op / (1.0 - new_op)), np.log(new_sc)


def relocate(P, M, V, *, min_opacity, seed, step, weights=None):
    """-> (P', M', V', sources, dead rows).  weights: the per-row float32 opacities the device formed (its expf and numpy's
    differ in the last bit, which moves a boundary of the CDF by 1e-7 of its width); None: computed here."""
    w = sigmoid32(P["opacities"]) if weights is None else np.asarray(weights, dtype=np.float32).copy()
    if weights is None:
        w[w <= np.float32(min_opacity)] = 0.0
    dead = np.nonzero(w == 0.0)[0]
    P, M, V = ({k: v.copy() for k, v in d.items()} for d in (P, M, V))
    if len(dead) == 0 or not w.sum() > 0:
        return P, M, V, np.zeros(0, np.int64), dead
    src = draw(w, len(dead), seed, step, 0)
    new_logit, new_ls = _new_values(P, src, min_opacity)
    P["opacities"][src] = new_logit
    P["scales"][src] = new_ls
    for k in ORDER:
        P[k][dead] = P[k][src]
        M[k][src] = 0
        V[k][src] = 0
    return P, M, V, src, dead


def sample_add(P, M, V, *, min_opacity, cap_max, seed, step, weights=None):
    N = len(P["opacities"])
    n_add = max(0, min(cap_max, int(1.05 * N)) - N)
    P, M, V = ({k: v.copy() for k, v in d.items()} for d in (P, M, V))
    if n_add == 0:
        return P, M, V, np.zeros(0, np.int64)
    w = sigmoid32(P["opacities"]) if weights is None else np.asarray(weights, dtype=np.float32)
    src = draw(w, n_add, seed, step, 1)
    new_logit, new_ls = _new_values(P, src, min_opacity)
    P["opacities"][src] = new_logit
    P["scales"][src] = new_ls
    for k in ORDER:
        P[k] = np.concatenate([P[k], P[k][src]])
        M[k] = np.concatenate([M[k], np.zeros_like(M[k][src])])
        V[k] = np.concatenate([V[k], np.zeros_like(V[k][src])])
    return P, M, V, src


def noise_normals(seed: int, step: int, n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i, np.zeros(n, np.uint64), np.full(n, step, np.uint64), np.full(n, NOISE_STREAM, np.uint64)], -1)
    key = np.stack([np.full(n, seed & MASK32, np.uint64), np.full(n, (seed >> 32) & MASK32, np.uint64)], -1)
    r = philox4x32_10(ctr, key)
    u0, u1, u2, u3 = (_u24(r[:, k]) for k in range(4))
    r0, r1 = np.sqrt(-2.0 * np.log(u0)), np.sqrt(-2.0 * np.log(u2))
    return np.stack([r0 * np.cos(2 * np.pi * u1), r0 * np.sin(2 * np.pi * u1), r1 * np.cos(2 * np.pi * u3)], -1)
