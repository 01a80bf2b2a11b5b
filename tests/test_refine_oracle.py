"""The refinement oracle (oracle/refine_oracle.py) is pinned before it is trusted:
  * its Philox4x32-10 against the published known-answer vectors of Random123 (Salmon et al., SC'11);
  * its vectorised decisions against the loop-level restatement oracle/strategy_oracle.py::refine_masks;
  * structural properties of a refinement (row accounting, order, zeroed moments, children statistics).
And the PRODUCT's generator (splat_one_amd/csrc/so_rng.hpp, built for the host by tests/host_harness) against the
oracle's: integer stream bit for bit, normals within float32 rounding."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import refine_oracle as RO
from oracle import strategy_oracle as SO
from tests.host_harness.build import build

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
KAT = [((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = RO.philox4x32_10(np.array([ctr], dtype=np.uint64), np.array([key], dtype=np.uint64))[0]
        assert tuple(int(v) for v in got) == want, (ctr, [hex(int(v)) for v in got])


@pytest.fixture(scope="module")
def hh():
    return ctypes.CDLL(build())


def test_product_rng_matches_the_oracle(hh):
    rng = np.random.default_rng(0)
    n = 4096
    ctr = rng.integers(0, 2 ** 32, size=(n, 4), dtype=np.uint64)
    key = np.array([0x9E3779B9, 0x12345678], dtype=np.uint64)
    for c, k, want in KAT:                       # the product's integer stream also meets the published vectors
        out = np.zeros(4, dtype=np.uint32)
        hh.hh_philox4x32_10(1, np.array(c, dtype=np.uint32).ctypes.data_as(ctypes.c_void_p),
                            np.array(k, dtype=np.uint32).ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        assert tuple(int(v) for v in out) == want
    c32, k32 = ctr.astype(np.uint32), key.astype(np.uint32)
    out = np.zeros((n, 4), dtype=np.uint32)
    hh.hh_philox4x32_10(n, c32.ctypes.data_as(ctypes.c_void_p), k32.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    want = RO.philox4x32_10(ctr, np.broadcast_to(key, (n, 2)))
    assert np.array_equal(out.astype(np.uint64), want)
    # normals: float32 Box-Muller of the product vs float64 of the oracle
    seed, step = 0x0123456789ABCDEF, 1700
    ids = rng.integers(0, 2 ** 30, size=n, dtype=np.uint64)
    for child in (0, 1):
        z = np.zeros((n, 3), dtype=np.float32)
        hh.hh_split_normals(n, ctypes.c_ulonglong(seed), ctypes.c_uint32(step), ids.astype(np.uint32).ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_uint32(child), z.ctypes.data_as(ctypes.c_void_p))
        zo = RO.split_normals(seed, step, ids, child)
        assert np.abs(z - zo).max() < 5e-6
    # and they ARE standard normals, independent between children / components
    z0 = RO.split_normals(7, 3, np.arange(200_000), 0)
    z1 = RO.split_normals(7, 3, np.arange(200_000), 1)
    assert abs(z0.mean()) < 5e-3 and abs(z0.std() - 1) < 5e-3 and abs(np.mean(z0 ** 4) - 3) < 5e-2
    assert abs(np.corrcoef(z0[:, 0], z1[:, 0])[0, 1]) < 1e-2 and abs(np.corrcoef(z0[:, 0], z0[:, 2])[0, 1]) < 1e-2
    assert not np.array_equal(RO.split_normals(7, 4, np.arange(16), 0), z0[:16])       # the step is part of the counter


def _model(N, seed=0, K=4):
    g = np.random.default_rng(seed)
    base = np.exp(g.uniform(np.log(0.002), np.log(0.3), size=(N, 1)))      # sizes on both sides of every threshold
    P = {"means": g.normal(size=(N, 3)), "scales": np.log(base * g.uniform(0.6, 1.0, size=(N, 3))), "quats": g.normal(size=(N, 4)),
         "opacities": g.normal(size=N) * 3 - 1, "sh0": g.normal(size=(N, 1, 3)), "shN": g.normal(size=(N, K - 1, 3))}
    M = {k: g.normal(size=v.shape) for k, v in P.items()}
    V = {k: g.uniform(size=v.shape) for k, v in P.items()}
    grad2d = g.uniform(0, 6e-4, size=N) * g.integers(1, 5, size=N)
    count = g.integers(0, 5, size=N).astype(np.float64)
    return P, M, V, grad2d, count


@pytest.mark.parametrize("step", [600, 3100])
def test_vectorised_masks_equal_the_loop_restatement(step):
    N = 700
    P, M, V, g2, cn = _model(N, seed=step)
    dup, spl, prune_self, _ = RO.refine_masks_np(g2, cn, P["scales"], P["opacities"], step, 1.3)
    d_l, s_l, prune_fn = SO.refine_masks(torch.from_numpy(g2), torch.from_numpy(cn), torch.from_numpy(P["scales"]),
                                         torch.from_numpy(P["opacities"]), step, 1.3)
    assert np.array_equal(dup, d_l.numpy()) and np.array_equal(spl, s_l.numpy())
    assert np.array_equal(prune_self, prune_fn(torch.from_numpy(P["scales"]), torch.from_numpy(P["opacities"])).numpy())
    assert dup.sum() > 10 and spl.sum() > 10 and prune_self.sum() > 10 and not (dup & spl).any()
    # the grown set pruned by the loop version == what refine_default keeps
    outP, outM, outV, rep = RO.refine_default(P, M, V, g2, cn, step=step, scene_scale=1.3, seed=5)
    grown_scales = np.concatenate([P["scales"][~spl], P["scales"][dup], np.log(np.exp(P["scales"][spl]) / 1.6),
                                   np.log(np.exp(P["scales"][spl]) / 1.6)])
    grown_opac = np.concatenate([P["opacities"][~spl], P["opacities"][dup], P["opacities"][spl], P["opacities"][spl]])
    keep = ~prune_fn(torch.from_numpy(grown_scales), torch.from_numpy(grown_opac)).numpy()
    assert rep["n_new"] == int(keep.sum()) and rep["n_dupli"] == int(dup.sum()) and rep["n_split"] == int(spl.sum())
    assert rep["n_prune"] == int((~keep).sum())
    assert np.allclose(outP["scales"], grown_scales[keep]) and np.allclose(outP["opacities"], grown_opac[keep])


def test_refinement_structure():
    N = 3000
    P, M, V, g2, cn = _model(N, seed=1)
    outP, outM, outV, rep = RO.refine_default(P, M, V, g2, cn, step=700, scene_scale=1.0, seed=11)
    dup, spl, prune, _ = RO.refine_masks_np(g2, cn, P["scales"], P["opacities"], 700, 1.0)
    nA = int((~spl & ~prune).sum())
    # segment A: survivors in source order, moments carried over
    for k in P:
        assert np.array_equal(outP[k][:nA], P[k][~spl & ~prune]) and np.array_equal(outM[k][:nA], M[k][~spl & ~prune])
        assert not outM[k][nA:].any() and not outV[k][nA:].any()           # every new row starts with zero moments
    nB = int((dup & ~prune).sum())
    assert np.array_equal(outP["means"][nA:nA + nB], P["means"][dup & ~prune])
    nC = (rep["n_new"] - nA - nB) // 2
    src = np.nonzero(spl & ~prune)[0]            # step 700 <= reset_every: children are pruned like their parents
    assert nC == len(src)
    c0, c1 = outP["means"][nA + nB:nA + nB + nC], outP["means"][nA + nB + nC:]
    # children: displaced by R (s * z): in the parent's frame, z recovered exactly
    R = RO.quat_to_rotmat(P["quats"][src])
    s = np.exp(P["scales"][src])
    for child, c in ((0, c0), (1, c1)):
        z = np.einsum("nji,nj->ni", R, c - P["means"][src]) / s
        assert np.allclose(z, RO.split_normals(11, 700, src, child), atol=1e-9)
    assert np.allclose(outP["scales"][nA + nB:nA + nB + nC], P["scales"][src] - np.log(1.6))
    # revised opacity: two children together cover what the parent covered: 1 - (1 - o')^2 = o
    outP2, _, _, _ = RO.refine_default(P, M, V, g2, cn, step=700, scene_scale=1.0, seed=11, revised_opacity=True, prune_opa=0.0)
    src2 = np.nonzero(spl)[0]
    o_child = 1 / (1 + np.exp(-outP2["opacities"][-len(src2):]))
    o_par = 1 / (1 + np.exp(-P["opacities"][src2]))
    assert np.allclose(1 - (1 - o_child) ** 2, o_par)
