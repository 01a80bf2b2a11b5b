"""MCMC densification strategy (row f2): the two HIP kernels against the loop-level float64 oracle, the
structural edits, and a short training run through both step implementations."""
import math

import pytest
import torch

from oracle import strategy_oracle as SO
from splat_one_amd.scene import pinhole_K, ring_cameras
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def test_compute_relocation_matches_oracle(dev):
    from splat_one_amd.strategy import MCMCStrategy, compute_relocation
    g = torch.Generator().manual_seed(0)
    N = 400
    op = torch.rand(N, generator=g) * 0.98 + 0.01
    sc = torch.rand(N, 3, generator=g) * 0.2 + 0.01
    ratios = torch.randint(1, 60, (N,), generator=g)          # > n_max gets clamped to 51
    binoms = MCMCStrategy().initialize_state()["binoms"]
    no_h, ns_h = compute_relocation(op.to(dev), sc.to(dev), ratios.to(dev), binoms.to(dev))
    no_o, ns_o = SO.compute_relocation(op, sc, ratios)
    assert rel_err(no_h, no_o) < 1e-5
    small = ratios <= 12                                       # alternating binomial sums: fp32 cancels for large n
    assert rel_err(ns_h.cpu()[small], ns_o[small]) < 1e-3
    assert torch.isfinite(ns_h).all()
    # n = 1 is the identity
    one = torch.ones(N, dtype=torch.int64)
    no1, ns1 = compute_relocation(op.to(dev), sc.to(dev), one.to(dev), binoms.to(dev))
    assert torch.allclose(no1.cpu(), op, rtol=1e-5) and torch.allclose(ns1.cpu(), sc, rtol=1e-5)


def test_inject_noise_matches_oracle(dev):
    from splat_one_amd import _lib
    g = torch.Generator().manual_seed(1)
    N = 1000
    means = torch.randn(N, 3, generator=g)
    ls = torch.log(torch.rand(N, 3, generator=g) * 0.3 + 0.01)
    quats = torch.randn(N, 4, generator=g)
    lo = torch.logit(torch.rand(N, generator=g) * 0.02 + 1e-4)     # low opacities: that is where the noise acts
    lo[::3] = 2.0                                                  # opaque ones receive ~no noise
    noise = torch.randn(N, 3, generator=g)
    scaler = 80.0
    ref = SO.inject_noise(means, ls, quats, lo, noise, scaler)
    m = means.clone().to(dev)
    args = [t.to(dev).contiguous() for t in (ls, quats, lo, noise)]
    _lib.call("so_inject_noise", N, _lib.ptr(m), *[_lib.ptr(t) for t in args], scaler, _lib.stream())
    assert rel_err(m - means.to(dev), ref - means.double()) < 1e-4
    assert (m.cpu() - means)[::3].abs().max() < 1e-6 * (m.cpu() - means).abs().max() + 1e-12


def _mcmc_runner(dev, fused, N=3000):
    from splat_one_amd.strategy import MCMCStrategy
    from splat_one_amd.trainer import Config, Runner
    strat = MCMCStrategy(refine_start_iter=10, refine_every=10, refine_stop_iter=1000, cap_max=3400)
    cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, opacity_reg=0.01, scale_reg=0.01, shN_init_std=0.05,
                 sh_degree_interval=10, max_steps=200, strategy=strat, fused=fused)
    return Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)


@pytest.mark.parametrize("fused", [False, True])
def test_mcmc_training_steps(dev, fused):
    W, H = 128, 96
    r = _mcmc_runner(dev, fused)
    c2w = ring_cameras(4).to(dev)
    Ks = pinhole_K(W, H)[None].to(dev)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    target = torch.stack([xx, yy, 0.5 * (xx + yy)], -1)[None].to(dev)
    with torch.no_grad():
        r.splats["opacities"][:200] = -9.0            # dead Gaussians: must be relocated at the first refine
    sizes = []
    for step in range(45):
        loss = r.train_step(c2w[step % 4:step % 4 + 1], Ks, target)
        sizes.append(len(r.splats["means"]))
    assert torch.isfinite(loss).all()
    # +5 % per refine step (20, 30, 40) until cap_max
    assert sizes[19] == 3000 and sizes[21] == 3150 and sizes[31] == 3307 and sizes[41] == 3400 and sizes[-1] == 3400
    assert (torch.sigmoid(r.splats["opacities"]) >= 0.004).all()      # nothing stays dead after relocation
    n = sizes[-1]
    for k, p in r.splats.items():
        assert p.shape[0] == n and torch.isfinite(p).all(), k
        assert r.optimizers[k].state[p]["exp_avg"].shape == p.shape


def test_mcmc_relocate_and_sample_add_semantics(dev):
    from splat_one_amd.strategy import MCMCStrategy, relocate, sample_add
    r = _mcmc_runner(dev, False, N=500)
    s = r.cfg.strategy
    binoms = s.initialize_state()["binoms"].to(dev)
    for k in r.splats:                        # give the optimisers a state
        r.splats[k].grad = torch.randn_like(r.splats[k])
        r.optimizers[k].step()
    with torch.no_grad():
        r.splats["opacities"][:50] = -9.0
    before = {k: v.detach().clone() for k, v in r.splats.items()}
    dead = torch.sigmoid(r.splats["opacities"]) <= s.min_opacity
    assert int(dead.sum()) == 50
    gen = torch.Generator(device=dev).manual_seed(3)
    relocate(r.splats, r.optimizers, {}, dead, binoms, s.min_opacity, generator=gen)
    gen2 = torch.Generator(device=dev).manual_seed(3)
    alive = (~dead).nonzero(as_tuple=True)[0]
    sampled = alive[torch.multinomial(torch.sigmoid(before["opacities"])[alive], 50, replacement=True, generator=gen2)]
    # dead rows are copies of their sampled sources (after the source was updated)
    assert torch.equal(r.splats["means"][:50], r.splats["means"][sampled])
    assert torch.equal(r.splats["means"][sampled], before["means"][sampled])
    assert torch.equal(r.splats["opacities"][:50], r.splats["opacities"][sampled])
    ratios = torch.bincount(sampled)[sampled] + 1
    exp_op = 1 - (1 - torch.sigmoid(before["opacities"][sampled])) ** (1.0 / ratios)
    assert torch.allclose(torch.sigmoid(r.splats["opacities"][sampled]), exp_op.clamp(min=s.min_opacity), rtol=1e-4, atol=1e-6)
    st = r.optimizers["means"].state[r.splats["means"]]
    assert st["exp_avg"][sampled].abs().max() == 0
    # sample_add grows by n with zero moments for the new rows
    sample_add(r.splats, r.optimizers, {}, 25, binoms, s.min_opacity, generator=gen)
    assert len(r.splats["means"]) == 525
    st = r.optimizers["shN"].state[r.splats["shN"]]
    assert st["exp_avg"].shape[0] == 525 and st["exp_avg"][500:].abs().max() == 0
