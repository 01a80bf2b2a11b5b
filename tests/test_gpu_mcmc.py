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


# ------------------------------------------------------------------------------------------------ MCMCStrategy on the device
ORDER = ("means", "scales", "quats", "opacities", "sh0", "shN")


def _random_model(N, K=16, seed=0, n_dead=0):
    import numpy as np
    rng = np.random.default_rng(seed)
    P = {"means": rng.normal(size=(N, 3)), "scales": np.log(rng.uniform(0.01, 0.2, (N, 3))), "quats": rng.normal(size=(N, 4)),
         "opacities": rng.normal(size=N) * 1.5, "sh0": rng.normal(size=(N, 1, 3)), "shN": rng.normal(size=(N, K - 1, 3))}
    P = {k: v.astype(np.float32) for k, v in P.items()}
    if n_dead:
        P["opacities"][rng.choice(N, n_dead, replace=False)] = -9.0
    M = {k: rng.normal(size=v.shape).astype(np.float32) for k, v in P.items()}
    V = {k: rng.uniform(0.1, 1, v.shape).astype(np.float32) for k, v in P.items()}
    return P, M, V


def _device_sets(dev, P, M, V, cap):
    import numpy as np
    from splat_one_amd import _lib
    N = P["means"].shape[0]
    sets = []
    for src in (P, M, V):
        d = {}
        for k in ORDER:
            t = torch.full((cap,) + src[k].shape[1:], float("nan"), dtype=torch.float32, device=dev)
            t[:N] = torch.from_numpy(src[k]).to(dev)
            d[k] = t
        sets.append(d)
    ms = _lib.ModelSet()
    for i, k in enumerate(ORDER):
        ms.p[i], ms.m[i], ms.v[i] = (s[k].data_ptr() for s in sets)
    return sets, ms


@pytest.mark.parametrize("N,n_dead,cap_max,cap", [(20_000, 700, 10 ** 6, 1 << 17), (5_000, 0, 5_100, 1 << 17), (1_000, 990, 10 ** 6, 1 << 17),
                                                  (70_001, 3_000, 72_000, 1 << 17),
                                                  # an ODD capacity and an ODD number of scan blocks (73): the float64 regions of
                                                  # the scratch must stay 8-byte aligned (`tot` was not)
                                                  (70_001, 3_000, 72_000, 74_751)])
def test_mcmc_refine_on_the_device_equals_the_oracle(dev, N, n_dead, cap_max, cap):
    """so_mcmc_refine (relocate, then sample_add, one phase per call so that the weights the device formed can be read
    back) against oracle/mcmc_oracle.py: the SAME rows drawn, parameters of sources / relocated / appended rows, moments."""
    import ctypes
    import numpy as np
    from oracle import mcmc_oracle as MO
    from splat_one_amd import _lib
    from splat_one_amd.strategy import MCMCStrategy
    P, M, V = _random_model(N, seed=N, n_dead=n_dead)
    sets, ms = _device_sets(dev, P, M, V, cap)
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev)
    binoms = MCMCStrategy().initialize_state()["binoms"].to(dev).contiguous()
    scratch = torch.zeros(int(_lib.load().so_mcmc_scratch_words(cap)), dtype=torch.int32, device=dev)
    report = torch.zeros(8, dtype=torch.int32, device=dev)
    seed, step, min_op = 0xABCDEF0123, 700, 0.005

    def run(skip_bits):
        prm = _lib.McmcParams(min_op, cap_max, seed, step, skip_bits)
        _lib.call("so_mcmc_refine", cap, 16, ctypes.byref(ms), _lib.ptr(n_dev), _lib.ptr(binoms), 51, ctypes.byref(prm),
                  _lib.ptr(scratch), _lib.ptr(report), _lib.stream())
        torch.cuda.synchronize()

    def host(n):
        return [{k: s[k][:n].cpu().numpy() for k in ORDER} for s in sets]

    # phase 0: relocation
    run(2)
    w0 = scratch[:N].view(torch.float32).cpu().numpy().copy()
    P1o, M1o, V1o, src_o, dead_o = MO.relocate(P, M, V, min_opacity=min_op, seed=seed, step=step, weights=w0)
    nd = len(dead_o)            # (the planted dead rows plus the odd random logit below logit(min_opacity))
    assert n_dead <= nd <= n_dead + 20 and int(n_dev) == N and report[0].item() == nd
    cap2 = cap + (cap & 1)
    src_d = scratch[cap2 + 3 * cap:cap2 + 3 * cap + nd].cpu().numpy()
    assert np.array_equal(src_d, src_o)                                   # the same rows drawn
    P1, M1, V1 = host(N)
    for k in ORDER:
        tol = 2e-5 if k in ("scales", "opacities") else 0.0              # (relocation formula in float32 on the device)
        assert np.allclose(P1[k], P1o[k], rtol=tol, atol=tol), k
        assert np.array_equal(M1[k], M1o[k]) and np.array_equal(V1[k], V1o[k]), k
    assert not scratch[cap2 + 4 * cap:cap2 + 5 * cap].any()               # the draw counters are left zero
    # phase 1: addition, on the relocated model
    run(1)
    n_add = max(0, min(cap_max, int(1.05 * N)) - N)
    assert int(n_dev) == N + n_add and report[1].item() == n_add and report[3].item() == N + n_add
    w1 = scratch[:N].view(torch.float32).cpu().numpy().copy()
    P2o, M2o, V2o, src2_o = MO.sample_add(P1, M1, V1, min_opacity=min_op, cap_max=cap_max, seed=seed, step=step, weights=w1)
    src2_d = scratch[cap2 + 3 * cap:cap2 + 3 * cap + n_add].cpu().numpy()
    assert np.array_equal(src2_d, src2_o)
    P2, M2, V2 = host(N + n_add)
    for k in ORDER:
        tol = 2e-5 if k in ("scales", "opacities") else 0.0
        assert P2[k].shape == P2o[k].shape and np.allclose(P2[k], P2o[k], rtol=tol, atol=tol), k
        assert np.array_equal(M2[k], M2o[k]) and np.array_equal(V2[k], V2o[k]), k
    if n_add:
        assert not M2["shN"][N:].any() and np.array_equal(P2["means"][N:], P2["means"][src2_o])
    # rows beyond the new N were never touched
    assert torch.isnan(sets[0]["means"][N + n_add:]).all()
    # without the device's weights the oracle's own float32 opacities give the same draw up to CDF-boundary ties
    _, _, _, src_own, _ = MO.relocate(P, M, V, min_opacity=min_op, seed=seed, step=step)
    assert len(src_own) == nd and (src_own != src_o).sum() <= max(1, 2e-3 * max(1, nd))


def test_device_noise_matches_the_oracle(dev):
    """so_inject_noise_dev: normals from (seed, optimiser step, row), scaler = lr0 gamma^step noise_lr evaluated on the device"""
    import numpy as np
    from oracle import mcmc_oracle as MO
    from splat_one_amd import _lib
    g = torch.Generator().manual_seed(1)
    N, cap = 3000, 4096
    means = torch.randn(N, 3, generator=g)
    ls = torch.log(torch.rand(N, 3, generator=g) * 0.3 + 0.01)
    quats = torch.randn(N, 4, generator=g)
    lo = torch.logit(torch.rand(N, generator=g) * 0.02 + 1e-4)
    lo[::3] = 2.0
    seed, t, lr0, gamma, noise_lr = 77, 1234, 1.6e-4, 0.99985, 5e5
    z = torch.from_numpy(MO.noise_normals(seed, t, N))
    ref = SO.inject_noise(means, ls, quats, lo, z, lr0 * gamma ** t * noise_lr)
    pad = lambda x: torch.cat([x, torch.full((cap - N,) + x.shape[1:], float("nan"))]).to(dev).contiguous()
    m, a_ls, a_q, a_lo = pad(means), pad(ls), pad(quats), pad(lo)
    n_dev = torch.tensor([N], dtype=torch.int32, device=dev)
    ctr = torch.tensor([t, 0], dtype=torch.int32, device=dev)
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    args = lambda: (cap, _lib.ptr(n_dev), _lib.ptr(m), _lib.ptr(a_ls), _lib.ptr(a_q), _lib.ptr(a_lo), seed, _lib.ptr(ctr), lr0, gamma,
                    noise_lr, _lib.ptr(skip), _lib.stream())
    _lib.call("so_inject_noise_dev", *args())
    got = m[:N].cpu().double() - means.double()
    assert rel_err(got, ref - means.double()) < 2e-4 and torch.isnan(m[N:]).all()
    skip.fill_(1)                                    # a void iteration: nothing moves
    before = m.clone()
    _lib.call("so_inject_noise_dev", *args())
    assert torch.equal(m[:N], before[:N])


def test_mcmc_engine_refines_without_touching_the_host(dev):
    """The fused Runner with MCMCStrategy keeps the model on the device: refinements change N there, the captured step
    follows, and the host's handles are re-pointed only when somebody looks (len(runner.splats[...]))."""
    W, H = 128, 96
    r = _mcmc_runner(dev, True)
    c2w, Ks = ring_cameras(4).to(dev), pinhole_K(W, H)[None].to(dev)
    target = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        r.splats["opacities"][:150] = -9.0
    for step in range(25):
        r.train_step(c2w[step % 4:step % 4 + 1], Ks, target)
    eng = r._engine
    assert eng.device_refine and eng.model_sets == 1 and eng.refinements == 1 and eng._host_stale      # nobody looked yet
    graphs_before = len(eng._graphs)
    for step in range(25, 45):
        r.train_step(c2w[step % 4:step % 4 + 1], Ks, target)
    assert eng.refinements == 3 and len(eng._graphs) == graphs_before                                   # no re-capture
    rep = eng.refine_report()
    assert rep["n_new"] == 3400 and len(r.splats["means"]) == 3400 and not eng._host_stale
    assert (torch.sigmoid(r.splats["opacities"]) >= 0.004).all() and torch.isfinite(r.splats["means"]).all()
    st = r.optimizers["shN"].state[r.splats["shN"]]
    assert st["exp_avg"].shape[0] == 3400 and torch.isfinite(st["exp_avg"]).all()
