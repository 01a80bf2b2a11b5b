"""The reduce-scatter / sharded-Adam / all-gather step of the replicated data-parallel scheme (SURVEY.md section 8e;
hyper-parameters per /root/reference/utils/gsplat_utils/gsplat_trainer.py:266-278) on CPU over gloo with 2, 4 and 8
ranks and a Gaussian count that divides by nothing: after several steps every rank holds the parameters a
single-process `torch.optim.Adam` reaches on the MEAN of the ranks' gradients, bit-equal across ranks; the moments of a
piece live on its owner and `gather_moments` restores them everywhere (what a refinement needs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

ROWS = {"means": 3, "scales": 3, "quats": 4, "opacities": 1, "sh0": 3, "shN": 45}
LRS = {"means": 1.6e-4, "scales": 5e-3, "quats": 1e-3, "opacities": 5e-2, "sh0": 2.5e-3, "shN": 2.5e-3 / 20}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout(N):
    pad = lambda n: (n + 63) // 64 * 64
    segs, off = {}, 0
    for k, rl in ROWS.items():
        segs[k] = (off, N * rl)
        off += pad(N * rl)
    return segs, off


def _grads(N, rank, step):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return {k: torch.randn(N * rl, generator=g) * (1 + rank) for k, rl in ROWS.items()}


def _hyper(world):
    from splat_one_amd.trainer import adam_hyperparameters
    return {k: adam_hyperparameters(LRS[k], 1, world) for k in ROWS}


def _worker(local_rank, world_rank, world_size, args):
    out_dir, N, n_chunks, steps, void_at = args
    from splat_one_amd.distributed import ShardedFlatAdam
    segs, total = _layout(N)
    sa = ShardedFlatAdam(total, n_chunks=n_chunks)
    assert sa.padded_total >= total and sa.piece % 64 == 0 and sa.chunk == sa.piece * world_size
    P = torch.zeros(sa.padded_total)
    G = torch.zeros(sa.padded_total)
    M = torch.zeros(sa.padded_total)
    V = torch.zeros(sa.padded_total)
    g0 = torch.Generator().manual_seed(7)
    for k, (off, n) in segs.items():
        P[off:off + n] = torch.randn(n, generator=g0)
    hyper = _hyper(world_size)
    t = [0]

    def adam_fn(a, b, skip, gscale):
        # torch restatement of so_adam_step_scaled on the flat range [a, b): one group per tensor segment it intersects;
        # the SUMMED gradient times gscale (1 / world); nothing happens when the summed void flag is non-zero
        assert gscale == 1.0 / world_size and skip is not None
        if float(skip[0]) != 0.0:
            return
        for k, (off, n) in segs.items():
            lo, hi = max(a, off), min(b, off + n)
            if lo >= hi:
                continue
            lr, eps, (b1, b2) = hyper[k]
            g = G[lo:hi] * gscale
            M[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
            V[lo:hi].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (V[lo:hi].sqrt() / (1 - b2 ** t[0]) ** 0.5).add_(eps)
            P[lo:hi].addcdiv_(M[lo:hi], denom, value=-lr / (1 - b1 ** t[0]))

    done = 0
    for step in range(steps):
        G.zero_()
        for k, (off, n) in segs.items():
            G[off:off + n] = _grads(N, world_rank, step)[k]
        t[0] = done + 1
        # iteration `void_at`: the LAST rank's binning pass "overflowed" -- every rank must skip it (the flag is summed by
        # a reduce-scatter, not by a collective the host waits for) and every rank reads the same sum afterwards
        mine = torch.tensor([1.0 if (step == void_at and world_rank == world_size - 1) else 0.0])
        sa.step(G, P, adam_fn, void_src=mine)
        assert float(sa.void_flag()[0]) == (1.0 if step == void_at else 0.0)
        done += 0 if step == void_at else 1       # (the host takes a void iteration's step count back)
    m_own = M.clone()
    sa.gather_moments(M, V)
    mine = torch.zeros(sa.padded_total, dtype=torch.bool)
    for a, b in sa.my_ranges():
        mine[a:b] = True
    assert torch.equal(M[mine], m_own[mine]) and not m_own[~mine].any() and M[~mine].any()
    torch.save({"P": P, "M": M, "V": V, "bytes": sa.bytes_per_link_and_step()}, os.path.join(out_dir, f"r{world_rank}.pt"))


@pytest.mark.parametrize("world,n_chunks,N,void_at", [(2, 4, 1001, -1), (4, 4, 777, 1), (8, 5, 1234, -1)])
def test_sharded_adam_equals_adam_on_the_mean_gradient(tmp_path, world, n_chunks, N, void_at):
    from splat_one_amd import distributed as sdist
    steps = 3 if void_at < 0 else 4
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        sdist.cli(_worker, (str(tmp_path), N, n_chunks, steps, void_at), world_size=world, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for o in outs[1:]:
        assert torch.equal(o["P"], outs[0]["P"]) and torch.equal(o["M"], outs[0]["M"]) and torch.equal(o["V"], outs[0]["V"])
    # single-process reference: torch.optim.Adam per tensor on the mean gradient, the reference's hyper-parameter rule
    segs, total = _layout(N)
    hyper = _hyper(world)
    g0 = torch.Generator().manual_seed(7)
    params, opts = {}, {}
    for k, (off, n) in segs.items():
        params[k] = torch.nn.Parameter(torch.randn(n, generator=g0))
        lr, eps, betas = hyper[k]
        opts[k] = torch.optim.Adam([params[k]], lr=lr, eps=eps, betas=betas)
    for step in range(steps):
        if step == void_at:                      # the void iteration never happened
            continue
        for k in ROWS:
            params[k].grad = sum(_grads(N, r, step)[k] for r in range(world)) / world
            opts[k].step()
    for k, (off, n) in segs.items():
        got = outs[0]["P"][off:off + n]
        assert torch.allclose(got, params[k].detach(), rtol=1e-5, atol=1e-7), k
        st = opts[k].state[params[k]]
        assert torch.allclose(outs[0]["M"][off:off + n], st["exp_avg"], rtol=1e-5, atol=1e-6), k
    padded = outs[0]["P"].numel()
    assert abs(outs[0]["bytes"] - 2 * (world - 1) / world * padded * 4) < 1


# ------------------------------------------------------------------------------------------------ row pieces
def _row_worker(local_rank, world_rank, world_size, args):
    """RowShardedAdam on capacity-sized tensors: N live rows of CAP, N changing between steps (as after a refinement,
    with the moments gathered first)."""
    out_dir, cap, Ns, n_chunks, void_at = args
    from splat_one_amd.distributed import RowShardedAdam
    ra = RowShardedAdam(n_chunks=n_chunks)
    assert all(ra.span(N) <= cap and ra.chunk_rows(N) % (world_size * 64) == 0 for N in Ns)
    g0 = torch.Generator().manual_seed(7)
    P = {k: torch.randn(cap, rl, generator=g0) for k, rl in ROWS.items()}
    G = {k: torch.full((cap, rl), float("nan")) for k, rl in ROWS.items()}      # rows beyond N: never read
    M = {k: torch.zeros(cap, rl) for k, rl in ROWS.items()}
    V = {k: torch.zeros(cap, rl) for k, rl in ROWS.items()}
    hyper = _hyper(world_size)
    t = [0]
    touched = []

    def adam_fn(names, a, b, skip, gscale):
        touched.append((tuple(names), a, b))
        assert gscale == 1.0 / world_size and skip is not None
        if float(skip[0]) != 0.0:                # the void flags of all ranks, summed by chunk 0's reduce-scatter
            return
        for k in names:
            lr, eps, (b1, b2) = hyper[k]
            g = G[k][a:b] * gscale
            M[k][a:b].mul_(b1).add_(g, alpha=1 - b1)
            V[k][a:b].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (V[k][a:b].sqrt() / (1 - b2 ** t[0]) ** 0.5).add_(eps)
            P[k][a:b].addcdiv_(M[k][a:b], denom, value=-lr / (1 - b1 ** t[0]))

    done = 0
    for step, N in enumerate(Ns):
        if step and N != Ns[step - 1]:          # "refinement": every rank needs all moments of the old rows first
            ra.gather([M[k] for k in ROWS] + [V[k] for k in ROWS], Ns[step - 1])
            for k in ROWS:                      # the new rows start with zero moments, like duplicated / split Gaussians
                M[k][Ns[step - 1]:].zero_()
                V[k][Ns[step - 1]:].zero_()
        for k, rl in ROWS.items():
            G[k][:N] = _grads(N, world_rank, step)[k].view(N, rl)
            G[k][N:] = float("nan")             # rows of the last piece beyond N are summed with the rest and never used
        t[0] = done + 1
        mine = torch.tensor([2.0 if (step == void_at and world_rank == 0) else 0.0])
        before = len(touched)
        if step % 2 == 0:                        # the whole step at once ...
            ra.step(G, P, N, adam_fn, void_src=mine)
        else:                                    # ... or chunk by chunk, as the trainer drives it under the backward kernels
            ra.begin(N, "cpu", False)
            for c in range(ra.n_chunks):
                ra.reduce_chunk(c, G, mine)
            # ... with the parameter all-gathers left in flight (round 5: their tail runs under the next iteration's staging);
            # the next begin() / gather() / an explicit wait_gathers() collects them
            ra.finish(G, P, adam_fn, defer_gather_wait=(step % 4 == 1))
            if step % 4 == 1 and world_size > 1:
                assert ra._pending_ag
                if step == len(Ns) - 1:
                    ra.wait_gathers()
                    assert not ra._pending_ag
        assert float(ra.void_flag()[0]) == (2.0 if step == void_at else 0.0)
        done += 0 if step == void_at else 1
        own = ra.owned(N)
        assert sorted({(x[1], x[2]) for x in touched[before:]}) == sorted(own), (touched[before:], own)
        assert sum(b - a for r in range(world_size) for a, b in ra.owned(N, r)) == N        # the pieces tile the live rows
    ra.gather([M[k] for k in ROWS] + [V[k] for k in ROWS], Ns[-1])
    N = Ns[-1]
    torch.save({"P": {k: P[k][:N] for k in ROWS}, "M": {k: M[k][:N] for k in ROWS}, "V": {k: V[k][:N] for k in ROWS},
                "bytes": ra.bytes_per_link_and_step(N)}, os.path.join(out_dir, f"r{world_rank}.pt"))


@pytest.mark.parametrize("world,cap,Ns,n_chunks,void_at", [(2, 128, (37, 37, 51), 1, -1), (4, 1024, (777, 1001, 1001, 1001), 2, 2),
                                                            (8, 2048, (1234, 1234, 3), 3, -1), (2, 1024, (700, 700, 300), 4, 0)])
def test_row_sharded_adam_equals_adam_on_the_mean_gradient(tmp_path, world, cap, Ns, n_chunks, void_at):
    from splat_one_amd import distributed as sdist
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        sdist.cli(_row_worker, (str(tmp_path), cap, Ns, n_chunks, void_at), world_size=world, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for o in outs[1:]:
        for q in ("P", "M", "V"):
            for k in ROWS:
                assert torch.equal(o[q][k], outs[0][q][k]), (q, k)
    # single-process reference: torch.optim.Adam per tensor on the mean gradient; a change of N keeps the old rows'
    # state and gives new rows zero moments (what DefaultStrategy's duplicate / split do)
    hyper = _hyper(world)
    g0 = torch.Generator().manual_seed(7)
    full = {k: torch.randn(cap, rl, generator=g0) for k, rl in ROWS.items()}
    params, opts = {}, {}
    for step, N in enumerate(Ns):
        for k, rl in ROWS.items():
            if step == 0 or N != Ns[step - 1]:
                old = params.get(k)
                new = torch.nn.Parameter(full[k][:N].clone())
                lr, eps, betas = hyper[k]
                opt = torch.optim.Adam([new], lr=lr, eps=eps, betas=betas)
                if old is not None:
                    n0 = min(N, old.shape[0])
                    with torch.no_grad():
                        new[:n0] = old[:n0]
                    st_old = opts[k].state[old]
                    st = opt.state[new]
                    st["step"] = st_old["step"].clone()
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
                    st["exp_avg"][:n0], st["exp_avg_sq"][:n0] = st_old["exp_avg"][:n0], st_old["exp_avg_sq"][:n0]
                params[k], opts[k] = new, opt
            if step != void_at:                  # the void iteration never happened (its refinement bookkeeping did)
                params[k].grad = sum(_grads(N, r, step)[k].view(N, rl) for r in range(world)) / world
                opts[k].step()
    for k in ROWS:
        assert torch.allclose(outs[0]["P"][k], params[k].detach(), rtol=1e-5, atol=1e-7), k
        assert torch.allclose(outs[0]["M"][k], opts[k].state[params[k]]["exp_avg"], rtol=1e-5, atol=1e-6), k
    N = Ns[-1]
    q = 64 * world
    span = -(-N // (n_chunks * q)) * q * n_chunks
    assert abs(outs[0]["bytes"] - 2 * (world - 1) / world * span * 59 * 4) < 1


# ------------------------------------------------------------------------------------------------ fallback agreement
def _fallback_worker(local_rank, world_rank, world_size, args):
    """The grouped collectives of RowShardedAdam go through a PRIVATE torch API: if it is unusable on ANY rank (forced here
    on one rank only), every rank must leave it together -- one all_reduce(MIN) decides -- and the step must give the same
    numbers through the per-tensor collectives.  Phase timers on (bench.py's config.comm_ms)."""
    out_dir, cap, N, forced = args
    import warnings
    from splat_one_amd import distributed as sdist
    if forced is not None:
        os.environ["SPLAT_ONE_AMD_FORCE_COALESCE_FAIL"] = forced
    sdist.COMM_TIMING = True
    ra = sdist.RowShardedAdam()
    assert ra.mode is None and ra.timer is not None
    P = {k: torch.ones(cap, rl) for k, rl in ROWS.items()}
    G = {k: torch.full((cap, rl), float(world_rank + 1)) for k, rl in ROWS.items()}
    seen = []
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        for _ in range(2):
            for k in G:
                G[k].fill_(float(world_rank + 1))
            ra.step(G, P, N, lambda names, a, b, skip, gs: [P[k][a:b].sub_(G[k][a:b] * gs) for k in names] and seen.append((a, b)))
    cm = ra.comm_ms()
    assert cm["steps_timed"] == 2 and cm["collectives"] == ra.mode and cm["total_ms"] >= 0.0
    assert set(sdist._PhaseTimer.PHASES) <= set(cm)
    torch.save({"mode": ra.mode, "P": {k: P[k][:N] for k in ROWS}, "warned": sum("fall" in str(w.message) for w in wlist)},
               os.path.join(out_dir, f"r{world_rank}.pt"))


@pytest.mark.parametrize("forced,want", [(None, "coalesced"), ("1", "per_tensor"), ("all", "per_tensor")])
def test_collective_fallback_is_agreed_on_by_all_ranks(tmp_path, forced, want):
    from splat_one_amd import distributed as sdist
    world, cap, N = 2, 128, 37
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SPLAT_ONE_AMD_FORCE_COALESCE_FAIL")
    env_backup = {k: os.environ.pop(k, None) for k in keys}
    try:
        sdist.cli(_fallback_worker, (str(tmp_path), cap, N, forced), world_size=world, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert [o["mode"] for o in outs] == [want] * world          # the rank that did NOT fail follows the one that did
    assert all(o["warned"] == (1 if want == "per_tensor" else 0) for o in outs)
    mean_g = sum(r + 1 for r in range(world)) / world
    for o in outs:
        for k in ROWS:
            assert torch.equal(o["P"][k], torch.full_like(o["P"][k], 1.0 - 2 * mean_g)), k


def _trial_vote_worker(local_rank, world_rank, world_size, args):
    out_dir, forced = args
    import warnings
    from splat_one_amd import distributed as sdist
    if forced is not None:
        os.environ["SPLAT_ONE_AMD_FORCE_COALESCE_FAIL"] = forced
    ran = []
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        mode = sdist.agree_on_coalescing(lambda: None, "cpu", None, trial=lambda: ran.append(world_rank))
    torch.save({"mode": mode, "ran": ran, "warned": sum("fall" in str(w.message) for w in wlist)}, os.path.join(out_dir, f"t{world_rank}.pt"))


@pytest.mark.parametrize("forced,want,trial_ran", [(None, "coalesced", [True, True]), ("trial:1", "per_tensor", [True, False]),
                                                   ("0", "per_tensor", [False, False])])
def test_collective_vote_has_a_trial_phase_every_rank_follows(tmp_path, forced, want, trial_ran):
    """after the rank-local vote says "coalesced", every rank runs ONE trial grouped collective and votes again;
    a rank whose trial fails takes all ranks to "per_tensor"; a rank whose FIRST vote fails keeps every rank out of the trial."""
    from splat_one_amd import distributed as sdist
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SPLAT_ONE_AMD_FORCE_COALESCE_FAIL")
    env_backup = {k: os.environ.pop(k, None) for k in keys}
    try:
        sdist.cli(_trial_vote_worker, (str(tmp_path), forced), world_size=2, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
    outs = [torch.load(os.path.join(tmp_path, f"t{r}.pt")) for r in range(2)]
    assert [o["mode"] for o in outs] == [want] * 2
    assert [bool(o["ran"]) for o in outs] == trial_ran
    assert all(o["warned"] == (1 if want == "per_tensor" else 0) for o in outs)
