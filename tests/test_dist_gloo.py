"""N>1 path on CPU: world_size 2 over gloo.  Covers the launcher (`cli`), the single flattened
gradient all-reduce of the replicated data parallelism, the densification-statistics all-reduce that
keeps replicated Gaussians identical on every rank, and the equal-split all-to-all of the
Gaussian-sharded scheme (the full sharded step runs in tests/test_gpu_trainer.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_fn(local_rank, world_rank, world_size, out_dir):
    from splat_one_amd import distributed as sdist
    assert dist.is_initialized() and dist.get_world_size() == world_size == 2
    torch.manual_seed(0)
    shapes = [(10, 3), (10, 4), (10,), (10, 15, 3)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    for i, p in enumerate(params):
        p.grad = torch.full(p.shape, float(world_rank + 1) * (i + 1))
    red = sdist.GradientReducer()
    red.reduce(params)
    for i, p in enumerate(params):
        assert torch.allclose(p.grad, torch.full(p.shape, 1.5 * (i + 1))), (world_rank, i)
    # one flat buffer backs every gradient (single all-reduce)
    base = params[0].grad.untyped_storage().data_ptr()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in params)
    state = {"grad2d": torch.full((10,), float(world_rank + 1)), "count": torch.ones(10) * (world_rank + 2),
             "scene_scale": 1.0}
    sdist.all_reduce_strategy_state(state)
    assert torch.allclose(state["grad2d"], torch.full((10,), 3.0)) and torch.allclose(state["count"], torch.full((10,), 5.0))
    # Gaussian-sharded exchange: block j of the send buffer lands on rank j, in source-rank order
    from splat_one_amd.sharded import all_to_all_rows
    cap = 3
    send = torch.stack([torch.full((cap, 16), 10.0 * world_rank + j) for j in range(world_size)]).reshape(-1, 16)
    recv = torch.empty_like(send)
    all_to_all_rows(recv, send)
    want = torch.stack([torch.full((cap, 16), 10.0 * j + world_rank) for j in range(world_size)]).reshape(-1, 16)
    assert torch.equal(recv, want), (world_rank, recv[:, 0])
    # rasterization(distributed=True): the differentiable variable-split all-to-all of the projected rows
    # (rank r owns 2 + r rows per camera block) and the camera gather
    from splat_one_amd.rendering import _AllToAllRows, _gather_cameras
    n_mine, n_all = 2 + world_rank, [2, 3]
    inp = (100.0 * world_rank + torch.arange(2 * n_mine * 4, dtype=torch.float32).reshape(2 * n_mine, 4)).requires_grad_(True)
    got = _AllToAllRows.apply(inp, [n_mine] * 2, n_all)
    assert got.shape == (5, 4)
    for j, blk in enumerate(got.split(n_all)):      # block j came from rank j: its rows [world_rank*n_j, (world_rank+1)*n_j)
        want = 100.0 * j + torch.arange(2 * n_all[j] * 4, dtype=torch.float32).reshape(2 * n_all[j], 4)
        assert torch.equal(blk, want[world_rank * n_all[j]:(world_rank + 1) * n_all[j]]), (world_rank, j)
    (got * (world_rank + 1.0)).sum().backward()     # the gradient of what rank j received returns to the sender
    want_grad = torch.cat([torch.full((n_mine, 4), j + 1.0) for j in range(2)])
    assert torch.equal(inp.grad, want_grad), (world_rank, inp.grad)
    vm = torch.eye(4)[None] * (world_rank + 1.0)
    ks = torch.eye(3)[None] * (world_rank + 5.0)
    N_world, vms, kss = _gather_cameras(7 + world_rank, vm, ks)
    assert N_world == [7, 8] and vms.shape == (2, 4, 4) and kss.shape == (2, 3, 3)
    assert torch.equal(vms[1], torch.eye(4) * 2.0) and torch.equal(kss[0], torch.eye(3) * 5.0)
    torch.save({"ok": True, "rank": world_rank}, os.path.join(out_dir, f"rank{world_rank}.pt"))


def test_view_sharded_dp_world2_gloo(tmp_path):
    from splat_one_amd import distributed as sdist
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        sdist.cli(_worker_fn, str(tmp_path), world_size=2, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
    for r in range(2):
        assert torch.load(os.path.join(tmp_path, f"rank{r}.pt"))["ok"]


def test_adam_rule_scales_with_world_size():
    """gsplat_trainer.py:266-278 with BS = batch_size * world_size."""
    from splat_one_amd.trainer import adam_hyperparameters
    lr, eps, betas = adam_hyperparameters(1.6e-4, 1, 8)
    assert abs(lr - 1.6e-4 * 8 ** 0.5) < 1e-18 and abs(eps - 1e-15 / 8 ** 0.5) < 1e-30
    assert abs(betas[0] - 0.2) < 1e-12 and abs(betas[1] - 0.992) < 1e-12
