"""`rasterization(distributed=True)` -- what the reference passes when world_size > 1 (gsplat_trainer.py:490): every
rank brings a shard of the Gaussians and its own camera; two ranks on the one GPU of the box (gloo) must reproduce the
single-process render of all Gaussians into both cameras, images and gradients -- and the float64 oracle's."""
import os
import socket

import pytest
import torch

from splat_one_amd.scene import make_scene
from tests.util import rel_err

pytestmark = pytest.mark.gpu

W, H, N = 128, 96, 2001            # odd: the shards differ in length
KW = dict(sh_degree=3, near_plane=0.01, far_plane=1e8, render_mode="RGB+ED", rasterize_mode="antialiased")


def _inputs():
    splats, c2w, Ks = make_scene(N, W, H, "ref", n_views=2)
    g = torch.Generator().manual_seed(5)
    splats = {k: v.detach().clone() for k, v in splats.items()}
    splats["scales"] = splats["scales"] + torch.randn(N, 3, generator=g) * 0.4
    bg = torch.tensor([[0.2, 0.4, 0.6], [0.5, 0.1, 0.3]])
    w = torch.rand(2, H, W, 4, generator=g)
    return splats, torch.linalg.inv(c2w), Ks, bg, w


def _render(dev, p, viewmats, Ks, bg, w, **kw):
    from splat_one_amd import rasterization
    kw.setdefault("packed", False)
    rc, ra, meta = rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]),
                                 torch.cat([p["sh0"], p["shN"]], 1), viewmats.to(dev), Ks.to(dev), W, H,
                                 backgrounds=bg.to(dev), **KW, **kw)
    meta["means2d"].retain_grad()
    ((rc * w.to(dev)).sum() + 0.1 * ra.sum()).backward()
    return rc.detach().cpu(), ra.detach().cpu(), meta


def _worker(local_rank, world_rank, world_size, args):
    out_dir, packed = args
    dev = torch.device("cuda:0")
    splats, viewmats, Ks, bg, w = _inputs()
    r = world_rank
    p = {k: v[r::world_size].clone().to(dev).requires_grad_(True) for k, v in splats.items()}
    rc, ra, meta = _render(dev, p, viewmats[r:r + 1], Ks[r:r + 1], bg[r:r + 1], w[r:r + 1], distributed=True, packed=packed)
    torch.cuda.synchronize()
    out = {"rc": rc, "ra": ra, "grads": {k: v.grad.detach().cpu() for k, v in p.items()},
           "v_means2d": meta["means2d"].grad.detach().cpu(), "radii": meta["radii"].cpu(),
           "n_cameras": meta["n_cameras"], "flat": meta["flatten_ids"].numel()}
    if packed:
        out.update(camera_ids=meta["camera_ids"].cpu(), gaussian_ids=meta["gaussian_ids"].cpu())
    torch.save(out, os.path.join(out_dir, f"d{r}.pt"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("packed", [False, True])
def test_rasterization_distributed_two_ranks_one_gpu(dev, tmp_path, packed):
    """packed=True (gsplat's default; `cfg.packed`, gsplat_trainer.py:133, :487-490): only the visible rows travel."""
    from splat_one_amd import distributed as sdist
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        sdist.cli(_worker, (str(tmp_path), packed), world_size=2, backend="gloo", port=_free_port())
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
    out = [torch.load(os.path.join(tmp_path, f"d{i}.pt")) for i in range(2)]
    # single process: all Gaussians in rank-major order, both cameras
    splats, viewmats, Ks, bg, w = _inputs()
    order = torch.cat([torch.arange(0, N, 2), torch.arange(1, N, 2)])
    p = {k: v[order].clone().to(dev).requires_grad_(True) for k, v in splats.items()}
    rc, ra, meta = _render(dev, p, viewmats, Ks, bg, w)
    n0 = (N + 1) // 2
    rows = [slice(0, n0), slice(n0, N)]
    for r in range(2):
        o = out[r]
        assert o["n_cameras"] == 1 and o["flat"] > 0
        assert (o["rc"][0] - rc[r]).abs().max().item() < 1e-5 and (o["ra"][0] - ra[r]).abs().max().item() < 1e-6
        # meta keeps the shard-local grid over the cameras of ALL ranks (packed: its visible rows, camera-major)
        if packed:
            dense_r, dense_g = meta["radii"][:, rows[r]].cpu(), meta["means2d"].grad[:, rows[r]].cpu()
            cid, gid = o["camera_ids"], o["gaussian_ids"]
            assert torch.equal(torch.nonzero(dense_r.reshape(-1) > 0).squeeze(1), cid * dense_r.shape[1] + gid)
            assert torch.equal(o["radii"], dense_r[cid, gid])
            assert rel_err(o["v_means2d"], dense_g[cid, gid]) < 1e-5
        else:
            assert torch.equal(o["radii"], meta["radii"][:, rows[r]].cpu())
            assert rel_err(o["v_means2d"], meta["means2d"].grad[:, rows[r]].cpu()) < 1e-5
        for k, g in o["grads"].items():
            assert rel_err(g, p[k].grad[rows[r]].cpu()) < 1e-5, (k, r)
    # ... and the ORACLE's render of all Gaussians into both cameras (float64 autograd projection / SH / binning + the C
    # rasteriser): images <= 1e-4, the gradients of every shard <= 1e-3 -- parity of the distributed call itself
    from oracle import c_oracle as CO
    from oracle import torch_oracle as O
    q = {k: v[order].clone().double().requires_grad_(True) for k, v in splats.items()}
    rc_o, ra_o, _ = O.rasterization(q["means"], q["quats"], torch.exp(q["scales"]), torch.sigmoid(q["opacities"]),
                                    torch.cat([q["sh0"], q["shN"]], 1), viewmats.double(), Ks.double(), W, H,
                                    backgrounds=bg.double(), raster_fn=CO.raster_fn(), **KW)
    ((rc_o * w.double()).sum() + 0.1 * ra_o.sum()).backward()
    for r in range(2):
        o = out[r]
        assert (o["rc"][0].double() - rc_o[r]).abs().mean().item() <= 1e-4 and (o["ra"][0].double() - ra_o[r]).abs().mean().item() <= 1e-4
        for k, g in o["grads"].items():
            ref = q[k].grad[rows[r]]
            floor = 1e-5 * q["scales"].grad.norm().item() if k == "quats" else 0.0
            assert (g.double() - ref).norm().item() <= 1e-3 * ref.norm().item() + floor, (k, r)
