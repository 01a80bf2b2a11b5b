"""The RCCL (backend "nccl") branches of the multi-GPU code with a process group of ONE rank on the one GPU of the box:
every collective degenerates to a copy, but the calls, dtypes, split arguments and stream ordering are the ones the
8-GPU runs issue (the two-rank tests use gloo, whose branches stage through the host)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splat_one_amd.distributed import _single_node_sockets                           # noqa: E402
_single_node_sockets()          # bootstrap sockets on `lo`: no hostname lookups (minutes where the name does not resolve)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
print("backend", dist.get_backend(), flush=True)

from splat_one_amd import distributed as sdist, rasterization                       # noqa: E402
from splat_one_amd.rendering import _AllToAllRows, _gather_cameras                   # noqa: E402
from splat_one_amd.scene import make_scene, pinhole_K, ring_cameras                  # noqa: E402
from splat_one_amd.sharded import ShardedEngine, all_to_all_rows                     # noqa: E402
from splat_one_amd.trainer import Config, Runner                                     # noqa: E402

# collectives of the helpers
x = torch.arange(48, dtype=torch.float32, device=dev).reshape(3, 16)
y = torch.empty_like(x)
all_to_all_rows(y, x)
assert torch.equal(x, y)
inp = torch.randn(5, 4, device=dev, requires_grad=True)
got = _AllToAllRows.apply(inp, [5], [5])
(got * 2).sum().backward()
assert torch.equal(got, inp.detach()) and torch.equal(inp.grad, torch.full_like(inp, 2.0))
N_world, vm, ks = _gather_cameras(7, torch.eye(4, device=dev)[None], torch.eye(3, device=dev)[None])
assert N_world == [7] and vm.shape == (1, 4, 4) and ks.shape == (1, 3, 3)
ps = [torch.nn.Parameter(torch.zeros(4, 3, device=dev)), torch.nn.Parameter(torch.zeros(4, device=dev))]
for p in ps:
    p.grad = torch.ones_like(p)
red = sdist.GradientReducer()
red._flat = None
# world 1: reduce() returns early; call the collective itself
flat = torch.ones(16, device=dev)
dist.all_reduce(flat)
sdist.all_reduce_strategy_state({"grad2d": torch.ones(4, device=dev), "count": torch.ones(4, device=dev)})
# reduce-scatter / sharded Adam / all-gather of the replicated scheme: the RCCL branches (in-place
# reduce_scatter_tensor, all_gather_into_tensor, async work handles) -- with one rank step() short-cuts, so the
# collective helpers are called directly
sa = sdist.ShardedFlatAdam(1000, n_chunks=3)
assert sa.backend == "nccl" and sa.world == 1 and sa.padded_total % (3 * 64) == 0
buf = torch.arange(sa.padded_total, dtype=torch.float32, device=dev)
ref = buf.clone()
for c in range(sa.n_chunks):
    for wk in sa._reduce_scatter(buf, c):
        wk.wait()
    sa._all_gather(buf, c).wait()
torch.cuda.synchronize()
assert torch.equal(buf, ref)
sa.gather_moments(buf, buf.clone())
# row pieces of a device-resident model (RowShardedAdam): grouped reduce_scatter_tensor / all_gather_into_tensor through
# torch's coalescing manager, in place, async handles -- with one rank every collective is a copy onto itself
ra = sdist.RowShardedAdam(n_chunks=2)
assert ra.backend == "nccl" and ra.world == 1 and ra.rows(100, 0) == (0, 64) and ra.rows(100, 1) == (64, 128) and ra.span(100) == 128
ra.mode = "coalesced"                # (one rank: step() short-cuts -- drive the chunk collectives and the flag reduce-scatter directly)
rows = [torch.full((128, 3), 2.0, device=dev), torch.full((128,), 3.0, device=dev), torch.full((128, 15, 3), 4.0, device=dev)]
for mode in ("coalesced", "per_tensor"):
    ra.mode = mode
    for c in range(ra.n_chunks):
        for wk in ra._reduce_scatter(rows, 100, c):
            wk.wait()
        for wk in ra._all_gather(rows, 100, c):
            wk.wait()
ra.flags = torch.full((ra.world * ra.FLAG_STRIDE,), 5.0, device=dev)
for mode in ("coalesced", "per_tensor"):          # the void flags inside the grouped launch / as one more collective
    ra.mode = mode
    for wk in ra._reduce_scatter(rows, 100, 0, with_flags=True):
        wk.wait()
for wk in ra._reduce_scatter_flags():
    wk.wait()
torch.cuda.synchronize()
assert all(torch.equal(t, torch.full_like(t, v)) for t, v in zip(rows, (2.0, 3.0, 4.0))) and float(ra.void_flag()[0]) == 5.0
sdist.probe_coalescing_locally()     # the rank-local half of the fallback vote: the private API exists and can be entered empty
# ... and the second half: one tiny GROUPED reduce-scatter (one of them in place, as the void flags are) + all-gather
# through this torch / RCCL build, values checked; then the two-phase vote itself
sdist.trial_grouped_collectives(dev)
assert sdist.agree_on_coalescing(sdist.probe_coalescing_locally, dev, None, lambda: sdist.trial_grouped_collectives(dev)) == "coalesced"
for wk in sa._reduce_scatter_flags(torch.tensor([1.0], device=dev), dev):
    wk.wait()
torch.cuda.synchronize()
assert float(sa.void_flag()[0]) == 1.0
mx = torch.tensor([3], dtype=torch.int32, device=dev)
sdist.all_reduce_max_(mx)
assert int(mx) == 3
print("helpers ok", flush=True)

# rasterization(distributed=True) == the plain call when the group has one rank
W, H, N = 128, 96, 1500
splats, c2w, Ks = make_scene(N, W, H, "ref", n_views=2)
args = [splats["means"], splats["quats"], splats["scales"].exp(), splats["opacities"].sigmoid(),
        torch.cat([splats["sh0"], splats["shN"]], 1)]
args = [a.to(dev) for a in args] + [torch.linalg.inv(c2w).to(dev), Ks.to(dev), W, H]
rc0, ra0, _ = rasterization(*args, sh_degree=3, packed=False, fused=False)
rc1, ra1, m1 = rasterization(*args, sh_degree=3, packed=False, distributed=True)
# (alphas bit for bit; colours to rounding: the plain dense call evaluates its colour stage in one fused launch)
assert (rc0 - rc1).abs().max().item() <= 2e-6 and torch.equal(ra0, ra1) and m1["n_cameras"] == 2
print("distributed operator ok", flush=True)

# the sharded engine over RCCL (its all-to-alls, capacity probe, flag kernels, workspace collectives)
cfg = Config(init_num_pts=20000, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
Wb, Hb = 640, 360
eng = ShardedEngine(r.splats, r.optimizers, Wb, Hb, 0, 1, sh_degree=3)
cam = ring_cameras(8)[:1].to(dev)
K1 = pinhole_K(Wb, Hb)[None].to(dev)
px = torch.rand(1, Hb, Wb, 3, device=dev)
before = r.splats["means"].detach().clone()
t0 = time.time()
for _ in range(20):
    eng.step(cam, K1, px)
torch.cuda.synchronize()
st = eng.stats()
assert st["overflow"] == 0 and st["n_isects"] > 0 and not torch.equal(before, r.splats["means"].detach())
assert torch.isfinite(eng.loss()).all()
print(f"sharded engine over RCCL ok: {st}, {(time.time() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
dist.barrier()
dist.destroy_process_group()
print("RCCL-1 OK")
