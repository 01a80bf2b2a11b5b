import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import c_oracle as CO, ssim_oracle as SSO, torch_oracle as O
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.engine import FusedEngine
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
W, H, N, C = 1920, 1080, 500_000, 2
r = Runner(0, 0, 1, Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1), scene_scale=1.0 / 1.1)
c2w = ring_cameras(8)[[0, 3]].to(dev)
Ks = pinhole_K(W, H)[None].repeat(C, 1, 1).to(dev)
pixels = torch.rand(C, H, W, 3, generator=torch.Generator().manual_seed(2)).to(dev)
p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in r.splats.items()}
colors = torch.cat([p["sh0"], p["shN"]], 1)
rc, ra, meta = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), colors,
                               torch.linalg.inv(c2w.cpu()), Ks.cpu(), W, H, sh_degree=3, near_plane=0.01, far_plane=1e8, raster_fn=CO.raster_fn())
loss, _, _ = SSO.photometric_loss(rc, pixels.cpu(), 0.2)
loss.backward()
eng = FusedEngine(r.splats, r.optimizers, W, H, C, sh_degree=3, use_graph=False)
eng.set_views(c2w, Ks, pixels); eng.fwd_bwd()
gh = r.splats["means"].grad.cpu().double(); go = p["means"].grad.double()
err = (gh - go).norm(dim=1)
tot = err.pow(2).sum().sqrt().item()
top = torch.topk(err, 10)
print("total err", tot, "ref norm", go.norm().item())
print("top10 err", top.values.tolist())
print("err^2 share of top 10:", (top.values.pow(2).sum() / err.pow(2).sum()).item(), "top 100:", (torch.topk(err,100).values.pow(2).sum()/err.pow(2).sum()).item())
rad_h = eng.ws["radii"].cpu(); rad_o = meta["radii"]
mism = (rad_h != rad_o)
print("radii mismatches:", int(mism.sum()), "of", rad_o.numel(), "; among top-10 rows:", [bool(mism[:, i].any()) for i in top.indices.tolist()])
print("top rows ref grad norm:", go[top.indices].norm(dim=1).tolist())
print("top rows radii h/o:", rad_h[:, top.indices].tolist(), rad_o[:, top.indices].tolist())
mask = ~mism.any(0)
print("rel err excluding mismatched-radii Gaussians:", ((gh-go)[mask].norm()/go[mask].norm()).item())
