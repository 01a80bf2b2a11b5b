"""One seeded case of tests/test_gpu_fuzz.py::test_random_configuration_against_the_oracle in detail: which Gaussians carry
the gradient error, their radii / depths / opacities on both sides."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F
from oracle import c_oracle as CO
from oracle import torch_oracle as O
from splat_one_amd import rasterization
from splat_one_amd.scene import lookat_c2w
from tests.util import small_scene

seed = int(sys.argv[1])
dev = torch.device("cuda:0")
cfg = F._case(seed)
print(cfg)
W, H, C, N = cfg["W"], cfg["H"], cfg["C"], cfg["N"]
means, quats, scales, opac, sh = small_scene(N=N, seed=100 + seed, scale=cfg["scale"])
g = torch.Generator().manual_seed(1000 + seed)
c2w = torch.stack([lookat_c2w((6.0 * math.sin(0.9 * i + 0.3 * seed), 0.5 * i - 0.4, -6.0 * math.cos(0.9 * i + 0.3 * seed))) for i in range(C)])
viewmats = torch.linalg.inv(c2w).contiguous()
f = float(max(W, H)) * (0.04 if cfg["model"] == "ortho" else 0.9)
Ks = torch.tensor([[f, 0, W / 2.0], [0, f * 1.1, H / 2.0], [0, 0, 1]])[None].repeat(C, 1, 1)
X = 3
bg = torch.rand(C, 3, generator=g) if (cfg["bg"] and cfg["mode"].startswith("RGB")) else None
w_rgb, w_a = torch.rand(C, H, W, X, generator=g), torch.rand(C, H, W, 1, generator=g)
kw = dict(sh_degree=cfg["deg"], tile_size=cfg["tile"], render_mode=cfg["mode"], camera_model=cfg["model"],
          rasterize_mode="antialiased" if cfg["aa"] else "classic", near_plane=0.01, far_plane=1e8)
K = (cfg["deg"] + 1) ** 2

def run(fn, to, dt, **extra):
    p = [t.detach().clone().to(to).to(dt).requires_grad_(True) for t in (means, quats, scales, opac, sh[:, :K].contiguous())]
    rc, ra, meta = fn(*p, viewmats.to(to).to(dt), Ks.to(to).to(dt), W, H, backgrounds=None if bg is None else bg.to(to).to(dt), **kw, **extra)
    meta["means2d"].retain_grad() if hasattr(meta["means2d"], "retain_grad") and meta["means2d"].requires_grad else None
    ((rc * w_rgb.to(rc)).sum() + (ra * w_a.to(ra)).sum()).backward()
    return rc.detach().cpu().double(), ra.detach().cpu().double(), [t.grad.detach().cpu().double() for t in p], meta

rc_h, ra_h, g_h, m_h = run(rasterization, dev, torch.float32, packed=cfg["packed"])
keys = m_h["depths"].detach().cpu().clone()
rc_o, ra_o, g_o, m_o = run(O.rasterization, "cpu", torch.float64, raster_fn=CO.raster_fn(), sort_depths=keys)
print("fwd L1", (rc_h - rc_o).abs().mean().item(), "max", (rc_h - rc_o).abs().max().item())
d = (g_h[0] - g_o[0]).norm(dim=1)
print("means grad: err", d.norm().item(), "ref", g_o[0].norm().item())
top = torch.argsort(d, descending=True)[:6]
rh, ro = m_h["radii"].cpu(), m_o["radii"].cpu()
for i in top.tolist():
    print(i, "err", d[i].item(), "g_h", g_h[0][i].tolist(), "g_o", g_o[0][i].tolist(), "radii h", rh[:, i].tolist(), "o", ro[:, i].tolist(),
          "opac", opac[i].item(), "scales", scales[i].tolist())
    for c in range(C):
        print("   cam", c, "m2d h", m_h["means2d"][c, i].tolist(), "o", m_o["means2d"][c, i].tolist(), "depth", m_h["depths"][c, i].item(), "conic h", m_h["conics"][c, i].tolist(), "o", m_o["conics"][c, i].tolist())
print("radii differ:", int((rh != ro).sum()))
