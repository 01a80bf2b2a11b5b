"""One engine fuzz seed in detail: where does the gradient error of a tensor sit?  A discrete float32 decision -- one pixel of
one splat on the other side of alpha = 1/255 or of the T <= 1e-4 stop -- puts the whole error on one or two Gaussians (the
oracle takes the device's depth order and L1 signs over, not those decisions); an arithmetic defect spreads it.
    python tools/dbg_fuzz_one.py SEED [tensor]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F

args = [a for a in sys.argv[1:] if not a.startswith("--")]
seed = int(args[0])
name = args[1] if len(args) > 1 else "means"
if "--operator" in sys.argv:        # the rasterization() fuzz case of this seed
    cfg, (rc_h, rc_o), (ra_h, ra_o), gh, go = F._operator_against_the_oracle(torch.device("cuda:0"), seed)
    names = ["means", "quats", "scales", "opacities", "sh"]
    g_eng, g_ref = dict(zip(names, gh)), dict(zip(names, go))
    fwd = (rc_h - rc_o).abs()
else:
    cfg, g_eng, g_ref, K, fwd, (loss_eng, l1_o, ss_o) = F._engine_against_the_oracle(torch.device("cuda:0"), seed)
print("case", cfg)
print("forward: mean |diff| %.3e  max %.3e  pixels > 1e-3: %d  > 1e-4: %d" % (fwd.mean().item(), fwd.max().item(), int((fwd > 1e-3).sum()), int((fwd > 1e-4).sum())))
for k in g_eng:
    if g_ref[k] is not None:
        print("%-10s |engine - oracle| %.3e  |oracle| %.3e  relative %.2e" % (k, (g_eng[k] - g_ref[k]).norm().item(), g_ref[k].norm().item(),
                                                                        (g_eng[k] - g_ref[k]).norm().item() / max(g_ref[k].norm().item(), 1e-30)))
d = (g_eng[name] - g_ref[name]).reshape(g_eng[name].shape[0], -1)
row = d.norm(dim=1)
tot2 = (row ** 2).sum().item()
top = torch.topk(row, 6)
print(f"{name}: error by Gaussian -- share of the squared error in the 1 / 2 / 6 worst rows:",
      ["%.3f" % ((top.values[:n] ** 2).sum().item() / tot2) for n in (1, 2, 6)], "rows", top.indices.tolist())
for i in top.indices.tolist()[:3]:
    print("  row", i, "engine", g_eng[name][i].flatten()[:4].tolist(), "oracle", g_ref[name][i].flatten()[:4].tolist())
