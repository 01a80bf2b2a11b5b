"""Where does the untrimmed gradient error of the c2 parity test come from?  

For the c2 workload (100k Gaussians, 1080p, both regimes) compares, row by row,
  product (HIP float32, fused engine)   vs  float64 oracle
  float32 oracle (same restatement, REAL=float + torch float32)  vs  float64 oracle
and prints for the worst rows what they are (depth, radius, opacity, tiles, share of the tensor norm) and
whether the float32 restatement is off on the same rows (=> float32 arithmetic, not the product).
Usage (GPU box): python tools/dbg_gradflips.py [mcmc|ref] [N]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import c_oracle as CO          # noqa: E402
from oracle import ssim_oracle as SSO      # noqa: E402
from oracle import torch_oracle as O       # noqa: E402
from splat_one_amd.scene import front_camera, pinhole_K   # noqa: E402


def oracle_step(splats, c2w, Ks, W, H, pixels, dtype):
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in splats.items()}
    colors = torch.cat([p["sh0"], p["shN"]], 1)
    rc, ra, meta = O.rasterization(p["means"], p["quats"], torch.exp(p["scales"]), torch.sigmoid(p["opacities"]), colors,
                                   torch.linalg.inv(c2w.cpu()), Ks.cpu(), W, H, sh_degree=3, near_plane=0.01,
                                   far_plane=1e8, raster_fn=CO.raster_fn(), dtype=dtype)
    loss, l1, ss = SSO.photometric_loss(rc.double(), pixels.cpu(), 0.2)
    loss.backward()
    return rc.detach().double(), {k: v.grad.double() for k, v in p.items()}, meta


def main():
    regime = sys.argv[1] if len(sys.argv) > 1 else "mcmc"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    from splat_one_amd.engine import FusedEngine
    from splat_one_amd.trainer import Config, Runner
    dev = torch.device("cuda:0")
    W, H = 1920, 1080
    cfg = Config(init_num_pts=N, init_scale=(1.0 if regime == "ref" else 0.1), init_opa=(0.1 if regime == "ref" else 0.5),
                 shN_init_std=0.1)
    r = Runner(0, 0, 1, cfg, scene_scale=1.0 / 1.1)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        r.splats["scales"].add_((torch.randn(N, 3, generator=g) * 0.3).to(dev))
    c2w = front_camera()[None].to(dev)
    Ks = pinhole_K(W, H)[None].to(dev)
    pixels = torch.rand(1, H, W, 3, generator=g).to(dev)
    rc64, g64, meta64 = oracle_step(r.splats, c2w, Ks, W, H, pixels, torch.float64)
    rc32, g32, meta32 = oracle_step(r.splats, c2w, Ks, W, H, pixels, torch.float32)
    eng = FusedEngine(r.splats, r.optimizers, W, H, 1, sh_degree=3, use_graph=False)
    eng.set_views(c2w, Ks, pixels)
    eng.fwd_bwd()
    gh = {k: v.grad.detach().cpu().double() for k, v in r.splats.items()}
    out = {"regime": regime, "N": N,
           "fwd_L1_product": (eng.ws["render_colors"].cpu().double() - rc64).abs().mean().item(),
           "fwd_L1_f32oracle": (rc32 - rc64).abs().mean().item(), "tensors": {}}
    radii = meta64["radii"][0]
    depths = meta64["depths"][0]
    tpg = meta64["tiles_per_gauss"][0]
    opa = torch.sigmoid(r.splats["opacities"].detach().cpu())
    # discrete differences between the float32 and float64 restatements
    out["radii_differ_f32_vs_f64"] = int((meta32["radii"] != meta64["radii"]).sum())
    out["n_isects_f64"] = int(meta64["flatten_ids"].numel())
    out["n_isects_f32"] = int(meta32["flatten_ids"].numel())
    out["radii_differ_product_vs_f64"] = int((eng.ws["radii"].cpu() != meta64["radii"]).sum())
    for k in g64:
        ref = g64[k].reshape(N, -1)
        dh = gh[k].reshape(N, -1) - ref
        d32 = g32[k].reshape(N, -1) - ref
        tn = ref.norm().item()
        rowh, row32 = dh.norm(dim=1), d32.norm(dim=1)
        top = torch.topk(rowh, 12).indices
        rows = []
        for i in top.tolist():
            rows.append({"n": i, "err_product": rowh[i].item() / tn, "err_f32oracle": row32[i].item() / tn,
                         "row_norm_share": ref[i].norm().item() / tn, "row_rel_err": (rowh[i] / ref[i].norm().clamp_min(1e-300)).item(),
                         "depth": depths[i].item(), "radius": int(radii[i]), "opacity": opa[i].item(), "tiles": int(tpg[i])})
        srt = torch.sort(rowh ** 2, descending=True).values
        cum = torch.cumsum(srt, 0)
        out["tensors"][k] = {
            "rel_product": dh.norm().item() / tn, "rel_f32oracle": d32.norm().item() / tn,
            "rel_product_vs_f32oracle": (gh[k] - g32[k]).norm().item() / tn,
            "rows_for_half_of_err2": int((cum < 0.5 * cum[-1]).sum()) + 1,
            "rel_product_without_top100": (cum[-1] - cum[99]).clamp_min(0).sqrt().item() / tn,
            "worst_rows": rows}
        print(k, json.dumps({kk: vv for kk, vv in out["tensors"][k].items() if kk != "worst_rows"}))
        for rw in rows[:6]:
            print("   ", json.dumps(rw))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"gradflips_{regime}_{N}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "tensors"}))


if __name__ == "__main__":
    main()
