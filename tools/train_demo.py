#!/usr/bin/env python
"""train_demo.py -- does the path TRAIN a scene?  (the reference's loop ends in `eval()` with PSNR,
/root/reference/utils/gsplat_utils/gsplat_trainer.py:551-777, 780-842.)

A frozen ground-truth model (anisotropic Gaussians, view-dependent colour) is rendered at 16 ring views: the 8 even ones are
the training set, the 8 odd ones are held out.  The student is the reference's own random initialisation
(`create_splats_with_optimizers`, :204-281) trained with `Runner.train_step` -- DefaultStrategy or MCMCStrategy refining
every 100 steps -- through the fused engine or through the operator-level autograd path, float32 or float16 attribute
rows; then `Runner.eval` (PSNR / SSIM, :780-842) on the held-out views.  `oracle_steps > 0` also trains the same student for
that many steps with gradients from the float64 ORACLE (oracle/torch_oracle.py + oracle/c/raster_oracle.c, torch Adam on
the CPU) and evaluates both after those steps with the oracle's own renderer.

    python tools/train_demo.py                      # the four product runs + the oracle leg, one JSON line each
    python tools/train_demo.py --sweep              # teacher / student sizes (used once to choose the test's scene)
"""
import argparse
import json
import math
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def teacher_scene(n, scale_mult, opacity, seed, dev):
    """Ground truth: the reference's random cloud (seeded), log-scales spread by N(0, 0.4^2) per axis so that the splats are
    anisotropic, opacity `opacity`, random base colour, small higher SH bands (view-dependent colour)."""
    from splat_one_amd.scene import random_splats, set_random_seed
    set_random_seed(seed)
    s = random_splats(n, init_opacity=opacity, init_scale=scale_mult, shN_std=0.05, shN_seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    s["scales"] = s["scales"] + torch.randn(n, 3, generator=g) * 0.4
    return {k: v.to(dev) for k, v in s.items()}


def render_views(splats, cams, K, res, dev, camera_model="pinhole"):
    from splat_one_amd import rasterization
    out = []
    with torch.no_grad():
        for c2w in cams:
            rc, _, _ = rasterization(splats["means"], splats["quats"], torch.exp(splats["scales"]), torch.sigmoid(splats["opacities"]),
                                     torch.cat([splats["sh0"], splats["shN"]], 1), torch.linalg.inv(c2w)[None].to(dev), K[None].to(dev),
                                     (2 * res if camera_model == "spherical" else res), res, sh_degree=3, near_plane=0.01, far_plane=1e8,
                                     packed=False, camera_model=camera_model)
            out.append(rc[0, ..., :3].clamp(0.0, 1.0).contiguous())
    return out


def block_means(x, block=100):
    n = len(x) // block
    return [float(sum(x[i * block:(i + 1) * block]) / block) for i in range(n)]


def run(path="engine", strategy="default", attr_dtype="f32", steps=600, res=256, teacher_n=20_000, teacher_scale=1.0,
        teacher_opacity=0.6, student_n=20_000, refine_every=100, refine_start=100, oracle_steps=0, seed=7, device="cuda:0",
        result_dir=None, return_runner=False, train_views=8, init="random", reset_every=100_000, sh_interval=100, refine_stop=None,
        max_steps=None, time_blocks=0, scene_scale=1.0 / 1.1, world_rank=0, world_size=1, dp_mode="allreduce", camera_model="pinhole"):
    """init="random": the reference's random initialisation (init_type="random", :224-257).  init="sfm": init_type="sfm" (:216-223) --
    what the reference does on real data: the student starts from a sparse point cloud with colours, here `student_n` of the
    ground-truth centres displaced by N(0, 0.05^2) with their base colours."""
    from splat_one_amd.scene import pinhole_K, ring_cameras
    from splat_one_amd.strategy import DefaultStrategy, MCMCStrategy
    from splat_one_amd.trainer import Config, Runner
    dev = torch.device(device)
    n_ring = 2 * train_views
    ring = ring_cameras(n_ring)
    K = pinhole_K(res, res)
    if camera_model == "spherical":
        # the reference's DEFAULT camera model (gsplat_trainer.py:460-461): 360-degree equirectangular images, 2 res x res, taken from INSIDE
        # the cloud -- positions on a circle of radius 1 around its centre, a little up and down, each camera turned about the vertical axis
        K = pinhole_K(2 * res, res)
        ring = []
        for k in range(n_ring):
            th = 2 * math.pi * k / n_ring
            c2w = torch.eye(4)
            c2w[0, 0], c2w[0, 2], c2w[2, 0], c2w[2, 2] = math.cos(2 * th), math.sin(2 * th), -math.sin(2 * th), math.cos(2 * th)
            c2w[:3, 3] = torch.tensor([math.sin(th), 0.3 * math.cos(3 * th), math.cos(th)])
            ring.append(c2w)
        ring = torch.stack(ring)
    teacher = teacher_scene(teacher_n, teacher_scale, teacher_opacity, seed, dev)
    images = render_views(teacher, ring, K, res, dev, camera_model)
    train_ids, held_ids = list(range(0, n_ring, 2)), list(range(1, n_ring, 2))
    as_view = lambda i: {"camtoworld": ring[i], "K": K, "image": images[i] * 255.0}
    held = [as_view(i) for i in held_ids]
    if strategy == "default":
        strat = DefaultStrategy(refine_start_iter=refine_start, refine_every=refine_every,
                                refine_stop_iter=(steps - 50 if refine_stop is None else refine_stop), reset_every=reset_every, verbose=False)
        kw = dict(init_opa=0.1, init_scale=1.0)                                   # the `default` preset, :117-119
    else:
        strat = MCMCStrategy(refine_start_iter=refine_start, refine_every=refine_every, refine_stop_iter=(steps - 50 if refine_stop is None else refine_stop),
                             cap_max=int(1.5 * student_n), verbose=False)
        kw = dict(init_opa=0.5, init_scale=0.1, opacity_reg=0.01, scale_reg=0.01)  # the `mcmc` preset, :977-983
    tmp = result_dir or tempfile.mkdtemp(prefix="train_demo_")
    cfg = Config(init_num_pts=student_n, strategy=strat, sh_degree_interval=sh_interval, max_steps=(max_steps or steps), fused=(path == "engine"),
                 attr_dtype=attr_dtype, result_dir=tmp, init_type=init, camera_model=camera_model, **(dict(dp_mode=dp_mode) if world_size > 1 else {}), **kw)
    pts = rgbs = None
    if init == "sfm":
        from splat_one_amd.scene import SH_C0
        gi = torch.Generator().manual_seed(seed + 3)
        pick = torch.randperm(teacher_n, generator=gi)[:min(student_n, teacher_n)]
        pts = teacher["means"].cpu()[pick] + torch.randn(len(pick), 3, generator=gi) * 0.05
        rgbs = (teacher["sh0"].cpu()[pick, 0] * SH_C0 + 0.5).clamp(0.0, 1.0)
    # world_size > 1 (called from inside an initialised process group): replicated data parallelism, rank r trains on view
    # (it * world_size + r) of the cycle -- one view per rank and iteration, the north_star's scheme
    r = Runner(0, world_rank, world_size, cfg, scene_scale=scene_scale, points=pts, rgbs=rgbs)
    init_params = {k: v.detach().clone() for k, v in r.splats.items()}
    cams = [ring[i][None].contiguous().to(dev) for i in train_ids]
    Ks = K[None].to(dev)
    tg = [images[i][None].contiguous() for i in train_ids]
    psnr0 = r.eval(0, dataset=held, save_images=False).get("psnr")     # (rank 0 scores, as in the reference)
    losses, mid = [], None
    import time
    blocks, t_block = [], None
    if time_blocks:
        torch.cuda.synchronize()
        t_block = time.time()
    for it in range(steps):
        v = (it * world_size + world_rank) % len(cams)
        if r.sharded:      # the reference's scheme (Gaussian shards): a step takes the cameras of ALL ranks and the own image
            vs = [(it * world_size + q) % len(cams) for q in range(world_size)]
            losses.append(r.train_step(torch.cat([cams[q] for q in vs]), Ks.repeat(world_size, 1, 1), tg[v]).detach().clone())
        else:
            losses.append(r.train_step(cams[v], Ks, tg[v]).detach().clone())
        if time_blocks and (it + 1) % time_blocks == 0:      # wall clock per block of iterations (one synchronisation per block)
            torch.cuda.synchronize()
            now = time.time()
            blocks.append({"steps": it + 1, "seconds": round(now - t_block, 3), "it_s": round(time_blocks / (now - t_block), 1),
                           "gaussians": int(len(r.splats["means"]))})
            t_block = now
        if oracle_steps and it + 1 == oracle_steps:
            mid = {k: v_.detach().clone().cpu().double() for k, v_ in r.splats.items()}
    torch.cuda.synchronize()
    losses = torch.stack(losses).cpu().tolist()
    st = r.eval(steps, dataset=held, save_images=False)
    st_train = r.eval(steps, dataset=[as_view(i) for i in train_ids], save_images=False, stage="train")
    bm = block_means(losses, refine_every)       # (one refinement period per block: a refinement's new children bump the loss)
    out = {"path": path, "strategy": strategy, "attr_dtype": attr_dtype, "steps": steps, "res": res, "teacher_n": teacher_n,
           "teacher_scale": teacher_scale, "student_n": student_n, "train_views": train_views, "init": init, "n_final": int(len(r.splats["means"])),
           "psnr_heldout_before": psnr0, "psnr_heldout": st.get("psnr"), "ssim_heldout": st.get("ssim"), "psnr_train": st_train.get("psnr"),
           "loss_first": losses[0], "loss_last": losses[-1], "loss_block_means": bm,
           "loss_monotone": all(b < a for a, b in zip(bm, bm[1:])),
           "void_steps": getattr(getattr(r, "_engine", None), "void_steps", 0),
           "fused_engine_ran": getattr(r, "_engine", None) is not None}
    if blocks:
        out["blocks"] = blocks
        out["wall_seconds"] = round(sum(b["seconds"] for b in blocks), 2)
    if oracle_steps:
        out["oracle"] = oracle_leg(init_params, mid, [ring[i] for i in train_ids], [images[i].cpu() for i in train_ids],
                                   [ring[i] for i in held_ids], [images[i].cpu() for i in held_ids], K, res, oracle_steps, steps, cfg)
    return (out, r) if return_runner else out


def oracle_leg(init, product_after, train_cams, train_imgs, held_cams, held_imgs, K, res, n_steps, max_steps, cfg):
    """The same student, `n_steps` iterations with the float64 oracle's gradients (autograd through oracle/torch_oracle.py,
    rasteriser oracle/c/raster_oracle.c) and torch.optim.Adam with the reference's hyper-parameters (gsplat_trainer.py:246-278,
    ExponentialLR on the means :512-516); then BOTH parameter sets -- the oracle's and the product's after the same steps --
    are rendered at the held-out views by the oracle and scored."""
    from oracle import c_oracle as CO
    from oracle import torch_oracle as O
    from oracle.ssim_oracle import photometric_loss as oracle_loss
    from splat_one_amd.trainer import PARAM_LRS, adam_hyperparameters
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    raster = CO.raster_fn()
    p = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in init.items()}
    opts = {}
    for name, lr in PARAM_LRS:
        lr, eps, betas = adam_hyperparameters(lr, 1, 1)
        opts[name] = torch.optim.Adam([p[name]], lr=lr, eps=eps, betas=betas)
    gamma = 0.01 ** (1.0 / max_steps)

    def render(q, c2w, deg):
        rc, _, _ = O.rasterization(q["means"], q["quats"], torch.exp(q["scales"]), torch.sigmoid(q["opacities"]),
                                   torch.cat([q["sh0"], q["shN"]], 1), torch.linalg.inv(c2w.double())[None], K.double()[None], res, res,
                                   sh_degree=deg, near_plane=cfg.near_plane, far_plane=cfg.far_plane, raster_fn=raster)
        return rc

    for it in range(n_steps):
        v = it % len(train_cams)
        deg = min(it // cfg.sh_degree_interval, cfg.sh_degree)
        rc = render(p, train_cams[v], deg)
        loss, _, _ = oracle_loss(rc, train_imgs[v][None].double(), cfg.ssim_lambda)
        loss.backward()
        for o in opts.values():
            o.step()
            o.zero_grad(set_to_none=True)
        opts["means"].param_groups[0]["lr"] *= gamma

    def psnr(q):
        vals = []
        with torch.no_grad():
            for c2w, img in zip(held_cams, held_imgs):
                rc = render(q, c2w, cfg.sh_degree).clamp(0.0, 1.0)[0]
                vals.append(10.0 * math.log10(1.0 / float(((rc - img.double()) ** 2).mean())))
        return sum(vals) / len(vals)

    q_o = {k: v.detach() for k, v in p.items()}
    rel = {k: float((product_after[k] - q_o[k]).norm() / (q_o[k] - init[k].cpu().double()).norm().clamp_min(1e-30)) for k in q_o}
    return {"steps": n_steps, "psnr_heldout_oracle_gradients": psnr(q_o), "psnr_heldout_product": psnr(product_after),
            "parameter_update_rel_diff": rel}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--sweep2", action="store_true")
    ap.add_argument("--long", action="store_true",
                    help="the reference's OWN schedule to its first evaluation: 7000 iterations, DefaultStrategy defaults (refine from 500 every 100, "
                         "opacity reset every 3000), SH degree +1 every 1000, max_steps 30000; wall clock per 1000 iterations")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--path", default="engine", choices=["engine", "operator"])
    ap.add_argument("--train-views", type=int, default=8)
    ap.add_argument("--init", default="random", choices=["random", "sfm"])
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--oracle-steps", type=int, default=20)
    ap.add_argument("--teacher-n", type=int, default=20_000)
    ap.add_argument("--teacher-scale", type=float, default=1.0)
    ap.add_argument("--student-n", type=int, default=20_000)
    args = ap.parse_args()
    if args.long:
        for strategy in ("default", "mcmc"):
            o = run(args.path, strategy, steps=7000, res=args.res, teacher_n=args.teacher_n, teacher_scale=args.teacher_scale, student_n=args.student_n,
                    train_views=args.train_views, init=args.init, refine_start=500, refine_every=100, reset_every=3000, sh_interval=1000,
                    refine_stop=15000, max_steps=30000, time_blocks=1000,
                    # the parser's scene scale: the largest distance of a camera from the cameras' centroid (opensfm.py:300-304) = the ring's
                    # radius, 9.  (With the 1 / 1.1 of the short runs every splat of this cloud is "too big" -- larger than prune_scale3d x
                    # scene_scale = 0.1 -- once step > reset_every, and DefaultStrategy prunes the model away: seen in both paths.)
                    scene_scale=9.0)
            o["loss_block_means"] = [round(x, 5) for x in o["loss_block_means"][::5]]
            print(json.dumps(o), flush=True)
        return
    if args.sweep:
        for tn, ts, sn in [(20_000, 1.0, 20_000), (20_000, 1.5, 20_000), (20_000, 1.0, 50_000), (5_000, 1.5, 20_000), (2_000, 2.0, 20_000),
                           (20_000, 0.5, 50_000)]:
            for strategy in ("default", "mcmc"):
                o = run("engine", strategy, steps=args.steps, teacher_n=tn, teacher_scale=ts, student_n=sn)
                o.pop("loss_block_means")
                print(json.dumps(o), flush=True)
        return
    if args.sweep2:
        for tv, init, steps in [(8, "sfm", 600), (16, "random", 600), (32, "random", 600), (32, "random", 1500), (16, "sfm", 600), (32, "sfm", 1000)]:
            for strategy in ("default", "mcmc"):
                o = run("engine", strategy, steps=steps, train_views=tv, init=init, teacher_n=args.teacher_n, teacher_scale=args.teacher_scale,
                        student_n=args.student_n)
                o.pop("loss_block_means")
                print(json.dumps(o), flush=True)
        return
    common = dict(steps=args.steps, teacher_n=args.teacher_n, teacher_scale=args.teacher_scale, student_n=args.student_n,
                  train_views=args.train_views, init=args.init)
    for path, strategy, attr in [("engine", "default", "f32"), ("operator", "default", "f32"), ("engine", "mcmc", "f32"),
                                 ("engine", "default", "f16")]:
        o = run(path, strategy, attr, oracle_steps=(args.oracle_steps if (path, strategy, attr) == ("engine", "default", "f32") else 0), **common)
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
