#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native 3DGS training path.

    python bench.py --gpus N --steps K --warmup W

N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when
no RANK is set -- bench.py starts its own N ranks (one child process per GPU, like `cli(main, cfg)` at
/root/reference/utils/gsplat_utils/gsplat_trainer.py:998) from a parent that never touches the GPU, forwards rank 0's JSON
line and exits non-zero if any rank failed.

Workload (BASELINE.json configs[1], "c2"): 100k random Gaussians (reference init, trainer
`mcmc` preset = trained-like small splats), 1920x1080, SH degree 3, ONE view per GPU per step;
a step = forward + photometric loss + backward + fused Adam (+ densification statistics), i.e.
one iteration of Runner.train (/root/reference/utils/gsplat_utils/gsplat_trainer.py:551-763).
As the reference samples a new view every iteration (:561-572), every step stages ANOTHER camera and target
image: 8 ring cameras (camera 0 = the front camera of BASELINE.md) and 8 resident targets per GPU, cycled.
N GPUs = N views per step, replicated Gaussians, one RCCL all-reduce of the gradients (weak scaling).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue roof: 256 CUs x 4 SIMD-32 units, one wave64 vector instruction per 2 cycles per SIMD at 2.4 GHz
# (MI355X_MICROARCH.md, "A wave (64 lanes) ... issues each VALU instruction over 2 cycles"; v_fma_f32 2 cyc throughput)
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2.0


def algorithmic_bytes(N, V, I, P, K):
    """algorithmic HBM bytes per launch of each kernel family (fp32, one view)."""
    return {
        "so_projection_fwd": 40 * N + 28 * N,
        "so_sh_fwd": V * (12 * K + 12),
        "so_isect_count": 12 * N + 12 * I,             # first half of the 24 B/intersection binning term
        "so_isect_fill": 12 * N + 12 * I,
        "so_rasterize_fwd": 40 * I + 20 * P,
        "so_rasterize_bwd": 24 * P + 40 * I + 36 * I,
        "so_sh_bwd": V * (12 + 12 * K),
        "so_projection_bwd": 76 * V + 40 * N,
        "so_adam_step": N * (11 + 3 * K) * 28,
    }


def cpu_baseline(n, width, height, regime, seconds_budget=20.0):
    """The oracle (torch fp32 projection/SH/binning + plain-C fp32 rasteriser, OpenMP) timed on the
    host cores for the SAME workload: full iterations (fwd + loss + bwd + torch Adam)."""
    from oracle import torch_oracle as O, c_oracle as CO
    from splat_one_amd.scene import make_scene
    from oracle.ssim_oracle import photometric_loss as oracle_loss
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    splats, c2w, Ks = make_scene(n, width, height, regime=regime)
    params = {k: v.clone().requires_grad_(True) for k, v in splats.items()}
    opt = torch.optim.Adam(params.values(), lr=1e-3, eps=1e-15)
    viewmats = torch.linalg.inv(c2w)
    g = torch.Generator().manual_seed(7)
    pixels = torch.rand(1, height, width, 3, generator=g)
    raster = CO.raster_fn()
    times, t_start = [], time.time()
    fwd_times = []
    while True:
        t0 = time.time()
        colors = torch.cat([params["sh0"], params["shN"]], 1)
        rc, ra, meta = O.rasterization(params["means"], params["quats"], torch.exp(params["scales"]),
                                       torch.sigmoid(params["opacities"]), colors, viewmats, Ks, width, height,
                                       sh_degree=3, near_plane=0.01, far_plane=1e8, raster_fn=raster,
                                       dtype=torch.float32)
        t1 = time.time()
        loss, _, _ = oracle_loss(rc, pixels, 0.2, dtype=torch.float32)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        t2 = time.time()
        times.append(t2 - t0)
        fwd_times.append(t1 - t0)
        if len(times) >= 3 and (time.time() - t_start > seconds_budget or len(times) >= 25):
            break
    times_s = sorted(times[1:])
    med = times_s[len(times_s) // 2]
    fwd = sorted(fwd_times[1:])[len(fwd_times[1:]) // 2]
    return {"value": 1.0 / med, "unit": "it/s", "cores": threads, "kind": "port",
            "sample": f"{len(times) - 1} full iterations (median) of the same {n}-Gaussian {width}x{height} "
                      f"{regime} workload: torch fp32 projection/SH/binning/loss/Adam + oracle/c/raster_oracle.c (f32, OpenMP)",
            "forward_mpix_per_s": width * height / fwd / 1e6}

OTHER_CONFIGS = {
    # key: (flags, what)
    "ref": (["--regime", "ref", "--steps", "30", "--warmup", "5"],
            "c2 in the reference's default preset (init_scale 1.0, init_opa 0.1)"),
    "c3_share": (["--gaussians", "500000", "--steps", "30", "--warmup", "5"],
                 "configs[2], one GPU's share: 500k Gaussians, 1080p, one view per step"),
    "c4_densify": (["--gaussians", "1000000", "--width", "2560", "--height", "1440", "--densify", "100", "--steps", "100"],
                   "configs[3]: 1M Gaussians, 1440p, DefaultStrategy refining every 100 iterations on the device"),
    "c5_share_f16": (["--gaussians", "2000000", "--attr-dtype", "f16", "--steps", "20", "--warmup", "5"],
                     "configs[4], one GPU's share: 2M Gaussians, float16 attribute rows, pinhole view"),
}


def other_configs(timeout_s=300):
    """Compact sub-results of the other BASELINE configurations: {key: {it_s, ms_per_step, hbm_iter_fraction, dominant kernel and
    its HBM fraction, ...}}, each from one child process `python bench.py <flags> --no-cpu-baseline --no-operator-path
    --no-other-configs` (started by PID-less subprocess.run: the parent waits; the GPU is shared sequentially)."""
    import subprocess
    res = {}
    for key, (flags, what) in OTHER_CONFIGS.items():
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--no-cpu-baseline", "--no-operator-path", "--no-other-configs"]
        t0 = time.time()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = next((l for l in reversed(pr.stdout.splitlines()) if l.startswith("{")), None)
            if pr.returncode != 0 or line is None:
                res[key] = {"what": what, "error": f"exit code {pr.returncode}: {pr.stderr.strip().splitlines()[-1:] or ''}"}
                continue
            j = json.loads(line)
            rf = j["roofline"]
            res[key] = {"what": what, "flags": " ".join(flags), "it_s": j["value"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                        "hbm_iter_fraction": j["hbm_iter_fraction"], "dominant_kernel": rf["kernel"],
                        "dominant_kernel_us": rf["mean_launch_us"], "dominant_hbm_frac": rf["frac"], "bound": rf["bound"],
                        "traffic": rf.get("traffic"), "valu_frac": (rf.get("valu") or {}).get("frac"),
                        "tile_intersections": j["config"]["tile_intersections"], "visible_gaussians": j["config"]["visible_gaussians"],
                        "backward_rasteriser": j["config"]["backward_rasteriser"], "void_steps": j["void_steps"],
                        "gaussians_after": (j.get("densify") or {}).get("gaussians_after"),
                        "wall_s": round(time.time() - t0, 1)}
        except subprocess.TimeoutExpired:
            res[key] = {"what": what, "error": f"timed out after {timeout_s} s"}
        except Exception as e:   # noqa: BLE001
            res[key] = {"what": what, "error": repr(e)}
    return res


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher (gsplat_trainer.py:998 `cli(main, cfg)` spawns one process per GPU the
    same way): start N children of THIS script with torchrun's environment contract, one rank per GPU.  The parent makes
    no HIP call (it only waits), children inherit stdout / stderr (rank 0 prints the JSON line).  If a rank fails the
    others are terminated -- by PID -- and its exit code becomes ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                SPLAT_ONE_AMD_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    print("[bench] started ranks: pids " + " ".join(str(p.pid) for p in procs), file=sys.stderr)
    rc = 0
    alive = set(range(n_ranks))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"[bench] rank {r} exited with code {code}: stopping the other ranks", file=sys.stderr)
                    for q in alive:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc


def rank_report(dist, local_rank, dev_index):
    """What the process group itself reports, gathered from every rank: backend, world size, and the device each rank
    drives (index, name, PCI bus id / uuid when torch exposes them) -- config.rccl of the JSON line."""
    mine = {"rank": dist.get_rank(), "local_rank": local_rank, "pid": os.getpid(), "device_index": dev_index}
    if torch.cuda.is_available() and dev_index is not None:
        pr = torch.cuda.get_device_properties(dev_index)
        mine["device_name"] = pr.name
        for k in ("uuid", "pci_bus_id", "pci_device_id", "gcnArchName"):
            v = getattr(pr, k, None)
            if v is not None:
                mine[k] = str(v)
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine)
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": every,
           "self_launched": os.environ.get("SPLAT_ONE_AMD_SELF_LAUNCHED") == "1"}
    if dist.get_backend() == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            pass
        out["distinct_devices"] = len({r.get("uuid") or r.get("pci_bus_id") or r["device_index"] for r in every})
    return out


XGMI_GBS_PER_LINK_DIRECTION = 76.8      # 153.6 GB/s per link, both directions together; 7 links per GPU, fully connected
XGMI_EFFICIENCY = 0.8                   # what a large RCCL transfer is assumed to reach of it
RCCL_LAUNCH_US = 15.0                   # per grouped collective launch (latency floor), assumed


def comm_model(N, K, n_chunks, t_step_ms=None, t_bwd_rows_ms=None, t_stage_ms=0.005):
    """Predicted reduce-scatter / all-gather time per optimiser step of the replicated scheme at 2 / 4 / 8 ranks, from bytes
    per xGMI link and direction -- printed next to the measured config.comm_ms so that the first
    multi-GPU run explains itself.  Direct exchange on the fully connected mesh: piece j of every rank's rows goes straight
    to rank j, so a link carries rows x 4 (11 + 3 K) bytes / world per phase and direction.
    With t_step_ms (one GPU's step) and t_bwd_rows_ms (its per-Gaussian backward, the kernel the reduce-scatters run under):
    what stays EXPOSED -- the last chunk's reduce-scatter plus whatever of the others the backward chunks cannot cover; the
    all-gather minus the next iteration's staging launch (its tail runs under so_step_inputs: `RowShardedAdam.finish(
    defer_gather_wait=True)`) -- the extra HBM round trip of the un-fused gradients (2 x 4 (11 + 3 K) N bytes at ~4 TB/s), and
    `scaling_efficiency_predicted` = t_step / (t_step + exposed + un-fused)."""
    floats_per_row = 11 + 3 * K
    out = {"assumptions": {"xgmi_GBs_per_link_direction": XGMI_GBS_PER_LINK_DIRECTION, "efficiency": XGMI_EFFICIENCY,
                           "launch_us_per_grouped_collective": RCCL_LAUNCH_US, "pattern": "direct (piece j -> rank j), all links busy",
                           "collectives_per_phase": 2 * n_chunks, "one_gpu_step_ms": t_step_ms, "backward_rows_ms": t_bwd_rows_ms,
                           "staging_ms_the_gather_tail_runs_under": t_stage_ms, "unfused_gradient_round_trip_TBs": 4.0}, "by_world": {}}
    for w in (2, 4, 8):
        q = w * 64 * n_chunks
        span = -(-N // q) * q
        per_link = span * floats_per_row * 4.0 / w
        ms = per_link / (XGMI_GBS_PER_LINK_DIRECTION * XGMI_EFFICIENCY * 1e9) * 1e3 + 2 * n_chunks * RCCL_LAUNCH_US * 1e-3
        e = {"rows_exchanged": span, "bytes_per_link_and_direction_per_phase": per_link,
             "reduce_scatter_ms": ms, "all_gather_ms": ms, "ring_bytes_per_link_per_phase": per_link * (w - 1)}
        if t_step_ms:
            cover = (t_bwd_rows_ms or 0.0) * (n_chunks - 1) / n_chunks          # backward chunks 1 .. n-1 run over reductions 0 .. n-2
            rs_exposed = ms / n_chunks + max(0.0, ms * (n_chunks - 1) / n_chunks - cover)
            ag_exposed = max(0.0, ms - t_stage_ms)
            unfused = 2.0 * N * floats_per_row * 4.0 / 4.0e12 * 1e3
            e.update(reduce_scatter_exposed_ms=rs_exposed, all_gather_exposed_ms=ag_exposed, unfused_gradient_ms=unfused,
                     scaling_efficiency_predicted=t_step_ms / (t_step_ms + rs_exposed + ag_exposed + unfused))
        out["by_world"][str(w)] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rewarm-steps", type=int, default=None,
                    help="untimed iterations between the read-backs that precede the warm-up and the --warmup steps (counted in "
                         "config.preparation_steps); default: at least 16 and at least 50 ms of them")
    ap.add_argument("--n", "--gaussians", dest="n", type=int, default=100_000)   # (--gaussians: torchrun's own parser trips over "--n")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--regime", default="mcmc", choices=["mcmc", "ref"])
    ap.add_argument("--camera-model", default="pinhole", choices=["pinhole", "fisheye", "spherical"],
                    help="not a BASELINE configuration (always 'custom'): the same step through another camera model; spherical = the reference's "
                         "default (360-degree equirectangular images; give --width 2 x --height), cameras INSIDE the cloud")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--operator-path", action="store_true",
                    help="time the operator-level autograd path instead of the fused engine")
    ap.add_argument("--no-operator-path", action="store_true",
                    help="skip the extra timing of the operator-level autograd path (operator_path_it_s of the N=1 line)")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel times to stderr")
    ap.add_argument("--densify", type=int, default=0, metavar="EVERY",
                    help="BASELINE.json configs[3]: DefaultStrategy duplicates / splits / prunes every EVERY iterations "
                         "(reference default 100) inside the timed region; 0 = the reference's first 500 iterations (off)")
    ap.add_argument("--cloud-scale", type=float, default=1.0,
                    help="multiply the initial positions by this factor (< 1: the splats gather in the middle of the image -- an "
                         "unevenly loaded tile grid, as real scenes have; the default workload is the uniform cube of BASELINE.md)")
    ap.add_argument("--scale-spread", type=float, default=0.0,
                    help="not a BASELINE configuration: N(0, S^2) added to every log-scale (per axis) and random rotations -- anisotropic splats whose "
                         "sizes spread over e^(+-2S): the statistics of a trained scene rather than of the random init")
    ap.add_argument("--views", type=int, default=8, help="ring cameras / target images cycled per GPU (the reference draws a new view per step)")
    ap.add_argument("--attr-dtype", default="f32", choices=["f32", "f16"],
                    help="f16: float16 attribute rows (BASELINE.json configs[4]); float32 arithmetic and masters")
    ap.add_argument("--max-gaussians", type=int, default=None,
                    help="capacity (rows) of the device-resident model, Config.max_gaussians; default max(2 N, 2^20)")
    ap.add_argument("--loss-kernels", type=int, default=1, choices=[1, 2],
                    help="fused step: 1 = so_ssim_l1_fused (loss and gradient in one launch); 2 = the so_ssim_l1_fwd/bwd pair")
    ap.add_argument("--dp-mode", default="allreduce", choices=["auto", "gaussian_sharded", "allreduce"],
                    help="multi-GPU scheme of the headline value (ignored at --gpus 1): allreduce (= auto) -- replicated "
                         "Gaussians, reduce-scatter / sharded Adam / all-gather of the gradient SoA, BASELINE.json's "
                         "north_star; gaussian_sharded -- the reference's own scheme (projected Gaussians exchanged by "
                         "all-to-all).  The other scheme is timed too and reported in config.other_scheme")
    ap.add_argument("--targets", default="teacher", choices=["teacher", "noise"],
                    help="teacher (default): the target images are renders of the frozen INITIAL model at the ring views and the "
                         "student starts from the same geometry with re-drawn colours -- full-size coherent gradients, and a "
                         "workload (tile intersections) that stays put however many steps are run; noise: seeded U[0,1) images "
                         "(rounds 1-4: the splats grow against them, I drifts ~20 %% over 200 steps)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload only: skip the compact sub-results of the other BASELINE configurations "
                         "(other_configs: ref, c3_share, c4_densify, c5_share_f16 -- each a child process of this script)")
    ap.add_argument("--step-trace", action="store_true", help="diagnostic: an event per timed step, gaps printed to stderr")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, build the process group, print what it reports (config.rccl) and exit: no kernel "
                         "runs (the CPU test of the self-launcher; works without a GPU over gloo)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="--launch-check: this rank exits with code 3 (failure propagation)")
    ap.add_argument("--fail-late", action="store_true",
                    help="--launch-check --fail-rank R: rank R fails AFTER the group is up and has run its first collectives "
                         "(the others are then inside a barrier it never joins)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start the N ranks ourselves, before anything in this process touches the GPU
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.densify:      # warm up past the first two refinements: both model sets' graphs are captured before timing
        args.warmup = max(args.warmup, 2 * args.densify + 1)

    from splat_one_amd import _lib, distributed as sdist
    from splat_one_amd.scene import pinhole_K, ring_cameras, front_camera
    from splat_one_amd.trainer import Config, Runner

    n_dev = torch.cuda.device_count()           # (does not initialise HIP)
    if args.gpus > 1 and n_dev < args.gpus and not os.environ.get("SPLAT_ONE_AMD_BACKEND"):
        # fewer devices than ranks: RCCL refuses two ranks on one device.  A REHEARSAL of the N-rank flow over gloo, ranks
        # sharing devices -- flagged in config.rccl.backend; its numbers say nothing about xGMI
        os.environ["SPLAT_ONE_AMD_BACKEND"] = "gloo"
        if os.environ.get("RANK", "0") == "0":
            print(f"[bench] --gpus {args.gpus} on a box with {n_dev} device(s): falling back to the gloo backend "
                  f"(functional rehearsal, ranks share devices)", file=sys.stderr)
    local_rank, rank, world = sdist.init_from_env()
    assert world == args.gpus, (f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                                f"--nproc-per-node {args.gpus}, or with no launcher at all")
    if args.launch_check:
        if rank == args.fail_rank and not args.fail_late:
            sys.exit(3)
        rep = rank_report(dist, local_rank, local_rank if n_dev else None) if world > 1 else None
        if world > 1 and args.fail_late and args.fail_rank >= 0:
            dist.barrier()                       # the group is up and has worked
            if rank == args.fail_rank:
                os._exit(3)                      # dies without leaving the group: the others wait in the barrier below
            time.sleep(0.5)
        if rank == 0 and not (args.fail_late and args.fail_rank >= 0):
            print(json.dumps({"launch_check": True, "n_gpus": world, "config": {"rccl": rep}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path exists for the product)"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    rccl = rank_report(dist, local_rank, local_rank) if world > 1 else None

    W, H, N = args.width, args.height, args.n
    init_scale, init_opa = (1.0, 0.1) if args.regime == "ref" else (0.1, 0.5)
    # views: step i of rank r renders ring camera (i mod 8) * world + r against its own resident target image
    NV = max(1, args.views)
    ring = ring_cameras(NV * world) if NV * world > 1 else front_camera()[None]
    if args.camera_model == "spherical":      # panorama cameras on a circle of radius 1 inside the cloud, each turned about the vertical axis
        ring = []
        for k in range(NV * world):
            th = 2 * math.pi * k / (NV * world)
            c2w = torch.eye(4)
            c2w[0, 0], c2w[0, 2], c2w[2, 0], c2w[2, 2] = math.cos(2 * th), math.sin(2 * th), -math.sin(2 * th), math.cos(2 * th)
            c2w[:3, 3] = torch.tensor([math.sin(th), 0.3 * math.cos(3 * th), math.cos(th)])
            ring.append(c2w)
        ring = torch.stack(ring)
    g = torch.Generator().manual_seed(100 + rank)
    teacher = args.targets == "teacher" and not args.densify
    if args.densify:      # a smooth target (a shifted colour ramp per view): gradients that make densification grow the set
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
        targets = [torch.stack([(xx + 0.1 * v) % 1.0, (yy + 0.07 * v) % 1.0, 0.5 * (xx + yy)], -1)[None].contiguous().to(dev)
                   for v in range(NV)]
    elif teacher:
        targets = None    # rendered from the runner's own initial model (make_runner)
    else:
        targets = [torch.rand(1, H, W, 3, generator=g).to(dev) for _ in range(NV)]
    K1 = pinhole_K(W, H)[None]

    def teacher_targets(r):
        """A STATIONARY workload: the targets are renders of the frozen initial model at this rank's ring
        views; the student keeps that geometry and gets its colours re-drawn (sh0 + N(0, 0.5^2), shN ~ N(0, 0.1^2), other seed),
        so the loss has full-size, coherent gradients, its optimum keeps the splats where they are, and the tile-intersection
        count no longer depends on how many steps ran before or inside the timed region."""
        out = []
        with torch.no_grad():
            for v in range(NV):
                cam = ring[v * world + rank:v * world + rank + 1].contiguous().to(dev)
                rc, _, _ = r.rasterize_splats(camtoworlds=cam, Ks=K1.to(dev), width=W, height=H, sh_degree=3,
                                              near_plane=r.cfg.near_plane, far_plane=r.cfg.far_plane)
                out.append(rc[..., :3].detach().clone().contiguous())
            gs = torch.Generator().manual_seed(4242 + (rank if r.sharded else 0))
            r.splats["sh0"].add_((torch.randn(r.splats["sh0"].shape, generator=gs) * 0.5).to(dev))
            r.splats["shN"].copy_((torch.randn(r.splats["shN"].shape, generator=gs) * 0.1).to(dev))
        torch.cuda.synchronize()
        r._workspace.clear()         # (the operator path's buffers of these renders are not kept)
        torch.cuda.empty_cache()
        return out

    def make_runner(dp_mode):
        cfg = Config(init_num_pts=N, init_scale=init_scale, init_opa=init_opa, batch_size=1, shN_init_std=0.1,
                     camera_model=args.camera_model, sh_degree_interval=1,    # SH degree 3 from step 3 on
                     fused=not args.operator_path, dp_mode=dp_mode, attr_dtype=args.attr_dtype,
                     loss_kernels=args.loss_kernels, max_gaussians=args.max_gaussians)
        if args.densify:
            from splat_one_amd.strategy import DefaultStrategy
            cfg.strategy = DefaultStrategy(refine_start_iter=0, refine_every=args.densify, reset_every=3000, verbose=False)
        else:
            # a fixed model: the strategy's per-iteration hooks run (the densification statistics are gathered, as in the reference's
            # step), but it never refines -- the reference's default starts at iteration 500, which the untimed passes of a long
            # run would cross (and then N, V and I of the counters pass are no longer those of the timed region)
            cfg.strategy.refine_start_iter = 10 ** 9
            if hasattr(cfg.strategy, "reset_every"):
                cfg.strategy.reset_every = 10 ** 9          # (... nor resets the opacities at iteration 3000 of a very long run)
        r = Runner(local_rank, rank, world, cfg, scene_scale=1.0 / 1.1)   # effective scene_scale 1.0
        if args.cloud_scale != 1.0:
            with torch.no_grad():
                r.splats["means"].mul_(args.cloud_scale)
        if args.scale_spread > 0.0:      # anisotropic splats of very different sizes (a trained scene's statistics rather than the random init's)
            with torch.no_grad():
                gs = torch.Generator().manual_seed(777)
                r.splats["scales"].add_((torch.randn(r.splats["scales"].shape, generator=gs) * args.scale_spread).to(r.splats["scales"].device))
                r.splats["quats"].copy_(torch.randn(r.splats["quats"].shape, generator=gs).to(r.splats["quats"].device))
        tg = teacher_targets(r) if teacher else targets
        if r.sharded:      # the step takes the cameras of every rank (rank r renders camera r of the step's group)
            views = [(ring[v * world:(v + 1) * world].contiguous().to(dev), K1.repeat(world, 1, 1).to(dev), tg[v]) for v in range(NV)]
        else:
            views = [(ring[v * world + rank:v * world + rank + 1].contiguous().to(dev), K1.to(dev), tg[v]) for v in range(NV)]
        return cfg, r, views

    class Stepper:
        """One training iteration per call, on the next view of the cycle."""
        def __init__(self, runner, views):
            self.runner, self.views, self.i = runner, views, 0

        def __call__(self):
            c2w, Ks, px = self.views[self.i % len(self.views)]
            self.i += 1
            return self.runner.train_step(c2w, Ks, px)

    def probe(mode, n_warm=8, n_timed=16):
        """(seconds per step, runner tuple) of one scheme, max over ranks; (inf, None) if any rank failed."""
        ok = torch.ones(1, device=dev)
        tup, dt = None, float("inf")
        try:
            tup = make_runner(mode)
            st = Stepper(tup[1], tup[2])
            for _ in range(n_warm):
                st()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(n_timed):
                st()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n_timed
        except Exception as e:   # noqa: BLE001
            print(f"[bench] dp_mode {mode} failed on rank {rank}: {e!r}", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        t = torch.tensor([dt if ok.item() else 1e30], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return (float(t.item()), tup) if ok.item() else (float("inf"), None)

    dp_probe = None
    other_mode = None
    sdist.COMM_TIMING = world > 1       # phase events around the reduce-scatter / Adam / all-gather of every step (config.comm_ms)

    def comm_objects(r):
        return [o for o in (getattr(r, "_radam", None), getattr(r, "_sadam", None)) if o is not None and o.timer is not None]

    if world == 1:
        cfg, runner, views = make_runner("allreduce")
    else:
        # BASELINE.json's north_star scheme -- replicated Gaussians, gradients reduced over xGMI -- is the headline
        # unless --dp-mode asks for the reference's Gaussian sharding; the OTHER scheme is timed for the same --steps
        # after the headline run and reported next to it (config.other_scheme).  A scheme that fails on any rank is
        # replaced by the other one on every rank.
        first = "gaussian_sharded" if args.dp_mode == "gaussian_sharded" else "allreduce"
        other_mode = "allreduce" if first == "gaussian_sharded" else "gaussian_sharded"
        t_one, tup = probe(first)
        dp_probe = {first + "_ms": None if tup is None else t_one * 1e3}
        if tup is None:
            first, other_mode = other_mode, None
            t_one, tup = probe(first)
            dp_probe[first + "_ms"] = None if tup is None else t_one * 1e3
        assert tup is not None, "both multi-GPU schemes failed"
        cfg, runner, views = tup
    step_once = Stepper(runner, views)
    c2w, Ks, pixels = views[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fused = cfg.fused
    lib = _lib.load()

    # preparation (untimed, before the W warm-up steps and whatever W is): two cycles over the views, so that the SH-degree
    # ramp is over, every hipGraph the timed region replays is captured and the per-tile bins have seen every view -- the
    # counterpart of a compile step; a small --warmup then times steady-state iterations, not captures
    preparation_steps = 0
    if world == 1 and not args.densify:
        preparation_steps = 2 * len(views)
        for _ in range(preparation_steps):
            step_once()
    def measure_forward():
        """forward-only rate: the eval / viewer render (projection + SH + binning + sort + rasterise), seconds per render"""
        if getattr(runner, "_engine", None) is None and fused:
            step_once()                       # (the engine is built by the first training step)
        if runner.sharded:
            fwd_call = lambda: None           # a render needs every shard on one rank: not a per-step quantity here
        elif fused:
            fwd_call = runner._engine.render
        else:
            def fwd_call():
                with torch.no_grad():
                    runner.rasterize_splats(c2w, Ks, W, H, sh_degree=3, near_plane=cfg.near_plane, far_plane=cfg.far_plane)
        for _ in range(5):
            fwd_call()
        barrier()
        t0_ = time.time()
        for _ in range(50):
            fwd_call()
        barrier()
        _lib.profile_summary()  # discard
        return (time.time() - t0_) / 50

    def last_I():
        """tile intersections of the last iteration (synchronises; outside the timed region)"""
        info = getattr(runner, "last_info", None) or {}
        if "engine" in info:
            return info["engine"].stats()["n_isects"]
        if "n_isects" in info:
            return int(info["n_isects"].item())
        return int(info["flatten_ids"].numel()) if "flatten_ids" in info else None

    # Everything that idles the GPU or reads it back -- the forward-only timing, the workload counters before the timed region --
    # happens BEFORE the warm-up, so that the W warm-up steps run straight into the barrier of the timed region (round 5,
    # bench.py --step-trace: after an idle gap the first ~30 iterations run 5 % slower, 278 against 263 us).  Runs with
    # densification measure the forward after their warm-up refinements instead (the model it renders is the grown one).
    fwd_s = None if args.densify else measure_forward()
    I_start = None if args.densify else last_I()
    if preparation_steps:
        # ... and the GPU gets its load back before the W warm-up steps: after the read-backs above it runs ~5 % slower for its
        # next ~30 iterations (--step-trace), more than a small --warmup covers.  Untimed, named in the line.
        # (how long: tools/gpu_r05_m.sh -- with 16 such iterations (4 ms) the 20 timed steps of the driver's command line run at
        # 268 us, with 64 at 263, with 128 or 256 (35 / 70 ms) at 258, the long-run rate: the card needs ~40 ms of load to
        # reach its clocks.  Default: at least 16 iterations and at least 50 ms; the host runs at most two iterations ahead of the
        # GPU (engine.py: the status words), so its clock is the GPU's here.)
        # With several ranks every rank must run the same count (the steps hold collectives): a fixed 128 there.
        n_rewarm, t_rewarm = 0, time.time()
        fixed = args.rewarm_steps if args.rewarm_steps is not None else (128 if world > 1 else None)
        while (n_rewarm < fixed) if fixed is not None else (n_rewarm < 16 or (time.time() - t_rewarm < 0.05 and n_rewarm < 1024)):
            step_once()
            n_rewarm += 1
        preparation_steps += n_rewarm

    # warm-up
    if not fused:
        _lib.PROFILE = "all"     # operator path: per-entry-point HIP events from Python
    for i in range(max(1, args.warmup)):
        step_once()
        if args.densify and fused and i == args.densify - 2 and getattr(runner._engine, "device_refine", False):
            # densification that actually grows the set: refine the top 40 % of the accumulated screen-space
            # gradient at every refinement (one read of the statistics, in the warm-up)
            eng = runner._engine
            n_live = eng.sync_host()
            avg = (eng.dstats["grad2d"][:n_live] / eng.dstats["count"][:n_live].clamp_min(1))
            thr = torch.quantile(avg[torch.randperm(n_live, device=dev)[:1_000_000]], 0.6).reshape(1)
            if world > 1:            # replicas must take the same decisions: rank 0's threshold (its own views' statistics)
                dist.broadcast(thr, src=0)
            cfg.strategy.grow_grad2d = float(thr)
    torch.cuda.synchronize()
    if not fused:
        prof = _lib.profile_summary()
        dominant = max(((k, v) for k, v in prof.items() if k not in ("so_rec_pack", "so_rec_unpack_grads", "so_camera_inverse")),
                       key=lambda kv: kv[1][0] * kv[1][1])[0]
        _lib.PROFILE = {dominant}

    # timed region: EXACTLY --steps iterations between barriers
    n_before_timed = runner._engine.sync_host() if (fused and not runner.sharded and getattr(runner._engine, "device_refine", False)) else None
    void0 = getattr(getattr(runner, "_engine", None), "void_steps", 0)

    if args.densify:
        fwd_s = measure_forward()
        I_start = last_I()
    for o in comm_objects(runner):
        o.timer.clear()
    barrier()
    t0 = time.time()
    if args.step_trace:      # (diagnostic: one event per step -- where inside the timed region does the time go?)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        host_t = []
        evs[0].record()
        for i in range(args.steps):
            step_once()
            evs[i + 1].record()
            host_t.append(time.time() - t0)
    else:
        for _ in range(args.steps):
            step_once()
    barrier()
    elapsed = time.time() - t0
    if args.step_trace and rank == 0:
        gaps = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(args.steps)]
        print("[step-trace] us between consecutive step-end events:", [round(g_, 1) for g_ in gaps[:24]], "... mean of the rest",
              round(sum(gaps[24:]) / max(1, len(gaps[24:])), 1), file=sys.stderr)
        print("[step-trace] host time at which step i had been issued (us):", [round(h * 1e6) for h in host_t[:24]],
              " whole region (us):", round(elapsed * 1e6), file=sys.stderr)
    I_end = last_I()
    comm_ms = None
    for o in comm_objects(runner):          # mean per step over exactly the timed iterations (this rank's stream)
        comm_ms = o.comm_ms()
        o.timer = None
    void_steps = getattr(getattr(runner, "_engine", None), "void_steps", 0) - void0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if runner.sharded:
        # the sharded step is a sequence of C-ABI calls from Python: time each entry point with HIP events
        _lib.PROFILE = "all"
        _lib.profile_summary()
        for _ in range(args.steps):
            step_once()
        prof = {k.replace("_packed", ""): v for k, v in _lib.profile_summary().items()}
        _lib.PROFILE = None
        # (the first launch after isect_tiles' host read starts on an idle GPU: its event pair also spans the host time up
        # to the launch, so the small record-packing helpers are not candidates for "dominant kernel")
        helpers = ("so_rec_pack", "so_rec_unpack_grads", "so_camera_inverse")
        dominant = max(((k, v) for k, v in prof.items() if k not in helpers), key=lambda kv: kv[1][0] * kv[1][1])[0]
        dom_calls, dom_ms = prof[dominant]
    elif fused:
        # Per-kernel durations with HIP events on the launch stream.  Events cannot be recorded
        # inside a hipGraph replay, so the same --steps iterations are re-run un-captured with the
        # library's stage timers on (same kernels, same inputs; parameters keep training).
        eng = runner._engine
        saved_graph_flag = eng.use_graph
        eng.use_graph = False
        lib.so_profile_enable(1)
        for _ in range(args.steps):
            step_once()
        prof = _lib.stage_profile()
        lib.so_profile_enable(0)
        eng.use_graph = saved_graph_flag
        dominant = max(prof.items(), key=lambda kv: kv[1][0] * kv[1][1])[0]
        dom_calls, dom_ms = prof[dominant]
    else:
        dom_calls, dom_ms = _lib.profile_summary()[dominant]
        _lib.PROFILE = None
    if args.kernel_table and rank == 0:
        tot = sum(n * ms for n, ms in prof.values())
        for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            print(f"  {k:24s} calls {n:5d}  mean {ms * 1e3:9.1f} us  share {n * ms / tot:5.1%}", file=sys.stderr)

    # cost of ONE refinement on the device (after the timed region; HIP events around the five launches)
    refine_ms = None
    if args.densify and fused and not runner.sharded and getattr(runner._engine, "device_refine", False):
        eng = runner._engine
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        runner.refine_on_device(runner.step)          # (with --gpus N: incl. the statistics / moments collectives and the read of N)
        e1.record()
        torch.cuda.synchronize()
        refine_ms = e0.elapsed_time(e1)
        step_once()
    if args.densify and not (fused and getattr(runner._engine, "device_refine", False)):
        step_once()       # a host-side refinement rebuilt the workspace: read the workload counters after a plain iteration
        while (runner.step - 1) % args.densify == 0:
            step_once()
    # workload counters: mean over the views of the cycle (one more pass, each read back)
    N0, N = N, (N if runner.sharded else int(runner.splats["means"].shape[0]))     # densification changes N
    Vs, Is = [], []
    for _ in range(len(views)):
        step_once()
        info = runner.last_info
        if "engine" in info:
            st_ = info["engine"].stats()
            Vs.append(st_["visible"])
            Is.append(st_["n_isects"])
        else:
            n_rows = info["radii"].shape[-1]
            Vs.append(int((info["radii"] > 0).sum().item()))
            Is.append(int(info["flatten_ids"].numel()) if "n_isects" not in info else int(info["n_isects"].item()))
    V, I = int(sum(Vs) / len(Vs)), int(sum(Is) / len(Is))
    P = W * H
    K = (cfg.sh_degree + 1) ** 2
    ab = algorithmic_bytes(N, V, I, P, K)
    ab["so_preprocess_fwd"] = ab["so_projection_fwd"] + ab["so_sh_fwd"] + 12 * N          # fused K1+K4+count
    ab["so_preprocess_bwd"] = ab["so_projection_bwd"] + ab["so_sh_bwd"]                    # fused K2+K5
    ab["so_adam_step_dev"] = ab["so_adam_step"]
    ab["so_isect_scan"] = 8 * (W // 16 + 1) * (H // 16 + 1)
    ab["so_tile_order"] = 8 * (W // 16 + 1) * (H // 16 + 1)        # list lengths read, workgroup -> tile table written
    ab["so_bins_gather"] = 16 * I                                   # replicated bin counters (small images): keys read and written once
    ab["so_ssim_l1_fwd"] = 24 * P + 36 * P
    ab["so_ssim_l1_bwd"] = 60 * P + 12 * P
    ab["so_ssim_l1_fused"] = 24 * P + 12 * P          # two images in, one gradient image out
    b_iter = sum(ab[k] for k in ("so_projection_fwd", "so_sh_fwd", "so_isect_count", "so_isect_fill", "so_rasterize_fwd",
                                 "so_rasterize_bwd", "so_sh_bwd", "so_projection_bwd", "so_adam_step"))
    dominant = dominant.replace("_packed", "")          # the packed-record entry points share the byte model
    traffic = None
    profile_notes = {}

    def collected_near(j, what, tol=0.05):
        """A committed counter summary applies to this run only if it was collected at (nearly) the same workload state: the
        passes walk tile lists, and their number follows the tile intersections I.  profiles/*.json hold
        the collection of the default command and, under "_also", the one made at the round-end driver's command line
        (--steps 20 --warmup 5): returns the collection within 5 % of this run's I, or None."""
        seen = []
        for col in [j] + list(j.get("_also") or []):
            at = (col.get("_collected_at") or {}).get("tile_intersections")
            if not at:
                continue
            seen.append(at)
            if abs(I - at) <= tol * at:
                profile_notes[what] = f"collected at I = {at} (this run: I = {I})"
                return col
        profile_notes[what] = ("refused: the summary in profiles/ does not say at which tile-intersection count it was collected" if not seen
                               else f"refused: collected at I = {seen}, this run has I = {I} (> {tol:.0%} apart)")
        return None

    def collection_for(j, what):
        """c2: the root collection (or one of its "_also"); any other named workload: j["_by_workload"][key] -- collected by
        tools/gpu_profiles_r05.sh at that workload's own command line (15 % on I: densification moves it inside a run)."""
        if wl_short == "c2":
            return collected_near(j, what)
        col = (j.get("_by_workload") or {}).get(wl_short)
        if col is None:
            profile_notes[what] = f"no counter collection for workload '{wl_short}' in profiles/"
            return None
        return collected_near(col, what, tol=0.15)

    # which BASELINE configuration this is (profiles/*.json keep one counter collection per key under "_by_workload")
    wl = (N0, W, H, bool(args.densify), args.regime if args.camera_model == "pinhole" else args.regime + "/" + args.camera_model, args.attr_dtype,
          args.cloud_scale if args.scale_spread == 0.0 else (args.cloud_scale, args.scale_spread))
    workload_key = {(100_000, 1920, 1080, False, "mcmc", "f32", 1.0): "c2",
                    (100_000, 1920, 1080, False, "ref", "f32", 1.0): "c2-ref (the reference's default preset)",
                    (500_000, 1920, 1080, False, "mcmc", "f32", 1.0): "c3 (per-GPU share: 500k Gaussians, one view)",
                    (1_000_000, 2560, 1440, True, "mcmc", "f32", 1.0): "c4",
                    (2_000_000, 1920, 1080, False, "mcmc", "f16", 1.0): "c5 (per-GPU share: 2M Gaussians, float16 rows, pinhole view)",
                    (2_000_000, 1920, 1080, False, "mcmc", "f32", 1.0): "2M-f32"}.get(wl, "custom")
    wl_short = workload_key.split(" ")[0]
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    # the PMC collections in profiles/ are keyed by workload (c2 at the root; c2-ref, c4, 2M-f32, ... under "_by_workload")
    is_c2_engine = workload_key != "custom" and world == 1 and len(views) == 8 and fused     # (a named workload with a collection of its own)
    if os.path.exists(tpath) and is_c2_engine:
        try:
            tj = collection_for(json.load(open(tpath)), "traffic")
            traffic = tj.get(dominant) if tj else None
        except Exception:
            traffic = None
    # rocprofv3's own average for the dominant kernel (profiles/kernel_us.json <- the committed --kernel-trace --stats summary
    # of this command), next to the live stage timer
    kernel_us_rocprof = None
    kpath = os.path.join(ROOT, "profiles", "kernel_us.json")
    if os.path.exists(kpath) and is_c2_engine:
        try:
            kj = collection_for(json.load(open(kpath)), "kernel_us") or {}
            if dominant in kj:
                kernel_us_rocprof = {"us": kj[dominant], "source": kj.get("_note"), "collected_at": kj.get("_collected_at")}
        except Exception:   # noqa: BLE001
            kernel_us_rocprof = None

    # every timed stage against the HBM roofline (the table of DESIGN.md section 4): which kernels stream at the
    # roofline and which are bound by instruction issue
    by_kernel = {}
    fused_adam = fused and not runner.sharded and "so_adam_step_dev" not in prof and world == 1
    for k, (n_calls, ms) in prof.items():
        kk = k.replace("_packed", "")
        nbytes = ab.get(kk)
        if kk == "so_preprocess_bwd" and fused_adam:      # Adam runs inside: + its traffic, - the gradient round trip
            nbytes = ab["so_preprocess_bwd"] + ab["so_adam_step"] - 8 * N * (11 + 3 * K)
        if kk == "so_preprocess_fwd" and fused and not runner.sharded and runner._engine.binned:
            nbytes = ab["so_preprocess_fwd"] + 8 * I     # the keys are written here (no scatter pass)
        if kk == "so_isect_fill" and fused and not runner.sharded and runner._engine.binned:
            nbytes = 12 * I                                # sort only: 8 B key read + 4 B id written
        if nbytes and ms > 0:
            gbs = nbytes / (ms * 1e-3) / 1e9
            by_kernel[kk] = {"us": round(ms * 1e3, 1), "algorithmic_bytes": int(nbytes), "GB/s": round(gbs, 1),
                             "frac": round(gbs / HBM_PEAK_GBS, 4)}

    # the dominant kernel's algorithmic bytes as the per-kernel table states them (a fused launch carries the traffic of
    # everything fused into it: Adam inside so_preprocess_bwd, the key writes inside so_preprocess_fwd)
    if dominant not in by_kernel and dominant not in ab:      # the operator path's two whole-path entry points
        ab["so_rasterization_fwd"] = ab["so_preprocess_fwd"] + 20 * I + ab["so_rasterize_fwd"]
        ab["so_rasterization_bwd"] = ab["so_rasterize_bwd"] + ab["so_preprocess_bwd"]
    dom_bytes = by_kernel[dominant]["algorithmic_bytes"] if dominant in by_kernel else ab[dominant]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # The other roof of the dominant kernel: vector-instruction issue.  Wave-instructions per launch come from the committed
    # SQ-counter summary of this workload (profiles/valu.json <- rocprofv3 --pmc SQ_INSTS_VALU, tools/gpu_profiles_r03.sh);
    # the launch duration is the one measured live above.  `bound` names the roof with the larger fraction.
    valu = None
    vpath = os.path.join(ROOT, "profiles", "valu.json")
    if os.path.exists(vpath) and is_c2_engine:
        try:
            vj = collection_for(json.load(open(vpath)), "valu") or {}
            ent = vj.get(dominant)
            if ent:
                rate = ent["wave_instructions"] / (dom_ms * 1e-3)
                valu = {"wave_instructions_per_launch": ent["wave_instructions"], "achieved": rate, "peak": VALU_PEAK_WAVE_INSTR_PER_S,
                        "unit": "wave64 VALU instructions/s", "frac": rate / VALU_PEAK_WAVE_INSTR_PER_S,
                        "active_lane_fraction": ent.get("active_lane_fraction"),
                        "useful_lane_fraction": (vj.get("_useful_lane_fraction_model") or {}).get(dominant),
                        "source": "profiles/valu.json (SQ counters of this workload; useful lanes: tools/passsim.py)"}
        except Exception:   # noqa: BLE001
            valu = None
    hbm_frac = achieved / HBM_PEAK_GBS
    bound = "valu" if (valu is not None and valu["frac"] > hbm_frac) else "hbm"
    out = {
        "metric": ("training iters/sec (100k Gaussians, 1080p, fwd+loss+bwd+Adam)" if (N0, W, H) == (100_000, 1920, 1080)
                   else f"training iters/sec ({N0} Gaussians, {W}x{H}, fwd+loss+bwd+Adam)"),
        # N = 1: training iterations per second.  N > 1 (weak scaling, one view per GPU and step): every optimiser step
        # consumes `world` views, so the whole-job aggregate is VIEWS per second (= single-GPU iterations' worth of work per
        # second, the number to divide by the N = 1 value); optimiser steps per second are reported next to it
        "value": world * args.steps / elapsed,
        "unit": "it/s" if world == 1 else "views/s",
        "optimizer_steps_per_s": args.steps / elapsed,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "path": ("sharded engine (C-ABI launches + 2 all-to-all)" if runner.sharded else
                 "fused engine (hipGraph replay)" if fused else "operator-level autograd path"),
        "config": {"workload": f"{workload_key}: {N0} Gaussians "
                               f"(reference random init, '{args.regime}' preset), "
                               f"{W}x{H}, SH degree 3, 1 view per GPU per step ({len(views)} ring cameras and targets cycled), {args.camera_model}"
                               + (f", DefaultStrategy refining every {args.densify} iterations ({n_before_timed or N0} -> {N} Gaussians "
                                  f"over the timed region and the stage-timer pass)" if args.densify else "")
                               + (", float16 attribute rows" if args.attr_dtype == "f16" else "")
                               + (f", initial positions scaled by {args.cloud_scale}" if args.cloud_scale != 1.0 else "")
                               + (f", log-scales spread by N(0, {args.scale_spread}^2) per axis, random rotations" if args.scale_spread > 0.0 else "")
                               + (", targets = renders of the frozen initial model (student: same geometry, re-drawn colours)" if teacher else
                                  (", smooth ramp targets" if args.densify else ", random-noise targets")),
                   "workload_key": workload_key, "targets": ("teacher" if teacher else ("ramp" if args.densify else "noise")),
                   "preparation_steps": preparation_steps,      # untimed, before --warmup: SH ramp, graph captures, bin probes of all views
                   "tile_intersections_timed_region": {("before_first_step" if args.densify else "before_the_warmup_steps"): I_start, "after_last_step": I_end},
                   "views_per_step": world, "views_cycled": len(views), "visible_gaussians": V, "tile_intersections": I,
                   "visible_gaussians_per_view": Vs, "tile_intersections_per_view": Is,
                   "tile_cull": bool(cfg.tile_cull and fused),   # I counts what is left after exact tile culling
                   "binned_lists": bool(fused and not runner.sharded and runner._engine.binned),
                   # per-tile bins: slots per tile, their memory, and whether a skewed view made the engine fall back to the
                   # compact slotted lists (FusedEngine.bin_budget_bytes)
                   "bin_capacity": (int(runner._engine.bin_capacity) if fused and not runner.sharded else None),
                   "bin_memory_mb": (12.0 * runner._engine.M * runner._engine.bin_capacity / 1e6 if fused and not runner.sharded and runner._engine.binned else None),
                   "fell_back_to_compact_lists": bool(fused and not runner.sharded and getattr(runner._engine, "fell_back_to_compact", False)),
                   "backward_rasteriser": ("one wave per 16x16 tile" if fused and not runner.sharded and runner._engine.cfg.get("raster_impl") == 1
                                           else "one wave per 8x8 quadrant"),
                   "tile_order": ({"each": "longest list first (device-built table, one launch per step)",
                                   "build": "longest list first (table kept per view, rebuilt every 16th visit)",
                                   "kept": "longest list first (table kept per view, rebuilt every 16th visit)"}.get(
                                       getattr(runner._engine, "_order_mode", "none"), "XCD-local runs of 8 tiles")
                                  if fused and not runner.sharded else "XCD-local runs of 8 tiles"),
                   "parallelism": (f"gaussian-sharded dp{world}: one view per GPU, N/{world} Gaussians per GPU, "
                                   "projected Gaussians exchanged by all-to-all" if runner.sharded else
                                   f"view-sharded dp{world}" + ("" if world == 1 else
                                                                f", replicated device-resident Gaussians (device-side refinement): reduce-scatter / 1/{world} Adam / "
                                                                f"all-gather over row pieces of the capacity-sized tensors in {getattr(runner, '_dp_chunks', 1)} row chunk(s), two grouped RCCL launches per chunk and phase, each chunk's reduce-scatter under the next chunk's backward kernel"
                                                                if getattr(runner._engine, "device_refine", False) else
                                                                f", replicated Gaussians: reduce-scatter / 1/{world} Adam / all-gather of the "
                                                                f"gradient SoA in {getattr(runner, '_dp_chunks', cfg.dp_chunks)} chunk(s) over RCCL")),
                   "dp_mode_probe_ms_per_step": dp_probe,
                   "rccl": rccl, "comm_ms": comm_ms,
                   "comm_model": comm_model(N, K, getattr(runner, "_dp_chunks", 1) or 1,
                                            t_step_ms=(elapsed / args.steps * 1e3 if world == 1 else None),
                                            t_bwd_rows_ms=((prof.get("so_preprocess_bwd") or (0, 0.0))[1] if world == 1 else None))},
        "forward_mpix_per_s": None if runner.sharded else world * P / fwd_s / 1e6,
        "hbm_iter_fraction": b_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9),
        "algorithmic_bytes_per_iter": b_iter,
        "roofline": {"bound": bound, "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": hbm_frac, "traffic": traffic,
                     "algorithmic_bytes_per_launch": dom_bytes, "mean_launch_us": dom_ms * 1e3,
                     "launches_timed": dom_calls, "valu": valu, "kernel_us_rocprof": kernel_us_rocprof,
                     "profile_notes": profile_notes},
        # BASELINE.json north_star: >= 1000 it/s at c2 on one MI355X AND >= 40 % of the HBM roofline
        "target": {"it_s": 1000, "hbm_frac": 0.40, "it_s_met": bool(world * args.steps / elapsed >= 1000 * world),
                   "hbm_iter_fraction": b_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9),
                   "hbm_frac_met": bool(b_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9) >= 0.40),
                   "dominant_kernel_hbm_frac": hbm_frac},
        "roofline_by_kernel": by_kernel,
        "void_steps": void_steps,
    }
    if world > 1:
        # what config.comm_model expects of this world size, from THIS run's own compute time (step minus the measured waits)
        waits = ((comm_ms or {}).get("reduce_scatter_wait_ms") or 0.0) + ((comm_ms or {}).get("all_gather_wait_ms") or 0.0)
        t_comp = max(elapsed / args.steps * 1e3 - waits, 1e-6)
        cm = comm_model(N, K, getattr(runner, "_dp_chunks", 1) or 1, t_step_ms=t_comp,
                        t_bwd_rows_ms=(prof.get("so_preprocess_bwd") or (0, 0.0))[1])["by_world"].get(str(world))
        out["scaling_efficiency_predicted"] = (cm or {}).get("scaling_efficiency_predicted")
        out["scaling_efficiency_predicted_from"] = {"compute_ms_per_step_this_run": t_comp, "measured_waits_ms": waits, "model": cm}
    if args.densify:
        out["densify"] = {"every": args.densify, "gaussians_before_timed_region": n_before_timed, "gaussians_after": N,
                          "device_side": bool(fused and getattr(runner._engine, "device_refine", False)),
                          "grow_grad2d": cfg.strategy.grow_grad2d, "one_refinement_ms": refine_ms}
    if world > 1 and other_mode is not None:
        del runner
        torch.cuda.empty_cache()
        other = {"dp_mode": other_mode, "value": None, "ms_per_step": None}
        ok = torch.ones(1, device=dev)
        dt = 0.0
        try:
            _cfg2, r2, v2 = make_runner(other_mode)
            st2 = Stepper(r2, v2)
            for _ in range(max(1, args.warmup)):
                st2()
            barrier()
            t0 = time.time()
            for _ in range(args.steps):
                st2()
            barrier()
            dt = time.time() - t0
        except Exception as e:   # noqa: BLE001
            print(f"[bench] dp_mode {other_mode} failed on rank {rank}: {e!r}", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if ok.item():
            other.update(value=world * args.steps / float(tt.item()), ms_per_step=float(tt.item()) / args.steps * 1e3)
        out["config"]["other_scheme"] = other
    if world == 1 and fused and not args.densify and not args.no_operator_path:
        # The same iteration through the reference's own call structure (gsplat_trainer.py:586-742): Runner.rasterize_splats
        # -> rasterization() [one library call each way] -> photometric_loss -> loss.backward() -> step_all -> strategy
        # hooks, under torch autograd, no hipGraph -- what a trainer written against gsplat's API gets after the import swap.
        runner = None
        torch.cuda.empty_cache()
        args.operator_path = True
        _cfg_o, r_o, v_o = make_runner("allreduce")
        st_o = Stepper(r_o, v_o)
        n_o = min(args.steps, 200)
        for _ in range(3 * len(v_o)):          # SH ramp, bin probe of every view, allocator warm-up
            st_o()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n_o):
            st_o()
        torch.cuda.synchronize()
        dt_o = (time.time() - t0) / n_o
        out["operator_path_it_s"] = 1.0 / dt_o
        out["operator_path"] = {"ms_per_step": dt_o * 1e3, "steps": n_o,
                                "what": "Runner.rasterize_splats -> rasterization() (so_rasterization_fwd/_bwd) -> photometric_loss "
                                        "-> backward -> step_all -> DefaultStrategy hooks, torch autograd, no hipGraph"}
        args.operator_path = False
    if world == 1 and workload_key == "c2" and fused and not args.no_other_configs:
        # The other BASELINE configurations next to the headline, each timed by a CHILD of this script
        # (its own process: a failure there cannot take the headline line with it) with the driver's own clock around it:
        # >= 20 timed steps, no CPU baseline, no operator path.
        runner = r_o = st_o = step_once = None        # free the device memory of this process's runners for the children
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["other_configs"] = other_configs()
        ref = out["other_configs"].get("ref") or {}
        out["ref_regime"] = {k: ref.get(k) for k in ("it_s", "ms_per_step", "dominant_kernel", "dominant_kernel_us", "dominant_hbm_frac",
                                                     "valu_frac", "backward_rasteriser", "tile_intersections", "error") if k in ref}
        out["ref_regime"]["what"] = ("the reference's DEFAULT preset (init_scale 1.0, init_opa 0.1: gsplat_trainer.py:117-119, what "
                                     "app/gsplat_manager.py:43-49 runs): ~520 list entries per tile")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N0, W, H, args.regime)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
