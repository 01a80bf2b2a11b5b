"""Training engine of the path -- the slice of the reference's `Runner`
(/root/reference/utils/gsplat_utils/gsplat_trainer.py) that is the hot loop:

    Config                        :62-201   (fields that reach the rasteriser / optimiser / strategy)
    create_splats_with_optimizers :204-281
    Runner.rasterize_splats       :446-497
    Runner.train (one iteration)  :551-763  pre-backward hook -> loss -> backward -> optimisers ->
                                            LR schedule -> post-backward (densification)

Same names, argument meaning and step order, so `app/gsplat_manager.py` style callers
(`Runner(local_rank, world_rank, world_size, cfg)`, `.train()`, `.rasterize_splats(...)`) read the
same.  The callers either side of the loop are here too:
    Runner.from_data_dir          :308-324  Parser/Dataset wiring (splat_one_amd.datasets)
    save / load checkpoints       :682-703, :950-957   ckpt_{step}_rank{r}.pt = {"step", "splats"}
    Runner.eval                   :779-838  PSNR + SSIM over the val split (LPIPS needs downloaded weights)
    Runner.render_traj            :841-901  interp / ellipse / spiral paths, RGB | normalised depth canvas
    Runner._viewer_render_fn      :916-944  one frame for an interactive viewer
Out of scope: the viewer itself, TensorBoard, compression,
pose/appearance/bilateral-grid modules, video encoding (frames are returned / written as PNG).

Multi-GPU: every rank renders its own view.  `Config.dp_mode = "gaussian_sharded"` (default on the
fused path) is the reference's scheme -- Gaussians strided over ranks as `points[world_rank::world_size]`
(:236-238), projected Gaussians exchanged by all-to-all (splat_one_amd/sharded.py); "allreduce" keeps
every Gaussian on every rank and all-reduces the gradients (splat_one_amd.distributed).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import distributed as sdist
from .losses import photometric_loss
from .optimizers import FusedAdam, SelectiveAdam, step_all
from .rendering import rasterization, rasterization_from_parameters
from .ops import camera_inverse
from .scene import knn, rgb_to_sh, set_random_seed
from .strategy import DefaultStrategy, MCMCStrategy


@dataclass
class Config:
    # names and defaults follow gsplat_trainer.py:62-182
    batch_size: int = 1
    steps_scaler: float = 1.0
    max_steps: int = 30_000
    init_type: str = "random"
    init_num_pts: int = 100_000
    init_extent: float = 3.0
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    init_opa: float = 0.1
    init_scale: float = 1.0
    ssim_lambda: float = 0.2
    near_plane: float = 0.01
    far_plane: float = 1e8
    strategy: Union[DefaultStrategy, MCMCStrategy] = field(default_factory=DefaultStrategy)
    packed: bool = False
    sparse_grad: bool = False
    visible_adam: bool = False
    antialiased: bool = False
    random_bkgd: bool = False
    opacity_reg: float = 0.0
    scale_reg: float = 0.0
    global_scale: float = 1.0
    # "pinhole" | "ortho" | "fisheye" | "spherical" (the reference's default, :89: 360-degree equirectangular shots; the
    # fork's kernel is not in the reference tree, so that model is the one csrc/splat_math.hpp defines -- parity
    # unpinned).  On the fused path also a list with one name per view of the batch (mixed batches).
    # None (default): what the shots of Config.data_dir are -- "spherical" for 360-degree shots (= the reference's
    # default on its own data sets), "pinhole" for perspective shots and for synthetic views
    camera_model: Optional[str] = None
    # data / results (gsplat_trainer.py:67-104): consumed by Runner.from_data_dir, eval, render_traj, checkpoints
    ckpt: Optional[List[str]] = None
    render_traj_path: str = "interp"
    data_dir: str = "data_dir"
    data_factor: int = 4
    result_dir: str = "results/"
    test_every: int = 8
    patch_size: Optional[int] = None
    normalize_world_space: bool = True
    eval_steps: List[int] = field(default_factory=lambda: [7_000, 30_000])
    save_steps: List[int] = field(default_factory=lambda: [7_000, 30_000])
    # :573-575, :595, :629-645 -- disparity L1 at SfM points: render_mode "RGB+ED", grid_sample of the expected depth at
    # `points`, |1/d - 1/d_gt| x scene_scale x depth_lambda.  The term runs on the autograd path (train_step(points=...,
    # depths_gt=...)); the OpenSfM data set -- like the reference's, opensfm.py:384-387 -- carries no "points", so
    # Runner.train refuses depth_loss there instead of training without the term
    depth_loss: bool = False
    # camera pose refinement (gsplat_trainer.py:150-156): runs through the operator-level path (viewmat gradients)
    pose_opt: bool = False
    pose_opt_lr: float = 1e-5
    pose_opt_reg: float = 1e-6
    pose_noise: float = 0.0
    # accepted so that the reference's Config(...) calls keep working (app/gsplat_manager.py:43-48 passes
    # disable_viewer=True): the viewer server, TensorBoard and LPIPS are outside this path and ignored; the appearance /
    # bilateral-grid / compression branches are refused by Runner when switched on (gsplat_trainer.py:106-112, 157-182)
    disable_viewer: bool = False
    port: int = 8080
    compression: Optional[str] = None
    app_opt: bool = False
    app_embed_dim: int = 16
    app_opt_lr: float = 1e-3
    app_opt_reg: float = 1e-6
    use_bilateral_grid: bool = False
    bilateral_grid_shape: Tuple[int, int, int] = (16, 16, 8)
    depth_lambda: float = 1e-2
    tb_every: int = 100
    tb_save_image: bool = False
    lpips_net: str = "alex"
    # extensions of this build
    isect_capacity: Optional[int] = None   # preallocated intersections -> no host sync in the step
    # FusedEngine: the whole training step as hipGraph replays of two C-ABI calls, whenever the step has the shape the
    # engine covers (Runner._fused_ok: no masks / random background / packed / pose refinement); False: every step goes
    # through Runner.rasterize_splats -> rasterization() -> photometric_loss -> step_all under torch autograd
    fused: bool = True
    # multi-GPU scheme (world_size > 1, fused path, one view per rank per step):
    #   "gaussian_sharded"  the reference's own: every rank owns N/world Gaussians, projected Gaussians are
    #                       exchanged (two all-to-alls of ~64 B/Gaussian), per-rank densification / checkpoints;
    #   "allreduce"         replicated Gaussians, one all-reduce of the 236 B/Gaussian gradient SoA
    dp_mode: str = "gaussian_sharded"
    # "f16": the fused step reads quaternions, log-scales and SH coefficients from float16 attribute rows (112
    # instead of 224 B per Gaussian; float32 masters + Adam state unchanged, checkpoints hold the masters) --
    # BASELINE.json configs[4].
    attr_dtype: str = "f32"
    # fused path: tiles no pixel of which can reach alpha >= 1/255 are left out of the per-tile lists (exact: images,
    # losses and gradients are unchanged; only the internal lists are shorter than gsplat's)
    tile_cull: bool = True
    # fused path: per-tile lists in fixed-capacity bins (no scan / scatter pass; sized from the first view, enlarged on
    # overflow).  False: gsplat's compact layout (one buffer of Config.isect_capacity entries)
    binned: bool = True
    fuse_adam: bool = True                 # single-GPU fused step: Adam inside the backward kernel (no gradient round trip)
    loss_kernels: int = 1                  # fused step: 1 = loss and its gradient in one launch; 2 = forward/backward pair
    bin_capacity: Optional[int] = None     # slots per tile; None: 8x the fullest tile of the first view, >= 1024
    bin_budget_gb: float = 4.0             # memory the per-tile bins may take; a view that needs more (a cloud gathered in a few
                                           # tiles) switches the engine to the compact slotted lists by itself, with a warning
    # single-GPU fused step with DefaultStrategy: the refinement (duplicate / split / prune / opacity reset) runs as a
    # stream compaction ON THE DEVICE (so_refine_default): capacity-preallocated parameters + Adam moments, N in device
    # memory, counter-based split noise, no host read-back, no graph re-capture.  False: the torch-level strategy ops.
    device_refine: bool = True
    max_gaussians: Optional[int] = None    # capacity of the device-resident model; None: max(2 N, 2^20), doubled when exceeded
    refine_seed: int = 1234                # seed of the split noise (same on every rank)
    dp_chunks: int = 0                     # dp_mode="allreduce": chunks of the reduce-scatter / Adam / all-gather pipeline
    #                                        (0 = by size: 4 from 64 MB of gradient on, i.e. ~285k Gaussians at SH degree 3,
    #                                        else 1 -- below that the latency of eight collectives outweighs the overlap)
    shN_init_std: float = 0.0              # >0: noise instead of zeros in the higher SH bands (bench scenes)
    # Runner.train draws the training views like the reference's DataLoader(shuffle=True) (gsplat_trainer.py:539-546): a
    # fresh permutation per epoch, seeded from the global torch generator exactly as torch's RandomSampler seeds its own.
    # False: views in dataset order (deterministic walks for tests)
    shuffle: bool = True
    # Runner.rasterize_splats hands the RAW parameters to the library (exp / sigmoid / cat inside the kernels) when the call
    # has the common shape.  False: the reference's own composition -- torch exp / sigmoid / cat, then rasterization() with
    # gsplat's signature -- on every call (what a trainer written against gsplat's API executes; bench.py times both)
    raw_params_call: bool = True

    def adjust_steps(self, factor: float):
        """gsplat_trainer.py:184-201"""
        self.eval_steps = [int(i * factor) for i in self.eval_steps]
        self.save_steps = [int(i * factor) for i in self.save_steps]
        self.max_steps = int(self.max_steps * factor)
        self.sh_degree_interval = int(self.sh_degree_interval * factor)
        s = self.strategy
        if isinstance(s, DefaultStrategy):
            s.refine_start_iter = int(s.refine_start_iter * factor)
            s.refine_stop_iter = int(s.refine_stop_iter * factor)
            s.reset_every = int(s.reset_every * factor)
            s.refine_every = int(s.refine_every * factor)
        elif isinstance(s, MCMCStrategy):
            s.refine_start_iter = int(s.refine_start_iter * factor)
            s.refine_stop_iter = int(s.refine_stop_iter * factor)
            s.refine_every = int(s.refine_every * factor)


PARAM_LRS = (("means", 1.6e-4), ("scales", 5e-3), ("quats", 1e-3), ("opacities", 5e-2),
             ("sh0", 2.5e-3), ("shN", 2.5e-3 / 20))


def adam_hyperparameters(lr: float, batch_size: int, world_size: int) -> Tuple[float, float, Tuple[float, float]]:
    """The batch-size scaling rule of gsplat_trainer.py:266-278 -> (lr, eps, betas)."""
    BS = batch_size * world_size
    return lr * math.sqrt(BS), 1e-15 / math.sqrt(BS), (1 - BS * (1 - 0.9), 1 - BS * (1 - 0.999))


def create_splats_with_optimizers(
    points: Optional[Tensor] = None, rgbs: Optional[Tensor] = None, init_type: str = "random",
    init_num_pts: int = 100_000, init_extent: float = 3.0, init_opacity: float = 0.1,
    init_scale: float = 1.0, scene_scale: float = 1.0, sh_degree: int = 3, sparse_grad: bool = False,
    visible_adam: bool = False, batch_size: int = 1, device: str = "cuda", world_rank: int = 0,
    world_size: int = 1, shard_gaussians: bool = False, shN_init_std: float = 0.0,
) -> Tuple[torch.nn.ParameterDict, Dict[str, torch.optim.Optimizer]]:
    """Restates gsplat_trainer.py:204-281 (init_type "sfm" takes `points`/`rgbs` in place of the parser)."""
    if init_type == "sfm":
        assert points is not None and rgbs is not None, "sfm init needs points [N,3] and rgbs [N,3] in 0..1"
        points, rgbs = points.float(), rgbs.float()
    elif init_type == "random":
        points = init_extent * scene_scale * (torch.rand((init_num_pts, 3)) * 2 - 1)
        rgbs = torch.rand((init_num_pts, 3))
    else:
        raise ValueError("Please specify a correct init_type: sfm or random")
    dist2_avg = (knn(points, 4)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg)
    scales = torch.log(dist_avg * init_scale).unsqueeze(-1).repeat(1, 3)
    if shard_gaussians:
        points, rgbs, scales = (t[world_rank::world_size] for t in (points, rgbs, scales))
    N = points.shape[0]
    quats = torch.rand((N, 4))
    opacities = torch.logit(torch.full((N,), init_opacity))
    colors = torch.zeros((N, (sh_degree + 1) ** 2, 3))
    colors[:, 0, :] = rgb_to_sh(rgbs)
    if shN_init_std > 0:
        g = torch.Generator().manual_seed(43)
        colors[:, 1:, :] = torch.randn(colors[:, 1:, :].shape, generator=g) * shN_init_std
    values = {"means": points, "scales": scales, "quats": quats, "opacities": opacities,
              "sh0": colors[:, :1, :].contiguous(), "shN": colors[:, 1:, :].contiguous()}
    lrs = dict(PARAM_LRS)
    lrs["means"] = lrs["means"] * scene_scale
    splats = torch.nn.ParameterDict({n: torch.nn.Parameter(v) for n, v in values.items()}).to(device)
    # gsplat_trainer.py:267-272: SparseAdam for row-sparse gradients (packed mode), SelectiveAdam, else Adam
    opt_cls = torch.optim.SparseAdam if sparse_grad else (SelectiveAdam if visible_adam else FusedAdam)
    optimizers = {}
    for name in values:
        lr, eps, betas = adam_hyperparameters(lrs[name], batch_size, world_size)
        optimizers[name] = opt_cls([{"params": splats[name], "lr": lr, "name": name}], eps=eps, betas=betas)
    return splats, optimizers


def disparity_loss(depths: Tensor, points: Tensor, depths_gt: Tensor, width: int, height: int) -> Tensor:
    """gsplat_trainer.py:629-644 without the scene scale: the rendered expected depth [B,H,W,1] sampled bilinearly at the
    SfM points [B,M,2] (pixel coordinates, align_corners=True as there), disparity 1/d where d > 0 else 0, mean
    |disp - 1/depths_gt| over the B x M points."""
    grid = torch.stack([points[:, :, 0] / (width - 1) * 2 - 1, points[:, :, 1] / (height - 1) * 2 - 1], dim=-1).unsqueeze(2)
    d = torch.nn.functional.grid_sample(depths.permute(0, 3, 1, 2), grid, align_corners=True).squeeze(3).squeeze(1)
    # (the reference writes where(d > 0, 1 / d, 0): same values, but its gradient is 0 * inf = NaN wherever a point lies over an
    # empty pixel, d = 0 exactly; the inner where keeps the value and gives those points the gradient 0 they should have)
    pos = d > 0.0
    disp = torch.where(pos, 1.0 / torch.where(pos, d, torch.ones_like(d)), torch.zeros_like(d))
    return torch.nn.functional.l1_loss(disp, 1.0 / depths_gt)


class Runner:
    """Engine for training (hot path only).  `views` replaces the reference's dataset/parser:
    a list of dicts {"K":[3,3], "camtoworld":[4,4], "image":[H,W,3] in 0..255} like
    `Dataset.__getitem__` (utils/datasets/opensfm.py:341-389)."""

    # `splats`, `optimizers`, `strategy_state`: with Config.device_refine the model lives in the engine's capacity-sized
    # buffers and its size on the device; these torch-side handles are re-pointed at the live rows lazily, on access
    # (one device->host read of N after a refinement, none otherwise) -- the training step itself never touches them.
    def _fresh(self):
        eng = self.__dict__.get("_engine")
        if eng is not None and getattr(eng, "before_param_access", None) is not None:
            eng.before_param_access()            # (replicas: parameter all-gathers still in flight)
        if eng is not None and getattr(eng, "_host_stale", False):
            eng.sync_host()

    @property
    def splats(self):
        self._fresh()
        return self._splats

    @splats.setter
    def splats(self, v):
        self._splats = v

    @property
    def optimizers(self):
        self._fresh()
        return self._optimizers

    @optimizers.setter
    def optimizers(self, v):
        self._optimizers = v

    @property
    def strategy_state(self):
        self._fresh()
        return self._strategy_state

    @strategy_state.setter
    def strategy_state(self, v):
        self._strategy_state = v

    def __init__(self, local_rank: int, world_rank: int, world_size: int, cfg: Config,
                 views: Optional[List[Dict[str, Tensor]]] = None, scene_scale: float = 1.0,
                 points: Optional[Tensor] = None, rgbs: Optional[Tensor] = None):
        # reference: set_random_seed(42 + local_rank) (:290); here Gaussians are replicated, so every
        # rank must draw the same stream
        set_random_seed(42)
        self.cfg = cfg
        self.world_rank, self.local_rank, self.world_size = world_rank, local_rank, world_size
        self.device = f"cuda:{local_rank}"
        for name in ("app_opt", "use_bilateral_grid", "compression"):
            if getattr(cfg, name):
                raise NotImplementedError(f"Config.{name}: the appearance / bilateral-grid / compression branches of the "
                                          "reference trainer are outside the rasterisation path this package replaces")
        # Runner(local_rank, world_rank, world_size, cfg) on a data directory, as the reference's constructor (:308-324)
        # and app/gsplat_manager.py:49 call it: parse cfg.data_dir, split train / val, scene scale and SfM points
        import os
        if views is None and points is None and os.path.isfile(os.path.join(cfg.data_dir, "reconstruction.json")):
            self.parser, self.trainset, points, rgbs = self._load_data_dir(cfg)
            from .datasets import Dataset
            self.valset = Dataset(self.parser, split="val")
            self.allset = Dataset(self.parser, split="all")
            views, scene_scale = self.trainset, self.parser.scene_scale
        if cfg.camera_model is None:
            cfg.camera_model = "pinhole"
        self.views = views if views is not None else []
        self.scene_scale = scene_scale * 1.1 * cfg.global_scale          # gsplat_trainer.py:322
        self.sharded = (world_size > 1 and cfg.fused and cfg.dp_mode == "gaussian_sharded" and cfg.batch_size == 1
                        and sdist.is_initialized())
        assert cfg.dp_mode in ("gaussian_sharded", "allreduce"), cfg.dp_mode
        # Config.camera_model may differ between ranks (BASELINE.json configs[4]: even ranks perspective, odd ranks
        # fisheye).  A gaussian_sharded step projects the own shard into the cameras of ALL ranks: learn their models.
        self.camera_models_all = [cfg.camera_model]
        if self.sharded:
            import torch.distributed as dist
            self.camera_models_all = [None] * world_size
            dist.all_gather_object(self.camera_models_all, cfg.camera_model)
        self.splats, self.optimizers = create_splats_with_optimizers(
            points, rgbs, init_type=cfg.init_type, init_num_pts=cfg.init_num_pts, init_extent=cfg.init_extent,
            init_opacity=cfg.init_opa, init_scale=cfg.init_scale, scene_scale=self.scene_scale,
            sh_degree=cfg.sh_degree, sparse_grad=cfg.sparse_grad, visible_adam=cfg.visible_adam,
            batch_size=cfg.batch_size, device=self.device, world_rank=world_rank, world_size=world_size,
            shard_gaussians=self.sharded, shN_init_std=cfg.shN_init_std)
        self.cfg.strategy.check_sanity(self.splats, self.optimizers)
        if isinstance(self.cfg.strategy, MCMCStrategy):
            self.strategy_state = self.cfg.strategy.initialize_state()           # gsplat_trainer.py:351-352
        else:
            self.strategy_state = self.cfg.strategy.initialize_state(scene_scale=self.scene_scale)
        # pose refinement / perturbation (:363-381): one SE(3) delta per training view
        self.pose_optimizers: List[torch.optim.Optimizer] = []
        n_views_total = max(len(self.views), 1)
        if cfg.pose_opt:
            from .pose import CameraOptModule
            self.pose_adjust = CameraOptModule(n_views_total).to(self.device)
            self.pose_adjust.zero_init()
            self.pose_optimizers = [torch.optim.Adam(self.pose_adjust.parameters(),
                                                     lr=cfg.pose_opt_lr * math.sqrt(cfg.batch_size),
                                                     weight_decay=cfg.pose_opt_reg)]
            self.pose_lr0 = self.pose_optimizers[0].param_groups[0]["lr"]
        if cfg.pose_noise > 0.0:
            from .pose import CameraOptModule
            self.pose_perturb = CameraOptModule(n_views_total).to(self.device)
            self.pose_perturb.random_init(cfg.pose_noise)
        self.means_lr0 = self.optimizers["means"].param_groups[0]["lr"]
        self.lr_gamma = 0.01 ** (1.0 / cfg.max_steps)                     # ExponentialLR, :512-516
        self.step = 0
        self.stop_training = False
        self._workspace: dict = {}
        self._reducer = sdist.GradientReducer()
        self._split_gen = torch.Generator(device=self.device)
        self._split_gen.manual_seed(1234)                                 # same on every rank
        self.last_info: Optional[dict] = None
        self._void_seen = 0
        self._replaying = False          # Runner.train is training a void iteration's view again (no second refinement)
        from collections import deque
        self._dp_hist: deque = deque(maxlen=2)   # [pinned flag, event, looked at] of the last two replicated iterations
        self._dp_pinned, self._dp_slot = None, 0
        self._no_refine_before = 0       # step labels below this were seen before void iterations were taken back

    # ------------------------------------------------------------------------------ :308-324
    @staticmethod
    def _load_data_dir(cfg: Config):
        """(parser, trainset, points, rgbs) of cfg.data_dir; resolves / checks Config.camera_model against the shots."""
        from .datasets import Dataset, Parser
        parser = Parser(data_dir=cfg.data_dir, factor=cfg.data_factor, normalize=cfg.normalize_world_space,
                        test_every=cfg.test_every)
        # the reference renders every training view with Config.camera_model (:89, default "spherical": its data sets
        # are 360-degree captures); here the shots' own projection type has to agree with it
        types = {parser.camtype_dict[c] for c in parser.camera_ids}
        if len(types) != 1:
            raise NotImplementedError(f"camera types {sorted(types)} in one data set: train the perspective and the "
                                      "spherical shots as separate runs (one camera model per Runner)")
        want = {"perspective": ("pinhole", "fisheye", "ortho"), "spherical": ("spherical",)}[types.pop()]
        if cfg.camera_model is None:
            cfg.camera_model = want[0]
        if cfg.camera_model not in want:
            raise ValueError(f"Config.camera_model = {cfg.camera_model!r}, but the shots of {cfg.data_dir!r} need one of "
                             f"{want} (the reference's default is 'spherical', gsplat_trainer.py:89)")
        trainset = Dataset(parser, split="train", patch_size=cfg.patch_size, load_depths=cfg.depth_loss)
        points = rgbs = None
        if cfg.init_type == "sfm":
            points = torch.from_numpy(parser.points).float()
            rgbs = torch.from_numpy(parser.points_rgb / 255.0).float()
        return parser, trainset, points, rgbs

    @classmethod
    def from_data_dir(cls, local_rank: int, world_rank: int, world_size: int, cfg: Config) -> "Runner":
        """The reference's constructor path (:308-324) under an explicit name: parse cfg.data_dir, split train/val, take
        scene_scale and (init_type "sfm") the point cloud from the parser.  `Runner(local_rank, world_rank, world_size,
        cfg)` does the same whenever cfg.data_dir holds a reconstruction.json."""
        import os
        if not os.path.isfile(os.path.join(cfg.data_dir, "reconstruction.json")):
            raise FileNotFoundError(f"{cfg.data_dir!r} holds no reconstruction.json")
        return cls(local_rank, world_rank, world_size, cfg)

    def _result_dirs(self):
        import os
        d = self.cfg.result_dir.rstrip("/")
        out = {k: f"{d}/{k}" for k in ("ckpts", "stats", "renders", "videos")}
        for v in out.values():
            os.makedirs(v, exist_ok=True)
        return out

    # ------------------------------------------------------------------------------ :682-703
    def save_checkpoint(self, step: Optional[int] = None) -> str:
        """{result_dir}/ckpts/ckpt_{step}_rank{world_rank}.pt holding {"step", "splats"}; readable by the
        reference's `main` (:950-957) and by `load_checkpoints`."""
        step = self.step - 1 if step is None else step
        # The reference's loader (and load_checkpoints below) CONCATENATES the rank files of a run (:950-957), which is
        # right for Gaussian shards only.  Replicated runs (operator path, dp_mode="allreduce") hold every Gaussian on
        # every rank: rank 0 alone writes, so that loading "all files of the run" never duplicates the scene.
        if self.world_size > 1 and not self.sharded and self.world_rank != 0:
            return ""
        path = f"{self._result_dirs()['ckpts']}/ckpt_{step}_rank{self.world_rank}.pt"
        # live rows only: with a device-resident model the ParameterDict entries are views [:n] of capacity-sized buffers,
        # and torch.save serialises the whole storage behind a view (a 100k-Gaussian model became a 250 MB file)
        data = {"step": step, "splats": {k: v.detach().clone() for k, v in self.splats.state_dict().items()}}
        # (extension, ignored by the reference's loader :950-957) the Adam moments, so that a resumed run continues with the
        # optimiser state it had instead of empty moments under late-step bias corrections (a burst of ~10x steps)
        opt_state = {}
        for k, opt in self.optimizers.items():
            st = opt.state.get(self.splats[k], {})
            if "exp_avg" in st and not st["exp_avg"].is_sparse:
                opt_state[k] = {"exp_avg": st["exp_avg"].detach().clone(), "exp_avg_sq": st["exp_avg_sq"].detach().clone()}
        if len(opt_state) == len(self.optimizers):
            data["adam_moments"] = opt_state
        if self.cfg.pose_opt:
            data["pose_adjust"] = self.pose_adjust.state_dict()                # :693-697
        torch.save(data, path)
        return path

    def load_checkpoints(self, files: List[str]) -> int:
        """Concatenate the per-rank shards of a run (:950-957) into this runner's splats; optimiser state
        is rebuilt empty, as the reference's evaluation-only path leaves it.  Returns the stored step."""
        ckpts = [torch.load(f, map_location=self.device, weights_only=True) for f in files]
        if not ckpts:
            raise ValueError("no checkpoint files given")
        keys = set(self.splats.keys())
        for c in ckpts:
            if set(c["splats"].keys()) != keys:
                raise KeyError(f"checkpoint holds {sorted(c['splats'].keys())}, runner expects {sorted(keys)}")
        for k in self.splats.keys():
            new = torch.cat([c["splats"][k] for c in ckpts]).to(self.device).contiguous()
            old = self.splats[k]
            fresh = torch.nn.Parameter(new, requires_grad=True)
            self.splats[k] = fresh
            opt = self.optimizers[k]
            opt.state.pop(old, None)
            opt.param_groups[0]["params"] = [fresh]
        self._engine = None
        self.step = int(ckpts[0]["step"]) + 1
        # resume where the schedule stood: ExponentialLR of the means (:512-516, :741-742), a fresh strategy state of
        # the new length (the old statistics belong to another Gaussian set), Adam bias corrections at `step`
        self.optimizers["means"].param_groups[0]["lr"] = self.means_lr0 * self.lr_gamma ** self.step
        if isinstance(self.cfg.strategy, MCMCStrategy):
            self.strategy_state = self.cfg.strategy.initialize_state()
        else:
            self.strategy_state = self.cfg.strategy.initialize_state(scene_scale=self.scene_scale)
        for k, opt in self.optimizers.items():
            if isinstance(opt, torch.optim.SparseAdam):
                continue                      # lazily initialised per row by torch; starts over
            prm = self.splats[k]
            if all("adam_moments" in c and k in c["adam_moments"] for c in ckpts):      # written by save_checkpoint above
                m = torch.cat([c["adam_moments"][k]["exp_avg"] for c in ckpts]).to(self.device).contiguous()
                v = torch.cat([c["adam_moments"][k]["exp_avg_sq"] for c in ckpts]).to(self.device).contiguous()
            else:
                # a checkpoint in the reference's format holds no optimiser state.  (Known consequence: bias corrections of
                # step `self.step` on empty moments take steps of ~lr / sqrt(1 - beta2^k) for the first k ~ 1000 iterations
                # after the resume; the reference itself only loads checkpoints to evaluate, :950-966.)
                m, v = torch.zeros_like(prm), torch.zeros_like(prm)
            opt.state[prm] = {"step": torch.tensor(float(self.step)), "exp_avg": m, "exp_avg_sq": v}
        return int(ckpts[0]["step"])

    # ------------------------------------------------------------------------------ :779-838
    @torch.no_grad()
    def eval(self, step: int, stage: str = "val", dataset=None, save_images: bool = True) -> dict:
        """PSNR / SSIM of the val split (data_range 1, 11x11 sigma-1.5 window over the valid region --
        the torchmetrics definitions the reference instantiates at :419-433).  LPIPS is not computed:
        its network weights are a download.  Writes {stage}_step{step:04d}.json and the side-by-side
        canvases like the reference; returns the stats dict."""
        import json
        import time
        from .losses import fused_ssim
        cfg, dev = self.cfg, self.device
        dataset = dataset if dataset is not None else getattr(self, "valset", None)
        if dataset is None or len(dataset) == 0:
            raise ValueError("Runner.eval needs a validation set (Runner.from_data_dir or dataset=...)")
        dirs = self._result_dirs()
        psnr, ssim, elapsed = [], [], 0.0
        for i in range(len(dataset)):
            data = dataset[i]
            c2w = data["camtoworld"][None].to(dev)
            Ks = data["K"][None].to(dev)
            pixels = data["image"][None].to(dev) / 255.0
            h, w = pixels.shape[1:3]
            torch.cuda.synchronize()
            tic = time.time()
            colors, _, _ = self.rasterize_splats(camtoworlds=c2w, Ks=Ks, width=w, height=h, sh_degree=cfg.sh_degree,
                                                 near_plane=cfg.near_plane, far_plane=cfg.far_plane)
            torch.cuda.synchronize()
            elapsed += time.time() - tic
            colors = torch.clamp(colors[..., :3], 0.0, 1.0)
            if self.world_rank == 0:
                if save_images:
                    self._write_png(f"{dirs['renders']}/{stage}_step{step}_{i:04d}.png", torch.cat([pixels, colors], dim=2)[0])
                mse = torch.mean((colors - pixels) ** 2)
                psnr.append(10.0 * torch.log10(1.0 / mse))
                ssim.append(fused_ssim(colors.permute(0, 3, 1, 2).contiguous(), pixels.permute(0, 3, 1, 2).contiguous(),
                                       padding="valid", train=False))
        stats = {}
        if self.world_rank == 0:
            stats = {"psnr": torch.stack(psnr).mean().item(), "ssim": torch.stack(ssim).mean().item(),
                     "ellipse_time": elapsed / len(dataset), "num_GS": len(self.splats["means"])}
            with open(f"{dirs['stats']}/{stage}_step{step:04d}.json", "w") as f:
                json.dump(stats, f)
        return stats

    @staticmethod
    def _write_png(path: str, canvas01: Tensor) -> None:
        from PIL import Image as PILImage
        PILImage.fromarray((canvas01.clamp(0, 1).cpu().numpy() * 255).astype("uint8")).save(path)

    # ------------------------------------------------------------------------------ :841-901
    def trajectory(self, camtoworlds=None):
        """[n,4,4] float64 camera-to-world path of cfg.render_traj_path through the parser's cameras
        (the reference drops the first and last five, :846)."""
        import numpy as np
        from .datasets import generate_ellipse_path_z, generate_interpolated_path, generate_spiral_path
        cfg = self.cfg
        if camtoworlds is None:
            camtoworlds = self.parser.camtoworlds[5:-5]
        c = np.asarray(camtoworlds, dtype=np.float64)
        if len(c) < 2:
            raise ValueError("a trajectory needs at least two key cameras (the reference trims 5 from each end)")
        if cfg.render_traj_path == "interp":
            path = generate_interpolated_path(c, 1)
        elif cfg.render_traj_path == "ellipse":
            path = generate_ellipse_path_z(c, height=c[:, 2, 3].mean())
        elif cfg.render_traj_path == "spiral":
            parser = getattr(self, "parser", None)
            if parser is None or not hasattr(parser, "bounds"):
                raise ValueError("spiral paths need parser.bounds / parser.extconf, which the OpenSfM parser "
                                 "does not provide (the reference fails the same way, :856-860)")
            path = generate_spiral_path(c, bounds=parser.bounds * self.scene_scale,
                                        spiral_scale_r=parser.extconf["spiral_radius_scale"])
        else:
            raise ValueError(f"Render trajectory type not supported: {cfg.render_traj_path}")
        bottom = np.tile(np.array([[[0.0, 0.0, 0.0, 1.0]]]), (len(path), 1, 1))
        return np.concatenate([path[:, :3, :], bottom], axis=1)

    @torch.no_grad()
    def render_traj(self, step: int, camtoworlds=None, K: Optional[Tensor] = None,
                    width: Optional[int] = None, height: Optional[int] = None, save: bool = True):
        """Render the trajectory as [n, H, 2W, 3] uint8 canvases (colour | min-max normalised expected depth).
        Frames go to {result_dir}/videos/traj_{step}/NNNN.png -- the mp4 muxing of the reference is left to
        the caller (no encoder in the image)."""
        cfg, dev = self.cfg, self.device
        path = torch.from_numpy(self.trajectory(camtoworlds)).float().to(dev)
        if K is None:
            K = torch.from_numpy(list(self.parser.Ks_dict.values())[0]).float()
            width, height = list(self.parser.imsize_dict.values())[0]
        K = K.float().to(dev)
        frames = []
        for i in range(len(path)):
            renders, _, _ = self.rasterize_splats(camtoworlds=path[i:i + 1], Ks=K[None], width=width, height=height,
                                                  sh_degree=cfg.sh_degree, near_plane=cfg.near_plane,
                                                  far_plane=cfg.far_plane, render_mode="RGB+ED")
            colors = torch.clamp(renders[..., 0:3], 0.0, 1.0)
            depths = renders[..., 3:4]
            depths = (depths - depths.min()) / (depths.max() - depths.min())
            canvas = torch.cat([colors, depths.repeat(1, 1, 1, 3)], dim=2)[0]
            frames.append((canvas * 255).to(torch.uint8))
        frames = torch.stack(frames).cpu().numpy()
        if save and self.world_rank == 0:
            import os
            from PIL import Image as PILImage
            d = f"{self._result_dirs()['videos']}/traj_{step}"
            os.makedirs(d, exist_ok=True)
            for i, fr in enumerate(frames):
                PILImage.fromarray(fr).save(f"{d}/{i:04d}.png")
        return frames

    # ------------------------------------------------------------------------------ :916-944
    @torch.no_grad()
    def _viewer_render_fn(self, camera_state, img_wh: Tuple[int, int], camera_model: Optional[str] = None):
        """One viewer frame: `camera_state` offers `.c2w` [4,4] and `.get_K(img_wh)` [3,3] (nerfview's
        CameraState), or is a (c2w, K) pair.  Returns [H,W,3] float32 numpy in 0..1."""
        W, H = img_wh
        if isinstance(camera_state, (tuple, list)):
            c2w, K = camera_state
        else:
            c2w, K = camera_state.c2w, camera_state.get_K(img_wh)
        c2w = torch.as_tensor(c2w).float().to(self.device)
        K = torch.as_tensor(K).float().to(self.device)
        colors, _, _ = self.rasterize_splats(camtoworlds=c2w[None], Ks=K[None], width=W, height=H,
                                             sh_degree=self.cfg.sh_degree, radius_clip=3.0, camera_model=camera_model)
        return colors[0, ..., :3].cpu().numpy()

    @torch.no_grad()
    def full_splats(self) -> Dict[str, Tensor]:
        """All shards of a gaussian_sharded run concatenated in rank order (COLLECTIVE: every rank calls it);
        the eval / viewer renders need every Gaussian on the rendering rank."""
        import torch.distributed as dist
        n_loc = torch.tensor([len(self.splats["means"])], device=self.device)
        sizes = [torch.zeros_like(n_loc) for _ in range(self.world_size)]
        dist.all_gather(sizes, n_loc)
        sizes = [int(t.item()) for t in sizes]
        cap = max(sizes)
        out = {}
        for k, v in self.splats.items():
            buf = torch.zeros((cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device)
            buf[:v.shape[0]] = v.detach()
            parts = [torch.empty_like(buf) for _ in range(self.world_size)]
            dist.all_gather(parts, buf)
            out[k] = torch.cat([p[:m] for p, m in zip(parts, sizes)])
        return out

    # ------------------------------------------------------------------------------ :446-497
    def rasterize_splats(self, camtoworlds: Tensor, Ks: Tensor, width: int, height: int,
                         masks: Optional[Tensor] = None, camera_model: Optional[str] = None,
                         **kwargs) -> Tuple[Tensor, Tensor, Dict]:
        if getattr(self, "_engine", None) is not None and not kwargs.pop("_locked", False):
            # another thread (the reference's GUI renders from the Qt thread, app/gsplat_manager.py:185) must not put
            # work on the default stream while the training thread captures a hipGraph
            from .engine import CAPTURE_LOCK
            with CAPTURE_LOCK:
                return self.rasterize_splats(camtoworlds, Ks, width, height, masks=masks, camera_model=camera_model,
                                             _locked=True, **kwargs)
        kwargs.pop("_locked", None)
        sp = self.full_splats() if self.sharded else self.splats
        if camera_model is None:
            camera_model = self.cfg.camera_model
        kwargs.pop("image_ids", None)
        rasterize_mode = "antialiased" if self.cfg.antialiased else "classic"
        # :456-494 -- exp / sigmoid / cat / inv and the rasterization() call.  This runner holds the raw parameters, so the
        # activations travel INTO the call (rendering.rasterization_from_parameters: same expression, evaluated inside the
        # kernels when the call has the common shape, composed exactly as the reference composes it otherwise)
        common = dict(packed=self.cfg.packed,
                      absgrad=(self.cfg.strategy.absgrad if isinstance(self.cfg.strategy, DefaultStrategy) else False),
                      sparse_grad=self.cfg.sparse_grad, rasterize_mode=rasterize_mode, distributed=False,
                      camera_model=camera_model, isect_capacity=self.cfg.isect_capacity, workspace=self._workspace)
        viewmats = camera_inverse(camtoworlds)
        if self.cfg.raw_params_call:
            render_colors, render_alphas, info = rasterization_from_parameters(
                sp["means"], sp["quats"], sp["scales"], sp["opacities"], sp["sh0"], sp["shN"], viewmats, Ks, width, height,
                **common, **kwargs)
        else:
            render_colors, render_alphas, info = rasterization(
                means=sp["means"], quats=sp["quats"], scales=torch.exp(sp["scales"]), opacities=torch.sigmoid(sp["opacities"]),
                colors=torch.cat([sp["sh0"], sp["shN"]], 1), viewmats=viewmats, Ks=Ks, width=width, height=height,
                **common, **kwargs)
        if masks is not None:
            render_colors[~masks] = 0
        return render_colors, render_alphas, info

    # ------------------------------------------------------------------------------ :551-763
    # ------------------------------------------------------------------------------ fused fast path
    def _fused_ok(self, masks) -> bool:
        c = self.cfg
        if not (c.fused and masks is None and not c.random_bkgd and not c.visible_adam and not c.packed
                and not c.pose_opt and c.pose_noise <= 0.0 and not c.depth_loss):
            return False
        if isinstance(c.strategy, DefaultStrategy):
            return c.strategy.refine_scale2d_stop_iter == 0
        return isinstance(c.strategy, MCMCStrategy)

    def _train_step_fused(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor, image_ids: Optional[Tensor] = None) -> Tensor:
        from .engine import FusedEngine
        eng = getattr(self, "_engine", None)
        if self.world_size > 1 and eng is not None:
            self._dp_check_void(eng)             # may take void iterations (and their step labels) back
        cfg, step, s = self.cfg, self.step, self.cfg.strategy
        B, H, W = pixels.shape[0], pixels.shape[1], pixels.shape[2]
        if eng is None or (eng.C, eng.H, eng.W) != (B, H, W):
            n_floats = sum(int(v.numel()) for v in self.splats.values())
            self._dp_chunks = int(cfg.dp_chunks) if cfg.dp_chunks > 0 else (4 if 4 * n_floats >= (64 << 20) else 1)
            # DefaultStrategy on the device (single GPU, and replicated data parallelism: every rank runs the same
            # compaction on the all-reduced statistics and the gathered moments; the optimiser step is sharded by ROW
            # pieces of the capacity-sized tensors -- a flat piece layout would move with N)
            mcmc = isinstance(s, MCMCStrategy)
            dev_refine = cfg.device_refine and (mcmc or (isinstance(s, DefaultStrategy) and s.refine_scale2d_stop_iter == 0))
            capacity = cfg.max_gaussians
            if mcmc and dev_refine:      # MCMCStrategy grows in place up to cap_max: the ONE model set holds that from the start
                n0 = len(self._splats["means"])      # (708 B per row at SH degree 3: the default cap_max of 1M is 0.7 GB)
                capacity = max(capacity or max(2 * n0, 1 << 20), int(s.cap_max), n0)
            eng = self._engine = FusedEngine(
                self.splats, self.optimizers, W, H, B, sh_degree=0, camera_model=cfg.camera_model,
                near_plane=cfg.near_plane, far_plane=cfg.far_plane, antialiased=cfg.antialiased,
                absgrad=getattr(s, "absgrad", False),
                ssim_lambda=cfg.ssim_lambda, opacity_reg=cfg.opacity_reg, scale_reg=cfg.scale_reg,
                strategy_state=(self.strategy_state if isinstance(s, DefaultStrategy) else None),
                lr_gamma_means=self.lr_gamma,
                isect_capacity=cfg.isect_capacity, use_graph=True,
                attr_dtype=cfg.attr_dtype, tile_cull=cfg.tile_cull,
                binned=cfg.binned, bin_capacity=cfg.bin_capacity, bin_budget_bytes=int(cfg.bin_budget_gb * 2 ** 30),
                fuse_adam=cfg.fuse_adam, device_refine=dev_refine, capacity=capacity, loss_kernels=cfg.loss_kernels,
                model_sets=(1 if mcmc else 2),
                mcmc_noise=({"noise_lr": s.noise_lr, "seed": cfg.refine_seed} if mcmc else None),
                row_multiple=(self._dp_chunks * self.world_size * sdist.RowShardedAdam.ALIGN_ROWS if self.world_size > 1 else 1),
                flat_multiple=(self._dp_chunks * self.world_size * sdist.ShardedFlatAdam.ALIGN
                               if self.world_size > 1 and not dev_refine else 0))
            self._sadam = self._radam = None
            eng.steps_done = step
            eng._step_dev[0] = step
            if self.world_size > 1:
                # replicas must not diverge: "this rank's binning pass overflowed" is summed over the ranks by the gradient
                # reduce-scatter itself (distributed.RowShardedAdam.flags), every rank's Adam launches skip on the sum on the
                # device, and every rank's host takes the same iterations back one step late (_dp_check_void) -- the engine
                # must not act on what it saw locally
                eng.on_overflow = "defer"
                self._dp_hist.clear()
        eng.set_sh_degree(min(step // cfg.sh_degree_interval, cfg.sh_degree))
        # densification statistics are accumulated inside the backward kernel while refinement is active
        stats_on = isinstance(s, DefaultStrategy) and step < s.refine_stop_iter
        if stats_on != (eng.strategy_state is not None):
            eng.bind_strategy_state(self._strategy_state if stats_on else None)
        # which training image this is (the engine keeps a tile table per view, FusedEngine.set_views): the loader's image id when it
        # is on the host (no read-back for it), else the engine goes by the address of the target image
        vkey = None
        if image_ids is not None and not getattr(image_ids, "is_cuda", False):
            vkey = ("id",) + tuple(int(i) for i in torch.as_tensor(image_ids).reshape(-1).tolist())
        if self.world_size == 1:
            eng.set_views(camtoworlds, Ks, pixels, schedule=True, view_key=vkey)   # the step below always runs the optimiser
            eng.step()
        elif eng.device_refine:
            # replicated Gaussians, device-resident model: reduce-scatter / 1/world Adam / all-gather over ROW pieces of the
            # capacity-sized tensors (distributed.RowShardedAdam), the per-Gaussian backward cut into Config.dp_chunks row
            # chunks so that the reduce-scatter of chunk c runs on RCCL's stream under the kernel of chunk c + 1; no
            # collective besides those (the void flag rides in chunk 0).  N is the host's copy (sync_host below)
            eng.set_views(camtoworlds, Ks, pixels, schedule=False, view_key=vkey)
            if self._radam is None:
                self._radam = sdist.RowShardedAdam(n_chunks=self._dp_chunks)
            ra, n = self._radam, eng.n_host
            grads, params = eng.ws["grads"], eng.sets[eng.active]["p"]
            eng.fwd_bwd_head()
            ra.begin(n, eng.device, True)
            for c in range(ra.n_chunks):
                lo, hi = ra.chunk_range(n, c)
                eng.bwd_rows(lo, hi)                                       # (rows past the live count: the kernel clamps)
                ra.reduce_chunk(c, grads, eng.ws["ovf_f32"])
            # the parameter all-gathers stay in flight: the next iteration's staging (so_step_inputs, a target upload) runs under
            # their tail, the engine waits for them in front of the first launch that touches the parameters
            eng.before_param_access = ra.wait_gathers
            ra.finish(grads, params, eng.adam_on_rows, defer_gather_wait=True)
            self._dp_note_void(ra.void_flag())
            eng.refresh_attrs()              # float16 attribute rows (if any) follow the gathered masters
            eng._advance_host_counters()
        else:
            # replicated Gaussians, torch-level refinement: reduce-scatter of the FLAT gradient in chunks, Adam on this
            # rank's 1/world of every chunk as it lands, all-gather of the updated parameters (distributed.ShardedFlatAdam);
            # the void flags travel in a 64-byte-per-rank reduce-scatter of their own, issued with the first chunk
            eng.set_views(camtoworlds, Ks, pixels, schedule=False, view_key=vkey)
            eng.fwd_bwd()
            if self._sadam is None or self._sadam.total != eng.flat_total:
                self._sadam = sdist.ShardedFlatAdam(eng.flat_total, n_chunks=self._dp_chunks)
            self._sadam.step(eng.ws["grads_flat"], eng.ws["params_flat"], eng.adam_on_flat_range, void_src=eng.ws["ovf_f32"])
            self._dp_note_void(self._sadam.void_flag())
            eng.refresh_attrs()              # float16 attribute rows (if any) follow the gathered masters
            eng._advance_host_counters()
        if isinstance(s, MCMCStrategy):
            refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                          and not self._replaying and step >= self._no_refine_before)
            if eng.device_refine:
                # MCMCStrategy on the device (gsplat_trainer.py:753-761): relocation + addition in place on the capacity-sized
                # model, then this iteration's position noise -- no host read, no re-capture.  Replicas (world_size > 1)
                # draw the same counter-based samples; the row-sharded moments are gathered first because the row pieces
                # move with N, and N (which sizes the next steps' collectives) is read back once and compared.
                if refine_now:
                    if self.world_size > 1 and self._radam is not None:
                        act = eng.sets[eng.active]
                        self._radam.gather([act[q][k] for q in ("m", "v") for k in act[q]], eng.n_host)
                    eng.mcmc_refine(s, step, self._strategy_state["binoms"], seed=cfg.refine_seed)
                    if s.verbose:
                        rep = eng.refine_report()
                        print(f"Step {step}: Relocated {rep['n_dupli']} GSs. Added {rep['n_split']} GSs. Now having {rep['n_new']} GSs.")
                    if self.world_size > 1:
                        n_new = eng.sync_host()
                        eng.reprobe_capacity()
                        chk = torch.tensor([n_new, -n_new], dtype=torch.int32, device=eng.device)
                        sdist.all_reduce_max_(chk)
                        if int(chk[0]) != n_new or int(chk[1]) != -n_new:
                            raise RuntimeError(f"replicas diverged: this rank holds {n_new} Gaussians after the MCMC refinement of step "
                                               f"{step}, others between {-int(chk[1])} and {int(chk[0])}")
                if self.world_size > 1:
                    # replicas: the noise kernel reads the optimiser step from device memory (Philox counter, lr decay) -- the
                    # data-parallel steps run Adam with host-side schedules and never advance it -- and must skip on the flag
                    # SUMMED over the ranks, as the Adam launches did: a rank-local overflow word would let the other
                    # replicas add noise the overflowing one leaves out
                    eng._step_dev[0] = eng.steps_done
                    flag = self._radam.void_flag() if self._radam is not None else (self._sadam.void_flag() if self._sadam is not None else None)
                    eng.inject_noise(skip=flag)
                else:
                    eng.inject_noise()
            else:
                # torch-level strategy ops on the host-side handles.  Replicated data parallelism shards the Adam moments by
                # flat pieces (ShardedFlatAdam): every rank rewrites ALL of them, so the other owners' pieces come first
                if refine_now and self.world_size > 1 and self._sadam is not None:
                    self._sadam.gather_moments(eng.ws["m_flat"], eng.ws["v_flat"])
                # lr = the means learning rate after this step's scheduler.step() (gsplat_trainer.py:753-761)
                # (a replayed void iteration: a label outside the refinement window -- the noise is drawn, nothing is relocated)
                n_rel, n_new = s.step_post_backward(params=self.splats, optimizers=self.optimizers,
                                                    state=self.strategy_state,
                                                    step=(-1 if (self._replaying or step < self._no_refine_before) else step), info={},
                                                    lr=self.optimizers["means"].param_groups[0]["lr"],
                                                    generator=self._split_gen)
                if n_rel or n_new:
                    eng.rebuild()
            self.last_info = {"radii": eng.ws["radii"], "engine": eng,   # eng.stats() / eng.tile_lists(): counts and lists
                              "flatten_ids": eng.ws["flatten_ids"], "means2d": eng.ws["means2d"]}
            self.step += 1
            return eng.loss()[0]
        refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                      and step % s.reset_every >= s.pause_refine_after_reset and not self._replaying
                      and step >= self._no_refine_before)
        reset_now = (step < s.refine_stop_iter and step % s.reset_every == 0 and step > 0 and not self._replaying
                     and step >= self._no_refine_before)
        if eng.device_refine:
            self.refine_on_device(step, refine_now, reset_now)
        elif refine_now or reset_now:
            # torch-level refinement: the tensors of the ParameterDict are replaced one by one -- a render from another
            # thread (Runner.rasterize_splats holds the same lock) must not see half of them
            from .engine import CAPTURE_LOCK
            with CAPTURE_LOCK:
                if refine_now:
                    if self.world_size > 1:
                        sdist.all_reduce_strategy_state(self.strategy_state)
                        if self._sadam is not None:      # every rank rewrites ALL moments: fetch the other owners' pieces
                            self._sadam.gather_moments(eng.ws["m_flat"], eng.ws["v_flat"])
                    n_dupli, n_split = s._grow_gs(self.splats, self.optimizers, self.strategy_state, step, self._split_gen)
                    n_prune = s._prune_gs(self.splats, self.optimizers, self.strategy_state, step)
                    if s.verbose:
                        print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                              f"Now having {len(self.splats['means'])} GSs.")
                    self.strategy_state["grad2d"].zero_()
                    self.strategy_state["count"].zero_()
                if reset_now:
                    from .strategy import reset_opa
                    reset_opa(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                              value=s.prune_opa * 2.0)
                eng.rebuild()
        self.last_info = {"radii": eng.ws["radii"], "engine": eng,   # eng.stats() / eng.tile_lists(): counts and lists
                          "flatten_ids": eng.ws["flatten_ids"], "means2d": eng.ws["means2d"]}
        self.step += 1
        return eng.loss()[0]

    # ---- data-parallel replicas: void iterations (a view's binning pass overflowed on SOME rank)
    def _dp_note_void(self, flag: Optional[Tensor]) -> None:
        """Remember this iteration's void flag -- the sum over all ranks the reduce-scatter left in this rank's piece, the
        same number everywhere -- for the host: an asynchronous 4-byte copy into pinned memory and an event, no wait."""
        if flag is None:
            return
        if self._dp_pinned is None:
            self._dp_pinned = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(4)]
        slot = self._dp_pinned[self._dp_slot % len(self._dp_pinned)]
        self._dp_slot += 1
        slot.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._dp_hist.append([slot, ev, False])

    def _dp_check_void(self, eng) -> None:
        """One step late and without a device-wide wait (FusedEngine._check_previous for replicas): was the iteration before
        the last one void?  Every rank reads the same summed flag, so every rank takes the same iterations back (step counter,
        Adam bias correction, lr schedule) -- the Adam launches skipped them on the device already -- and a rank that saw
        its own bins overflow enlarges them.  The views of void iterations are not trained again here."""
        h = self._dp_hist
        if len(h) < 2 or h[-2][2]:
            return
        old, new = h[-2], h[-1]
        old[1].synchronize()
        old[2] = True
        if float(old[0][0]) == 0.0:
            return
        torch.cuda.synchronize()                 # rare path: the last iteration's flag is needed too
        void = 1 + (1 if float(new[0][0]) != 0.0 else 0)
        h.clear()
        seen, needed = eng.local_overflow_recent()
        eng.take_back(void, needed, grow=seen)
        # the step LABELS go back with the optimiser steps (SH-degree ramp, lr schedule and refinement calendar follow the
        # number of steps actually taken, as on one GPU); what the strategy did at the labels that are now repeated is not
        # done a second time
        self._no_refine_before = max(self._no_refine_before, self.step)
        self.step -= void

    def refine_on_device(self, step: int, refine_now: bool = True, reset_now: bool = False) -> None:
        """DefaultStrategy's refinement (and / or opacity reset) of `step` on the device-resident model of the fused engine:
        five launches, nothing read back, the next step replays the other set's graph.  In replicated data parallelism
        (a collective call: every rank, same step) the statistics of all ranks' views are summed and the row-sharded
        moments gathered first, every rank then runs the same compaction, and N -- which sizes the following steps'
        collectives -- is read back once and compared across ranks."""
        cfg, s, eng = self.cfg, self.cfg.strategy, self._engine
        assert eng.device_refine
        if refine_now:
            if self.world_size > 1:
                n = eng.n_host
                sdist.all_reduce_strategy_state({"grad2d": eng.dstats["grad2d"][:n], "count": eng.dstats["count"][:n]})
                if self._radam is not None:      # the compaction rewrites ALL moments: fetch the other owners' rows
                    act = eng.sets[eng.active]
                    self._radam.gather([act[q][k] for q in ("m", "v") for k in act[q]], n)
            eng.refine(s, step, self.scene_scale, seed=cfg.refine_seed)
            if s.verbose:
                rep = eng.refine_report()           # (synchronises; verbose runs only)
                print(f"Step {step}: {rep['n_dupli']} GSs duplicated, {rep['n_split']} GSs split, {rep['n_prune']} GSs pruned. "
                      f"Now having {rep['n_new']} GSs.")
        if reset_now:
            eng.reset_opacity(s.prune_opa * 2.0)
        if refine_now and self.world_size > 1:
            n_new = eng.sync_host()
            eng.reprobe_capacity()       # a void iteration cannot be taken back here: size the bins for the grown model
            chk = torch.tensor([n_new, -n_new], dtype=torch.int32, device=eng.device)
            sdist.all_reduce_max_(chk)
            if int(chk[0]) != n_new or int(chk[1]) != -n_new:
                raise RuntimeError(f"replicas diverged: this rank holds {n_new} Gaussians after the refinement of step {step}, "
                                   f"others between {-int(chk[1])} and {int(chk[0])}")

    def _train_step_sharded(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor) -> Tensor:
        """Gaussian-sharded step (splat_one_amd/sharded.py): camtoworlds / Ks hold the cameras of ALL ranks,
        pixels the own view.  Densification runs on the own shard with its own statistics, like the
        reference's per-rank strategy state."""
        from .sharded import ShardedEngine
        cfg, step, s = self.cfg, self.step, self.cfg.strategy
        H, W = pixels.shape[1], pixels.shape[2]
        eng = getattr(self, "_engine", None)
        if eng is None or (eng.H, eng.W) != (H, W):
            eng = self._engine = ShardedEngine(
                self.splats, self.optimizers, W, H, self.world_rank, self.world_size, sh_degree=0,
                camera_model=self.camera_models_all, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
                antialiased=cfg.antialiased, absgrad=getattr(s, "absgrad", False), ssim_lambda=cfg.ssim_lambda,
                opacity_reg=cfg.opacity_reg, scale_reg=cfg.scale_reg,
                strategy_state=(self.strategy_state if isinstance(s, DefaultStrategy) else None),
                lr_gamma_means=self.lr_gamma, isect_capacity=cfg.isect_capacity, attr_dtype=cfg.attr_dtype, tile_cull=cfg.tile_cull)
            eng.steps_done = step
            eng._step_dev[0] = step
        eng.set_sh_degree(min(step // cfg.sh_degree_interval, cfg.sh_degree))
        stats_on = isinstance(s, DefaultStrategy) and step < s.refine_stop_iter
        eng.strategy_state = self.strategy_state if stats_on else None
        eng.step(camtoworlds, Ks, pixels)
        changed = False
        if isinstance(s, MCMCStrategy):
            n_rel, n_new = s.step_post_backward(params=self.splats, optimizers=self.optimizers,
                                                state=self.strategy_state, step=step, info={},
                                                lr=self.optimizers["means"].param_groups[0]["lr"],
                                                generator=self._split_gen)
            # ranks refine on the same steps, so the collective inside rebuild() is matched even when only
            # some shards changed
            changed = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0)
        else:
            refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                          and step % s.reset_every >= s.pause_refine_after_reset)
            reset_now = step < s.refine_stop_iter and step % s.reset_every == 0 and step > 0
            if refine_now:
                s._grow_gs(self.splats, self.optimizers, self.strategy_state, step, self._split_gen)
                s._prune_gs(self.splats, self.optimizers, self.strategy_state, step)
                self.strategy_state["grad2d"].zero_()
                self.strategy_state["count"].zero_()
            if reset_now:
                from .strategy import reset_opa
                reset_opa(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                          value=s.prune_opa * 2.0)
            changed = refine_now or reset_now
        if changed:
            eng.rebuild()
        self.last_info = {"radii": eng.ws["radii_full"], "n_isects": eng.ws["counters"][2 * eng.M + 1:2 * eng.M + 2],
                          "flatten_ids": eng.ws["flatten_ids"], "means2d": eng.ws["means2d_full"]}
        self.step += 1
        return eng.loss()[0]

    def train_step(self, camtoworlds: Tensor, Ks: Tensor, pixels: Tensor, masks: Optional[Tensor] = None,
                   image_ids: Optional[Tensor] = None, points: Optional[Tensor] = None,
                   depths_gt: Optional[Tensor] = None) -> Tensor:
        """One iteration on an already-on-device batch (camtoworlds[B,4,4], Ks[B,3,3],
        pixels[B,H,W,3] in 0..1; in gaussian_sharded runs the cameras of ALL ranks and the own image).  Returns the loss tensor (no host sync; on the fused path it is a
        view of the engine's static loss buffer, valid until the next step -- clone it to keep it)."""
        if self.cfg.depth_loss and (points is None or depths_gt is None):
            # (the reference reads data["points"] / data["depths"] at :573-575 and fails on a batch that has none)
            raise ValueError("Config.depth_loss=True needs points [B,M,2] (pixel coordinates) and depths_gt [B,M] with every batch: "
                             "train_step(..., points=..., depths_gt=...)")
        if self.sharded:
            assert self._fused_ok(masks), "gaussian_sharded runs need the fused path (no masks / random background / depth loss)"
            assert camtoworlds.shape[0] == self.world_size and pixels.shape[0] == 1, \
                "gaussian_sharded: pass the cameras of all ranks [world,4,4] / [world,3,3] and the own image [1,H,W,3]"
            return self._train_step_sharded(camtoworlds, Ks, pixels)
        if self._fused_ok(masks):
            return self._train_step_fused(camtoworlds, Ks, pixels, image_ids)
        cfg, step = self.cfg, self.step
        height, width = pixels.shape[1:3]
        if cfg.pose_noise > 0.0 or cfg.pose_opt:                               # :579-582
            assert image_ids is not None, "pose_opt / pose_noise need image_ids (index of each view in the training set)"
            ids = image_ids.to(self.device).long()
            if cfg.pose_noise > 0.0:
                camtoworlds = self.pose_perturb(camtoworlds, ids)
            if cfg.pose_opt:
                camtoworlds = self.pose_adjust(camtoworlds, ids)
        sh_degree_to_use = min(step // cfg.sh_degree_interval, cfg.sh_degree)
        renders, alphas, info = self.rasterize_splats(
            camtoworlds=camtoworlds, Ks=Ks, width=width, height=height, sh_degree=sh_degree_to_use,
            near_plane=cfg.near_plane, far_plane=cfg.far_plane, render_mode="RGB+ED" if cfg.depth_loss else "RGB", masks=masks)
        colors = renders[..., 0:3]
        depths = renders[..., 3:4] if renders.shape[-1] == 4 else None
        if cfg.random_bkgd:
            bkgd = torch.rand(1, 3, device=colors.device)
            colors = colors + bkgd * (1.0 - alphas)
        cfg.strategy.step_pre_backward(params=self.splats, optimizers=self.optimizers,
                                       state=self.strategy_state, step=step, info=info)
        loss, _l1, _ssim = photometric_loss(colors, pixels, cfg.ssim_lambda)
        if cfg.depth_loss:                                                     # :629-645
            depthloss = disparity_loss(depths, points.to(self.device), depths_gt.to(self.device), width, height) * self.scene_scale
            loss = loss + depthloss * cfg.depth_lambda
            self.last_depthloss = depthloss.detach()
        if cfg.opacity_reg > 0.0:
            loss = loss + cfg.opacity_reg * torch.abs(torch.sigmoid(self.splats["opacities"])).mean()
        if cfg.scale_reg > 0.0:
            loss = loss + cfg.scale_reg * torch.abs(torch.exp(self.splats["scales"])).mean()
        loss.backward()
        # view-sharded data parallelism: one all-reduce of the flattened gradient SoA
        if self.world_size > 1:
            if cfg.sparse_grad:                     # the flat all-reduce takes dense gradients
                for prm in self.splats.values():
                    if prm.grad is not None and prm.grad.is_sparse:
                        prm.grad = prm.grad.to_dense()
            self._reducer.reduce(self.splats.values())
        # Replicated Gaussians (world_size > 1, not sharded): the all-reduced gradient carries contributions from the
        # views of ALL ranks, so the set of rows the optimiser may touch is the union of every rank's visibility -- a
        # rank-local mask would update a different row subset on every rank and the replicas would diverge.
        vis_union = None
        if self.world_size > 1 and (cfg.sparse_grad or cfg.visible_adam):
            if cfg.packed:
                vis_union = torch.zeros(len(self.splats["means"]), dtype=torch.uint8, device=self.device)
                vis_union[info["gaussian_ids"]] = 1
            else:
                vis_union = (info["radii"] > 0).any(0).to(torch.uint8)
            sdist.all_reduce_max_(vis_union)
        if cfg.sparse_grad:                                                    # :705-717
            assert cfg.packed, "Sparse gradients only work with packed mode."
            gaussian_ids = info["gaussian_ids"]
            if vis_union is not None:
                rows = torch.nonzero(vis_union, as_tuple=True)[0]
            else:
                rows = gaussian_ids if len(Ks) == 1 else torch.unique(gaussian_ids)    # one index per visible Gaussian
            for prm in self.splats.values():
                grad = prm.grad
                if grad is None or grad.is_sparse:
                    continue
                prm.grad = torch.sparse_coo_tensor(indices=rows[None], values=grad[rows], size=prm.size(),
                                                   is_coalesced=True)
        # optimisers (one fused launch) + zero_grad(set_to_none=True)
        vis = None
        if cfg.visible_adam and vis_union is not None:
            vis = vis_union.bool()
        elif cfg.visible_adam:                                                 # :719-724
            if cfg.packed:
                vis = torch.zeros_like(self.splats["opacities"], dtype=torch.bool)
                vis.scatter_(0, info["gaussian_ids"], True)
            else:
                vis = (info["radii"] > 0).any(0)
        if cfg.sparse_grad:
            for opt in self.optimizers.values():                               # torch.optim.SparseAdam
                opt.step()
                opt.zero_grad(set_to_none=True)
        else:
            step_all(self.optimizers.values(), set_to_none=True, visibility=vis)
        for opt in self.pose_optimizers:                                       # :732-734, scheduler :517-522
            if self.world_size > 1:
                for prm in self.pose_adjust.parameters():
                    sdist.all_reduce_mean_(prm.grad)
            opt.step()
            opt.zero_grad(set_to_none=True)
            opt.param_groups[0]["lr"] = self.pose_lr0 * self.lr_gamma ** (step + 1)
        # ExponentialLR on the means
        self.optimizers["means"].param_groups[0]["lr"] = self.means_lr0 * self.lr_gamma ** (step + 1)
        # densification
        s = cfg.strategy
        if isinstance(s, DefaultStrategy):
            refine_now = (step < s.refine_stop_iter and step > s.refine_start_iter and step % s.refine_every == 0
                          and step % s.reset_every >= s.pause_refine_after_reset)
            if refine_now and self.world_size > 1:
                # local statistics of this step are added inside step_post_backward; reduce after it
                s._update_state(self.splats, self.strategy_state, info, packed=cfg.packed)
                sdist.all_reduce_strategy_state(self.strategy_state)
                self._post_backward_refine_only(step, info)
            else:
                s.step_post_backward(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                                     step=step, info=info, packed=cfg.packed, generator=self._split_gen)
        elif isinstance(s, MCMCStrategy):
            s.step_post_backward(params=self.splats, optimizers=self.optimizers, state=self.strategy_state, step=step,
                                 info=info, lr=self.optimizers["means"].param_groups[0]["lr"],
                                 generator=self._split_gen)
        self.last_info = info
        self.step += 1
        return loss.detach()

    def _post_backward_refine_only(self, step: int, info: dict) -> None:
        s = self.cfg.strategy
        n_dupli, n_split = s._grow_gs(self.splats, self.optimizers, self.strategy_state, step, self._split_gen)
        n_prune = s._prune_gs(self.splats, self.optimizers, self.strategy_state, step)
        if s.verbose:
            print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                  f"Now having {len(self.splats['means'])} GSs.")
        self.strategy_state["grad2d"].zero_()
        self.strategy_state["count"].zero_()
        if s.refine_scale2d_stop_iter > 0 and self.strategy_state.get("radii") is not None:
            self.strategy_state["radii"].zero_()
        if step % s.reset_every == 0 and step > 0:
            from .strategy import reset_opa
            reset_opa(params=self.splats, optimizers=self.optimizers, state=self.strategy_state,
                      value=s.prune_opa * 2.0)

    def _epoch_orders(self) -> List[List[int]]:
        """View indices of one epoch for every rank.  Config.shuffle: the seed is drawn from the global torch generator the
        way `torch.utils.data.RandomSampler` draws its own (the reference's DataLoader(shuffle=True), gsplat_trainer.py:
        539-546), then `randperm` on a private generator -- with one rank the order IS the RandomSampler's order for the
        same global generator state (tests/test_strategy_host.py).  Every rank draws the same seed (all ranks seed 42 and
        consume the generator alike) and can therefore derive every rank's list."""
        n, world = len(self.views), self.world_size
        bases = [list(range(j, n, world)) or [0] for j in range(world)]
        if not self.cfg.shuffle:
            return bases
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        out = []
        for j, base in enumerate(bases):
            g = torch.Generator()
            g.manual_seed(seed + j)
            out.append([base[i] for i in torch.randperm(len(base), generator=g).tolist()])
        return out

    def train(self, max_steps: Optional[int] = None) -> None:
        """Loop over `views` (batch_size per step, rank-strided like a DistributedSampler).
        Decoded views stay resident in HBM after their first use (a 1080p float view is 25 MB; the
        288 GB of an MI355X hold thousands), so steady-state steps do no host->device copies; random
        patches (cfg.patch_size) are re-cut each time and bypass the cache.  Checkpoints and evaluation
        fire at cfg.save_steps / cfg.eval_steps when the runner was built from a data directory."""
        assert len(self.views) > 0, "Runner.train needs views"
        cfg = self.cfg
        n = max_steps if max_steps is not None else cfg.max_steps
        B = cfg.batch_size
        dev = self.device
        # one list of view indices per rank (rank j walks views j, j + world, ...); every rank can derive every rank's
        # list, so a gaussian_sharded step knows the other ranks' cameras without a collective
        orders = self._epoch_orders()
        order = orders[self.world_rank]
        cursor = 0
        resident: Dict[int, Tuple[Tensor, Tensor, Tensor]] = {}
        cache = cfg.patch_size is None
        has_data = hasattr(self, "parser")

        def fetch(i: int):
            if cache and i in resident:
                return resident[i]
            v = self.views[i]
            t = (v["camtoworld"].to(dev), v["K"].to(dev), v["image"].to(dev) / 255.0)
            if cfg.depth_loss:
                # :573-575 reads data["points"] / data["depths"]; a data set without SfM points per view (the OpenSfM one,
                # like the reference's: opensfm.py:384-387 gives placeholder depths and no points) cannot feed the term
                if "points" not in v or "depths" not in v or v["depths"].dim() != 1:
                    raise ValueError("Config.depth_loss=True: every view must carry 'points' [M,2] (pixel coordinates of its SfM "
                                     "points) and 'depths' [M]; this data set has none (the reference's run stops at "
                                     "gsplat_trainer.py:574 for the same reason)")
                t = t + (v["points"].to(dev).float(), v["depths"].to(dev).float())
            if cache:
                resident[i] = t
            return t

        def camera_of(i: int):
            """Camera of view i without decoding its image (the other ranks' cameras of a sharded step)."""
            if i in resident:
                return resident[i][0], resident[i][1]
            ps = getattr(self.views, "parser", None)
            if ps is not None:
                g = int(self.views.indices[i])
                return (torch.from_numpy(ps.camtoworlds[g]).float().to(dev),
                        torch.from_numpy(ps.Ks_dict[ps.camera_ids[g]]).float().to(dev))
            v = self.views[i]
            return v["camtoworld"].to(dev), v["K"].to(dev)

        if self.sharded:
            assert B == 1 and cfg.patch_size is None, "gaussian_sharded: one full view per rank per step"
        import json
        import time
        global_tic = time.time()
        if has_data and self.world_rank == 0:           # the run's configuration next to its results (:505-507)
            import yaml
            d = self._result_dirs()
            with open(f"{cfg.result_dir.rstrip('/')}/cfg.yml", "w") as f:
                yaml.safe_dump({k: (v if isinstance(v, (int, float, str, bool, list, type(None))) else repr(v))
                                for k, v in vars(cfg).items()}, f)
        from collections import deque
        recent: deque = deque(maxlen=4)          # the last batches: a void iteration (below) is repeated on ITS view
        done = 0
        while done < n:
            if self.stop_training:
                break
            # epoch over: a new permutation (the reference re-iterates its DataLoader, :562-566).  Every rank ends its epoch
            # on the same step -- after the shortest rank list -- so that all ranks draw the same seeds
            if cursor + B > min(len(o) for o in orders) and cursor > 0:
                orders = self._epoch_orders()
                order, cursor = orders[self.world_rank], 0
            picked = [order[(cursor + i) % len(order)] for i in range(B)]
            batch = [fetch(i) for i in picked]
            if self.sharded:
                # every rank derives the cameras of all ranks from the same sampling rule: no collective
                cams = []
                for j in range(self.world_size):
                    oj = orders[j]
                    cams.append(batch[0][:2] if j == self.world_rank else camera_of(oj[cursor % len(oj)]))
                c2w = torch.stack([c[0] for c in cams])
                Ks = torch.stack([c[1] for c in cams])
            else:
                c2w = torch.stack([b[0] for b in batch])
                Ks = torch.stack([b[1] for b in batch])
            cursor += B
            pixels = torch.stack([b[2] for b in batch])
            step = self.step
            ids = torch.tensor(picked)
            recent.append((c2w, Ks, pixels, ids))
            if cfg.depth_loss:
                self.train_step(c2w, Ks, pixels, image_ids=ids, points=torch.stack([b[3] for b in batch]),
                                depths_gt=torch.stack([b[4] for b in batch]))
            else:
                self.train_step(c2w, Ks, pixels, image_ids=ids)
            done += 1
            # Void iterations (single GPU, fused engine): a tile's bin overflowed, the optimiser skipped that iteration on
            # the device and the host learns of it one or two steps late (FusedEngine._check_previous).  The reference never
            # drops a view: the void iterations' OWN views are trained again now -- two neighbouring views swap places.
            eng = getattr(self, "_engine", None)
            void = getattr(eng, "void_steps", 0) - self._void_seen
            while void > 0 and not self.sharded and self.world_size == 1:
                self._void_seen += void
                todo = ([recent[-3]] if void == 1 and len(recent) >= 3 else list(recent)[-3:-1])[:void]
                # rewind by what IS replayed (at most two views are remembered), and do not run the strategy a
                # second time at a repeated label -- its refinement / opacity reset already happened when the label was
                # first seen (MCMC: a second relocate-and-add would grow N by 1.05 twice)
                self.step -= len(todo)
                self._replaying = True
                try:
                    for args_ in todo:
                        self.train_step(args_[0], args_[1], args_[2], image_ids=args_[3])
                finally:
                    self._replaying = False
                void = getattr(self._engine, "void_steps", 0) - self._void_seen
            if has_data:
                if step in [i - 1 for i in cfg.save_steps] or step == n - 1:
                    stats = {"mem": torch.cuda.max_memory_allocated() / 1024 ** 3, "ellipse_time": time.time() - global_tic,
                             "num_GS": len(self.splats["means"])}                       # :683-691
                    with open(f"{self._result_dirs()['stats']}/train_step{step:04d}_rank{self.world_rank}.json", "w") as f:
                        json.dump(stats, f)
                    self.save_checkpoint(step)
                if step in [i - 1 for i in cfg.eval_steps] and len(self.valset) > 0:
                    self.eval(step)
