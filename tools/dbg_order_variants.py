"""Which workgroup -> tile table suits c2?  The engine keeps one per view now (so_step_desc.tile_order_ready), so a table may be as
elaborate as it likes: this script trains the c2 scene on 8 views and times 400 iterations with the tables replaced by
  (a) the device-built one (256 length classes, global longest-first)          (b) an exact global sort by length
  (c) longest-first WITHIN each XCD's own tiles (the default order's runs of 8 neighbouring tiles per XCD stay on their XCD)
  (d) the default XCD-local order written out as a table (no longest-first at all)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.trainer import Config, Runner

dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
cfg.strategy.refine_start_iter = cfg.strategy.reset_every = 10 ** 9
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
ring = ring_cameras(8).to(dev); Ks = pinhole_K(W, H)[None].to(dev)
targets = [torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(100 + v)).to(dev) * 0.2 + 0.4 for v in range(8)]
def steps(n):
    for i in range(n):
        r.train_step(ring[i % 8:i % 8 + 1], Ks, targets[i % 8])
steps(64)
for o in r.optimizers.values():          # freeze the model: the same lists at every visit of a view from here on
    for g_ in o.param_groups:
        g_["lr"] = 0.0
r.means_lr0 = 0.0
eng = r._engine
torch.cuda.synchronize()
M = eng.M
keys = [("px", t.data_ptr()) for t in targets]
import ctypes
from splat_one_amd import _lib
_l = _lib.load()
_l.so_bin_counter_index.restype = ctypes.c_int64
_l.so_bin_counter_index.argtypes = [ctypes.c_int64, ctypes.c_int64]
where = torch.tensor([_l.so_bin_counter_index(t, M) for t in range(M)], device=dev)       # the count of tile t lives at counters[where[t]]
lengths = {}
for v in range(8):      # the list lengths of every view
    r.train_step(ring[v:v + 1], Ks, targets[v])
    torch.cuda.synchronize()
    lengths[v] = eng.ws["counters"][:M][where].clone()
    if v == 0:
        print("list lengths of view 0: max", int(lengths[v].max()), "mean %.1f" % lengths[v].float().mean().item(), "zeros", int((lengths[v] == 0).sum()), flush=True)
def xcd_remap(b):        # rasterize_common.hpp, SO_TILE_ORDER 2
    grp = b >> 6
    out = torch.where((grp + 1) * 64 > M, b, grp * 64 + ((b & 7) << 3) + ((b >> 3) & 7))
    return out
b = torch.arange(M, device=dev)
default_tbl = xcd_remap(b).to(torch.int32)
def tables(kind, L):
    if kind == "exact":
        return torch.argsort(L, descending=True, stable=True).to(torch.int32)
    if kind == "default":
        return default_tbl.clone()
    if kind == "identity":
        return b.to(torch.int32)
    if kind.startswith("tix"):           # n length classes, longest first, TILE INDEX order inside a class (what k_tile_order does, roughly)
        n = int(kind[3:])
        cls = (L.float() * (n / float(L.max().item() + 1))).long()
        return torch.argsort(-cls, stable=True).to(torch.int32)
    if kind.startswith("classes"):       # n length classes, longest class first, the default (XCD-local) order inside a class
        n = int(kind[7:])
        Ld = L[default_tbl.long()]
        cls = (Ld.float() * (n / float(Ld.max().item() + 1))).long()
        order = torch.argsort(-cls, stable=True)
        return default_tbl[order].clone()
    if kind == "xcd":    # position b runs on XCD b & 7: its tiles, longest first
        out = torch.empty(M, dtype=torch.int32, device=dev)
        for x in range(8):
            pos = b[(b & 7) == x]
            mine = default_tbl[pos].long()
            order = torch.argsort(L[mine], descending=True, stable=True)
            out[pos] = mine[order].to(torch.int32)
        return out
    raise ValueError(kind)
def timed(label, n=400):
    steps(16)
    torch.cuda.synchronize(); t0 = time.time()
    steps(n)
    torch.cuda.synchronize()
    print("%-44s %.4f ms / step" % (label, (time.time() - t0) / n * 1e3), flush=True)
eng.order_refresh, eng.order_max_age = 10 ** 9, 10 ** 9
timed("(a) device-built classes, kept")
saved = {v: eng._order_cache[keys[v]][0].clone() for v in range(8)}
for kind, label in (("exact", "(b) exact global sort"), ("xcd", "(c) longest first within each XCD's tiles"), ("default", "(d) XCD-local order as a table"),
                    ("classes256", "(e) 256 classes, default order inside"), ("identity", "(f) plain tile order as a table"),
                    ("tix256", "(g) 256 classes, tile-index order inside"), ("tix32", "(g) 32 classes"), ("tix8", "(g) 8 classes"), ("tix2", "(g) 2 classes")):
    for v in range(8):
        ent = eng._order_cache[keys[v]]
        ent[0].copy_(tables(kind, lengths[v]))
        ent[1] = 0
    timed(label)
for v in range(8):
    eng._order_cache[keys[v]][0].copy_(saved[v])
timed("(a) again")
