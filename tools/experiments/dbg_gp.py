"""Gaussian-parallel backward experiment (rasterize_gp.hip) against so_rasterize_bwd_packed on the same inputs: the
gradient records of one fused-engine iteration in the dense `ref` regime (and, for contrast, the c2 regime), errors per
record slot and HIP-event timings of both kernels.

    SPLAT_ONE_AMD_LIB=build/variants/libsplat_one_amd_gp.so python tools/experiments/dbg_gp.py [ref|mcmc] [N] [min_len]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                     # noqa: E402
from splat_one_amd import _lib                                   # noqa: E402
from splat_one_amd.scene import front_camera, pinhole_K          # noqa: E402
from splat_one_amd.trainer import Config, Runner                 # noqa: E402

regime = sys.argv[1] if len(sys.argv) > 1 else "ref"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 1920, 1080
dev = torch.device("cuda:0")
init_scale, init_opa = (1.0, 0.1) if regime == "ref" else (0.1, 0.5)
cfg = Config(init_num_pts=N, init_scale=init_scale, init_opa=init_opa, shN_init_std=0.1, sh_degree_interval=1, fused=True,
             device_refine=False, fuse_adam=False)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
c2w, Ks = front_camera()[None].to(dev), pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(3):
    r.train_step(c2w, Ks, pixels)
eng = r._engine
eng.use_graph = False
eng.set_views(c2w, Ks, pixels)
eng.fwd_bwd()
torch.cuda.synchronize()
w, M = eng.ws, eng.M
st = eng.stats()
print(f"{regime}: N {N}, {st['n_isects']} intersections ({st['n_isects'] / M:.1f} per tile), overflow {st['overflow']}, binned {eng.binned}")
lib = _lib.load()
fn = lib.so_exp_rasterize_bwd_gp
fn.restype = ctypes.c_int
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
vrec_ref = w["vrec"].clone()
if eng.binned:
    offsets, n_dev, n_host = w["counters"][:M], None, -eng.bin_capacity
else:
    offsets, n_dev, n_host = w["isect_offsets"], w["counters"][2 * M + 1:], eng.capacity
args_common = (p(w["rec"]), p(offsets), p(w["flatten_ids"]), p(n_dev), ctypes.c_int64(n_host))


def run_gp(out):
    rc = fn(1, eng.N, W, H, 16, *args_common, p(w["render_colors"]), p(w["render_alphas"]), p(w["last_ids"]), p(w["v_render_colors"]),
            p(w["zero_v_alphas"]), p(out), min_len, ctypes.c_void_p(_lib.stream()))
    assert rc == 0, lib.so_last_error()


def run_product(out):
    _lib.call("so_rasterize_bwd_packed", 1, eng.N, W, H, 16, _lib.ptr(w["rec"]), 0, _lib.ptr(offsets), _lib.ptr(w["flatten_ids"]),
              _lib.ptr(n_dev) if n_dev is not None else 0, n_host, _lib.ptr(w["render_alphas"]), _lib.ptr(w["last_ids"]),
              _lib.ptr(w["v_render_colors"]), _lib.ptr(w["zero_v_alphas"]), _lib.ptr(out), 0, _lib.stream())


out_p, out_g = torch.zeros_like(vrec_ref), torch.zeros_like(vrec_ref)
run_product(out_p)
run_gp(out_g)
torch.cuda.synchronize()
a, b, ref = out_p.view(-1, 16)[:, :9].double(), out_g.view(-1, 16)[:, :9].double(), vrec_ref.view(-1, 16)[:, :9].double()
print("product re-run vs the step's own records:", ((a - ref).norm() / ref.norm()).item())
names = ["v_x", "v_y", "v_ca", "v_cb", "v_cc", "v_r", "v_g", "v_b", "v_opac"]
if min_len == 0:
    for s, nme in enumerate(names):
        print(f"  {nme:7s} |gp - product| / |product| = {((b[:, s] - a[:, s]).norm() / a[:, s].norm()).item():.3e}")
    print("all slots:", ((b - a).norm() / a.norm()).item(), " finite:", bool(torch.isfinite(out_g).all()))
for name, f in (("product (quadrant passes)", run_product), ("gaussian-parallel", run_gp)):
    scratch = torch.zeros_like(vrec_ref)
    for _ in range(3):
        f(scratch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        f(scratch)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us per launch")
