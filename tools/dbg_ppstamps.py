"""Phase times inside k_preprocess_fwd from s_memrealtime stamps (variant library built with -DPP_STAMPS:
tools/build_lib_variant.sh ppstamps "-DPP_STAMPS"; run with SPLAT_ONE_AMD_LIB=build/variants/libsplat_one_amd_ppstamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from splat_one_amd import _lib
from splat_one_amd.scene import pinhole_K, ring_cameras
from splat_one_amd.trainer import Config, Runner
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1 / 1.1)
SPREAD = float(os.environ.get("DBG_SPREAD", "0"))      # bench.py --scale-spread: anisotropic splats of very different sizes
if SPREAD > 0:
    with torch.no_grad():
        gs = torch.Generator().manual_seed(777)
        r.splats["scales"].add_((torch.randn(r.splats["scales"].shape, generator=gs) * SPREAD).to(dev))
        r.splats["quats"].copy_(torch.randn(r.splats["quats"].shape, generator=gs).to(dev))
ring = ring_cameras(8).to(dev); Ks = pinhole_K(W, H)[None].to(dev)
targets = [torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(100 + v)).to(dev) for v in range(8)]
for i in range(int(os.environ.get('DBG_STEPS', '220'))):
    r.train_step(ring[i % 8:i % 8 + 1], Ks, targets[i % 8])
torch.cuda.synchronize()
lib = _lib.load()
n_w = 16384
buf = (ctypes.c_ulonglong * (n_w * 8))()
lib.so_debug_pp_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.so_debug_pp_stamps_read(buf, n_w * 8) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_w, 8).astype(np.int64)
real = (N + 63) // 64
s = st[:real, :5]
t0 = s[:, 0].min()
rel = (s - t0) * 0.01      # 100 MHz -> us
print("waves with work", real, " kernel span (first start .. last end) %.1f us" % rel[:, 4].max())
print("start of a wave after the first:  mean %.1f  p50 %.1f  p95 %.1f  max %.1f us" % (rel[:, 0].mean(), np.median(rel[:, 0]), np.percentile(rel[:, 0], 95), rel[:, 0].max()))
names = ["params + projection", "SH colour", "rec / vrec stores", "binning (cull tests + atomics + key stores)"]
for k in range(4):
    d = rel[:, k + 1] - rel[:, k]
    print("%-44s mean %6.2f  p50 %6.2f  p95 %6.2f  max %6.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 95), d.max()))
life = rel[:, 4] - rel[:, 0]
print("wave life: mean %.2f p50 %.2f p95 %.2f max %.2f us" % (life.mean(), np.median(life), np.percentile(life, 95), life.max()))
idle = st[real:, :5]
idle = idle[idle[:, 0] > 0]
if len(idle):
    print("waves without work:", len(idle), " start mean %.1f us, life mean %.2f us" % (((idle[:, 0] - t0) * 0.01).mean(), ((idle[:, 4] - idle[:, 0]) * 0.01).mean()))

# ---- k_preprocess_bwd (with the fused Adam)
assert lib.so_debug_ppb_stamps_read(buf, n_w * 8) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_w, 8).astype(np.int64)
s = st[:real, :5]
t0 = s[:, 0].min()
rel = (s - t0) * 0.01
print("k_preprocess_bwd: kernel span %.1f us" % rel[:, 4].max())
names = ["loads + projection bwd + SH bwd (camera loop)", "Adam on means/scales/quats/opacity/sh0 (per lane)", "LDS row completion + barrier", "coalesced shN sweep (Adam)"]
for k in range(4):
    d = rel[:, k + 1] - rel[:, k]
    print("%-52s mean %6.2f  p50 %6.2f  p95 %6.2f  max %6.2f us" % (names[k], d.mean(), np.median(d), np.percentile(d, 95), d.max()))
life = rel[:, 4] - rel[:, 0]
print("wave life: mean %.2f p50 %.2f p95 %.2f max %.2f us;  start spread p95 %.2f us" % (life.mean(), np.median(life), np.percentile(life, 95), life.max(), np.percentile(rel[:, 0], 95)))

# when do the working waves START (workgroup = 4 consecutive waves; the grid is capacity-sized, the working workgroups come first)
starts = rel[:, 0]
print("k_preprocess_bwd wave starts, deciles (us):", [round(float(np.percentile(starts, q)), 1) for q in range(0, 101, 10)])
wg = starts[: (real // 4) * 4].reshape(-1, 4).min(axis=1)
ends = rel[: (real // 4) * 4, 4].reshape(-1, 4).max(axis=1)
for a in range(0, len(wg), 64):
    print("  workgroups %4d..%4d: start mean %5.1f  min %5.1f  max %5.1f   end mean %5.1f us" % (a, min(a + 63, len(wg) - 1), wg[a:a + 64].mean(), wg[a:a + 64].min(), wg[a:a + 64].max(), ends[a:a + 64].mean()))
