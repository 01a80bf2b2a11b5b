import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd.scene import pinhole_K, front_camera
from splat_one_amd.trainer import Config, Runner
from splat_one_amd.engine import FusedEngine
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
cfg = Config(init_num_pts=N, init_scale=0.1, init_opa=0.5, shN_init_std=0.1, sh_degree_interval=1, fused=True)
r = Runner(0, 0, 1, cfg, scene_scale=1/1.1)
c2w = front_camera()[None].to(dev); Ks = pinhole_K(W, H)[None].to(dev)
pixels = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(230):
    r.train_step(c2w, Ks, pixels)
eng = r._engine
st = eng.stats(); I = st["n_isects"]
rec = eng.ws["rec"]; off = eng.ws["isect_offsets"].reshape(-1).long(); fid = eng.ws["flatten_ids"][:I].long()
M = off.numel(); tw = math.ceil(W/16)
ends = torch.cat([off[1:], torch.tensor([I], device=dev)])
tile_of = torch.repeat_interleave(torch.arange(M, device=dev), ends - off)
R = rec[fid]
mx, my, ca, cb, cc, op = R[:,0], R[:,1], R[:,2], R[:,3], R[:,4], R[:,5]
ty, tx = tile_of // tw, tile_of % tw
tau = torch.log(op*255).clamp_min(0)
tot_pass = 0; covered = 0
last = eng.ws["last_ids"].reshape(-1)
for q in range(4):
    x0 = (tx*16 + (q&1)*8).float()+0.5 - mx; x1 = x0+7; y0 = (ty*16 + (q>>1)*8).float()+0.5 - my; y1 = y0+7
    inside = (x0<=0)&(x1>=0)&(y0<=0)&(y1>=0)
    best = torch.full_like(mx, float("inf"))
    for x in (x0, x1):
        y = torch.minimum(torch.maximum(-cb*x/cc, y0), y1); best = torch.minimum(best, 0.5*(ca*x*x+cc*y*y)+cb*x*y)
    for y in (y0, y1):
        x = torch.minimum(torch.maximum(-cb*y/ca, x0), x1); best = torch.minimum(best, 0.5*(ca*x*x+cc*y*y)+cb*x*y)
    hit = inside | (best <= tau)
    tot_pass += int(hit.sum())
    # covered pixel pairs (alpha>=1/255) in this quadrant
    xs = torch.arange(8, device=dev).float()
    dx = x0[:,None,None] + xs[None,None,:]; dy = y0[:,None,None] + xs[None,:,None]
    for chunk in range(0, I, 200000):
        s = slice(chunk, chunk+200000)
        sig = 0.5*(ca[s,None,None]*dx[s]**2 + cc[s,None,None]*dy[s]**2) + cb[s,None,None]*dx[s]*dy[s]
        al = op[s,None,None]*torch.exp(-sig)
        covered += int(((al >= 1/255) & hit[s,None,None]).sum())
print("I", I, "quadrant passes", tot_pass, "per isect", tot_pass/I, "covered pixel pairs", covered, "lane utilisation", covered/(tot_pass*64))
print("tile list len: mean", (ends-off).float().mean().item(), "max", (ends-off).max().item())
