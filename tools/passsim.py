import sys, torch, numpy as np, time
sys.path.insert(0,'/root/repo')
from oracle import torch_oracle as O
from splat_one_amd.scene import make_scene
regime = sys.argv[1] if len(sys.argv)>1 else "mcmc"
W,H,N=1920,1080,100000
splats,c2w,Ks=make_scene(N,W,H,regime=regime)
with torch.no_grad():
    radii,m2d,depths,conics,_=O.fully_fused_projection(splats["means"],None,splats["quats"],torch.exp(splats["scales"]),torch.linalg.inv(c2w),Ks,W,H,near_plane=0.01,far_plane=1e8,dtype=torch.float32)
    tw,th=(W+15)//16,(H+15)//16
    tpg,ids,fl=O.isect_tiles(m2d,radii,depths,16,tw,th)
    off=O.isect_offset_encode(ids,1,tw,th).reshape(-1).numpy()
op=torch.sigmoid(splats["opacities"]).numpy()
m2=m2d[0].numpy(); cn=conics[0].numpy(); fl=fl.numpy()
I=len(fl); off=np.append(off,I)
tile_of=np.repeat(np.arange(tw*th),np.diff(off))
g=fl
ty,tx=np.divmod(tile_of,tw)
# pixel grid 16x16
py,px=np.meshgrid(np.arange(16)+0.5,np.arange(16)+0.5,indexing="ij")
tot_pass=dict(q8x8=0,b4x4_max=0,b4x4_sum=0,b2x4_max=0,h4x8_max=0)
valid_px=0; pairs_any=0
B=200000
# per quadrant accumulators for max computations: counts per (tile, quadrant, block)
cnt4=np.zeros((tw*th,4,4),np.int32)   # [tile, quadrant, 4x4 block]
cnt8=np.zeros((tw*th,4,8),np.int32)   # 2(rows)x4(cols)? use 8 groups of 8 lanes: rows of 8 pixels (1x8)
cnt2=np.zeros((tw*th,4,2),np.int32)
cntq=np.zeros((tw*th,4),np.int32)
half_tb=half_lr=0   # (tile, Gaussian, half) pairs with a valid pixel: 16x8 halves (top/bottom), 8x16 halves (left/right)
for s in range(0,I,B):
    e=min(I,s+B)
    gg=g[s:e]
    dx=m2[gg,0][:,None,None]-(tx[s:e,None,None]*16+px[None])
    dy=m2[gg,1][:,None,None]-(ty[s:e,None,None]*16+py[None])
    sig=0.5*(cn[gg,0][:,None,None]*dx*dx+cn[gg,2][:,None,None]*dy*dy)+cn[gg,1][:,None,None]*dx*dy
    al=np.minimum(0.999,op[gg][:,None,None]*np.exp(-sig))
    inside=((tx[s:e,None,None]*16+px[None])<W)&((ty[s:e,None,None]*16+py[None])<H)
    v=(sig>=0)&(al>=1/255)&inside            # [n,16,16]
    valid_px+=v.sum()
    pairs_any+=v.any(axis=(1,2)).sum()
    # quadrants: [n, qy, 8, qx, 8]
    vq=v.reshape(-1,2,8,2,8).transpose(0,1,3,2,4).reshape(-1,4,8,8)   # [n,quadrant,8,8]
    anyq=vq.any(axis=(2,3))                                          # [n,4]
    b4=vq.reshape(-1,4,2,4,2,4).transpose(0,1,2,4,3,5).reshape(-1,4,4,16).any(axis=3)   # [n,quadrant,4 blocks]
    b8=vq.any(axis=3)                                                # [n,quadrant,8 rows of 8 px]
    h2=vq.reshape(-1,4,2,32).any(axis=3)
    half_tb+=(anyq[:,0]|anyq[:,1]).sum()+(anyq[:,2]|anyq[:,3]).sum()
    half_lr+=(anyq[:,0]|anyq[:,2]).sum()+(anyq[:,1]|anyq[:,3]).sum()
    np.add.at(cntq,tile_of[s:e],anyq.astype(np.int32))
    np.add.at(cnt4,tile_of[s:e],b4.astype(np.int32))
    np.add.at(cnt8,tile_of[s:e],b8.astype(np.int32))
    np.add.at(cnt2,tile_of[s:e],h2.astype(np.int32))
print("regime",regime,"pairs (tile,gaussian) in lists",I,"with any valid pixel",pairs_any,"valid (pixel,gaussian)",valid_px)
cur=cntq.sum()
print("current 8x8 passes",cur,"lane util",valid_px/(64*cur))
# several pixels per lane (VERDICT r1 task 3, third variant): a wave owns a 16x8 / 8x16 half tile (2 pixels per lane) or the
# whole tile (4 per lane); the per-pass cost that does not depend on the pixel count (cross-lane reduction, atomics, list
# walk: ~55 of the 143 instructions of a pass) is paid once per pass, the per-pixel part (~88) k times
S_,P_=55,88
for name,passes,k in (("8x8 quadrants (now)",cur,1),("16x8 halves",half_tb,2),("8x16 halves",half_lr,2),("16x16 tile",pairs_any,4)):
    print(f"{name:22s} passes {passes:9d}  instr/pass {S_+P_*k:4d}  issue total {passes*(S_+P_*k)/1e6:8.1f} M  vs now {passes*(S_+P_*k)/(cur*(S_+P_)):.3f}  pixel-slot util {valid_px/(64*k*passes):.3f}")
for name,c in (("4x4 rows (4 groups of 16)",cnt4),("1x8 rows (8 groups of 8)",cnt8),("4x8 halves (2 groups of 32)",cnt2)):
    it=c.max(axis=2).sum(); su=c.sum()
    print(name,"iterations",it,"ratio vs current",it/cur,"lane util",valid_px/(64*it),"sum of group passes",su, "ideal(avg) iterations",su/c.shape[2]/cur)

# ---- bbox-only block test (conservative): axis-aligned bound of the alpha >= 1/255 ellipse vs the block's pixel-centre rectangle
det=cn[:,0]*cn[:,2]-cn[:,1]**2
tau=np.maximum(np.log(np.maximum(op*255,1e-30)),0)
with np.errstate(all="ignore"):
    hx=np.sqrt(2*tau/det*cn[:,2])+0.01; hy=np.sqrt(2*tau/det*cn[:,0])+0.01
cnt4b=np.zeros((tw*th,4,4),np.int32); cntqb=np.zeros((tw*th,4),np.int32)
for s in range(0,I,B):
    e=min(I,s+B); gg=g[s:e]
    x=m2[gg,0]-tx[s:e]*16; y=m2[gg,1]-ty[s:e]*16      # centre relative to tile origin
    hits=np.zeros((e-s,4,4),bool); hq=np.zeros((e-s,4),bool)
    for q in range(4):
        qx0,qy0=(q&1)*8+0.5,(q>>1)*8+0.5
        hq[:,q]=~((x+hx[gg]<qx0)|(x-hx[gg]>qx0+7)|(y+hy[gg]<qy0)|(y-hy[gg]>qy0+7))
        for r in range(4):
            x0,y0=qx0+(r&1)*4,qy0+(r>>1)*4
            hits[:,q,r]=~((x+hx[gg]<x0)|(x-hx[gg]>x0+3)|(y+hy[gg]<y0)|(y-hy[gg]>y0+3))
    np.add.at(cnt4b,tile_of[s:e],hits.astype(np.int32)); np.add.at(cntqb,tile_of[s:e],hq.astype(np.int32))
print("bbox-only: quadrant passes",cntqb.sum(),"(exact",cur,") 4x4 rows iterations",cnt4b.max(axis=2).sum(),"ratio vs current exact",cnt4b.max(axis=2).sum()/cur)
