import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from splat_one_amd import _lib
dev = torch.device("cuda:0")
# one-hot probes: which lanes contribute to which output slot
for col in range(9):
    res = []
    for lane in range(64):
        x = torch.zeros(64, 9); x[lane, col] = 1.0
        out = torch.empty(1, 10, device=dev)
        xd = x.to(dev)
        _lib.call("so_debug_wave_reduce", 1, _lib.ptr(xd), _lib.ptr(out), _lib.stream())
        res.append(out[0].cpu())
    R = torch.stack(res)  # [lane, 10]
    if col < 8:
        print("col", col, "slot sums per lane-source:", R[:, col].tolist().count(1.0), "of 64; leaked to other slots:", int((R[:, :8].sum(1) - R[:, col]).abs().sum().item()))
        if R[:, col].tolist().count(1.0) != 64:
            print("   counts", R[:, col].tolist())
    else:
        print("col 8 allreduce", R[:, 8].tolist().count(1.0), "dpp", R[:, 9].tolist().count(1.0))
