"""Timing of so_rec_pack / so_rec_unpack_grads alone (HIP events, warm loop)."""
import torch
from splat_one_amd import _lib
from splat_one_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
for n in (100_000, 1_000_000):
    m = torch.rand(n, 2, device=dev); cn = torch.rand(n, 3, device=dev); col = torch.rand(n, 3, device=dev); op = torch.rand(n, device=dev)
    rec = torch.empty(n, 16, device=dev); vrec = torch.empty(n, 16, device=dev)
    outs = [torch.empty(n, 2, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev)]
    for name, fn in (("so_rec_pack", lambda: call("so_rec_pack", n, ptr(m), ptr(cn), ptr(col), ptr(op), ptr(rec), 0, stream())),
                     ("so_rec_pack+zero", lambda: call("so_rec_pack", n, ptr(m), ptr(cn), ptr(col), ptr(op), ptr(rec), ptr(vrec), stream())),
                     ("so_rec_unpack_grads", lambda: call("so_rec_unpack_grads", n, ptr(vrec), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), 0, stream()))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(n, name, f"{a.elapsed_time(b) / 50 * 1e3:.1f} us")
    r2 = rec.clone()
    assert torch.equal(r2[:, 0:2], m) and torch.equal(r2[:, 2:5], cn) and torch.equal(r2[:, 5], op) and torch.equal(r2[:, 6:9], col)
print("ok")
