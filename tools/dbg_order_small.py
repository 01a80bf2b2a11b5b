"""Seed 2344 of the rasterization() fuzzer fails its forward bar only after seed 2222 (same C x tiles = 30, other grid: the two
share a _Bins).  Where does the image differ, and is it the workgroup -> tile table that seed 2222's long lists switch on?
    python tools/dbg_order_small.py [FIRST SECOND]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F
import splat_one_amd.raster_op as R

dev = torch.device("cuda:0")
first, second = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2222, 2344)


def report(tag, seed):
    cfg, (rc_h, rc_o), (ra_h, ra_o), g_h, g_o = F._operator_against_the_oracle(dev, seed)
    d = (rc_h - rc_o).abs()
    C, H, W = d.shape[:3]
    ts = cfg["tile"]
    th, tw = -(-H // ts), -(-W // ts)
    print(f"[{tag}] seed {seed} {cfg['model']} C={C} {W}x{H}: forward mean |diff| {d.mean().item():.3e} max {d.max().item():.3e}", flush=True)
    for b in R._BINS.values():
        if b.M == C * th * tw:
            o = None if b.last_order is None else b.last_order.cpu().tolist()
            print(f"   bins M={b.M} slots={b.slots} replicas={b.replicas} mean_list={b.mean_list:.1f} fullest={b.fullest} probed={b.probed} n_probe={b.n_probe}")
            print("   table:", o, "" if o is None else ("permutation" if sorted(o) == list(range(b.M)) else "NOT A PERMUTATION"))
    if d.mean().item() > 1e-5:
        for c in range(C):
            for y in range(th):
                print("   cam", c, "row", y, " ".join("%.1e" % d[c, y * ts:(y + 1) * ts, x * ts:(x + 1) * ts].mean().item() for x in range(tw)))
        da = (ra_h - ra_o).abs()
        print("   alpha mean |diff| %.3e" % da.mean().item())


report("alone", second)
R._BINS.clear()
report("history", first)
report("after the history", second)
report("again", second)
R._BINS.clear()
orig = R.pick_tile_order
R.pick_tile_order = lambda *a: True
report("alone, table forced on", second)
report("again, table forced on", second)
R.pick_tile_order = lambda *a: False
R._BINS.clear()
report("history, table forced off", first)
report("after the history, table forced off", second)
