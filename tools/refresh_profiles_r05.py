"""Copies the round-5 measurement sets (tools/gpu_profiles_r05.sh, one per workload: gpurun_out/r05_<WL>/) into profiles/ and
regenerates profiles/traffic.json / valu.json / kernel_us.json, which bench.py reads for `roofline.traffic`, `roofline.valu`
and `roofline.kernel_us_rocprof`:

    root of each file      the c2 collection at the default command line (WL=c2), "_also": [the driver's command line, WL=c2d]
    "_by_workload": {key}  one collection per other named workload (bench.py `workload_key`: c2-ref, c3, c4, c5, 2M-f32)

every collection keyed by STAGE (entry-point name) with `_collected_at.tile_intersections` = what the profiled bench runs
themselves reported, so that bench.py can refuse a collection made at another workload state.

    python tools/refresh_profiles_r05.py
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SRC, DST = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

STAGES = [("so_preprocess_fwd", ("k_preprocess_fwd",)), ("so_preprocess_bwd", ("k_preprocess_bwd",)),
          ("so_rasterize_fwd", ("k_rasterize_fwd",)), ("so_rasterize_bwd", ("k_rasterize_bwd",)),
          ("so_ssim_l1_fused", ("k_ssim_l1_fused",)), ("so_isect_fill", ("k_tile_sort", "k_tile_order", "k_isect", "k_scan_tiles", "k_tile_merge")),
          ("so_step_inputs", ("k_step_inputs",)), ("so_adam_step_dev", ("k_adam",))]


def stage_of(kernel):
    base = kernel.split("(")[0]
    for st, pres in STAGES:
        if any(("so::" + p) in base for p in pres):
            return st
    return None


def collected_at(run_dir, kind):
    vals = []
    for f in glob.glob(os.path.join(run_dir, kind, "stdout*.txt")):
        for line in open(f):
            if line.startswith("{") and "tile_intersections" in line:
                try:
                    j = json.loads(line)
                    vals.append((j["config"]["tile_intersections"], j["steps"], j["warmup"], j["config"].get("workload_key")))
                except Exception:   # noqa: BLE001
                    pass
    if not vals:
        return None
    return {"tile_intersections": int(sum(v[0] for v in vals) / len(vals)), "steps": vals[0][1], "warmup": vals[0][2],
            "runs": len(vals), "workload": vals[0][3]}


COMPLEMENTARY = ("so_isect_fill",)      # stages whose kernels ALL run in one iteration (sum); elsewhere they are alternatives (mean)


def aggregate(per_kernel, calls):
    """{stage: value per launch} from {kernel: mean per launch} and {kernel: launches}: the per-tile sort launches up to four
    kernels per iteration (their means add, weighted by how often each ran); every other stage is ONE kernel per iteration out
    of several variants (SH-degree instantiations of the first steps, the two backward rasterisers): launch-weighted mean."""
    by_stage = {}
    for k, v in per_kernel.items():
        st = stage_of(k)
        if st and calls.get(k):
            by_stage.setdefault(st, []).append((calls[k], v))
    out = {}
    for st, lst in by_stage.items():
        tot = sum(c * v for c, v in lst)
        out[st] = tot / (max(c for c, _ in lst) if st in COMPLEMENTARY else sum(c for c, _ in lst))
    return out


def one(wl):
    run = os.path.join(SRC, "r05_" + wl)
    if not os.path.isdir(run):
        return None
    tag = "r05_" + wl.replace("-", "_")
    copies = [("bench.json", f"{tag}_bench.json"), ("pmc/traffic_summary.json", f"{tag}_pmc_traffic.json"),
              ("pmc/sq_summary.json", f"{tag}_sq_counters.json")]
    for a, b in copies:
        if os.path.exists(os.path.join(run, a)):
            shutil.copy(os.path.join(run, a), os.path.join(DST, b))
    ks = sorted(glob.glob(os.path.join(run, "prof", "**", "bench_kernel_stats.csv"), recursive=True))
    traffic = valu = ku = None
    calls = {}
    if ks:
        shutil.copy(ks[0], os.path.join(DST, f"{tag}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(ks[0])))
        calls = {r["Name"].split("(")[0]: int(r["Calls"]) for r in rows}
        ku = {st: round(v, 2) for st, v in aggregate({r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in rows}, calls).items()}
        ku["_kernels"] = {}
        for r in rows:
            st = stage_of(r["Name"])
            if st:
                ku["_kernels"].setdefault(st, []).append({"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)})
        ku["_collected_at"] = collected_at(run, "prof")
        ku["_note"] = (f"rocprofv3 --kernel-trace --stats of the bench command of workload {wl}: per stage the launch-weighted mean of its kernel "
                       f"variants (the per-tile sort: the sum of its kernels); source profiles/{tag}_kernel_stats.csv")
    tp = os.path.join(run, "pmc", "traffic_summary.json")
    if os.path.exists(tp):
        d = {k: v for k, v in json.load(open(tp)).items() if not k.startswith("_")}
        c2 = {k: v["launches"] for k, v in d.items()}
        traffic = aggregate({k: v["hbm_bytes_per_launch_corrected"] for k, v in d.items()}, c2)
        traffic["_collected_at"] = collected_at(run, "pmc")
        traffic["_note"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_profiles_r05.sh WL={wl}), KB -> bytes, FETCH_SIZE doubled "
                            f"as MI355X_MICROARCH.md prescribes for gfx950, per launch (launch-weighted over a stage's kernel variants; the sort: summed); "
                            f"source profiles/{tag}_pmc_traffic.json")
    sp = os.path.join(run, "pmc", "sq_summary.json")
    if os.path.exists(sp):
        d = {k: v for k, v in json.load(open(sp)).items() if not k.startswith("_")}
        names = sorted({c for v in d.values() for c in v})
        per = {c: aggregate({k: v[c] for k, v in d.items() if c in v}, calls) for c in names}
        valu = {}
        for st in per.get("SQ_INSTS_VALU", {}):
            g = lambda c: per.get(c, {}).get(st)
            valu[st] = {"wave_instructions": round(g("SQ_INSTS_VALU")), "waves": g("SQ_WAVES"),
                        "active_lane_fraction": (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
                                                 if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU") else None),
                        "valu_active_quad_cycles": g("SQ_ACTIVE_INST_VALU"), "busy_cycles": g("SQ_BUSY_CYCLES"),
                        "wave_quad_cycles": g("SQ_WAVE_CYCLES"), "wait_inst_any": g("SQ_WAIT_INST_ANY")}
            us = (ku or {}).get(st)
            if us and g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
                simd_cycles = us * 1e-6 * 2.4e9 * 1024            # 1024 SIMDs at 2.4 GHz over the kernel's rocprofv3 duration
                valu[st]["vector_pipe_busy"] = round(4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles, 3)
                valu[st]["resident_waves_per_simd"] = round(4.0 * g("SQ_WAVE_CYCLES") / simd_cycles, 2)
        valu["_collected_at"] = collected_at(run, "pmc")
        valu["_note"] = (f"rocprofv3 --pmc SQ_* (two passes, tools/gpu_profiles_r05.sh WL={wl}), per launch; active_lane_fraction = exec-mask lanes per "
                         f"issued VALU instruction / 64; vector_pipe_busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x 2.4 GHz x the kernel's "
                         f"rocprofv3 duration), resident_waves_per_simd = 4 x SQ_WAVE_CYCLES over the same; source profiles/{tag}_sq_counters.json")
        if wl in ("c2", "c2d"):
            valu["_useful_lane_fraction_model"] = {"so_rasterize_bwd": 0.393, "so_rasterize_fwd": 0.393,
                                                   "source": "tools/passsim.py mcmc (profiles/r02_experiments.json)"}
    print(wl, "->", {k: v for k, v in (ku or {}).items() if not k.startswith("_")}, (traffic or {}).get("_collected_at"))
    return traffic, valu, ku


main = one("c2")
also = one("c2d")
others = {wl: one(wl) for wl in ("c2-ref", "c3", "c4", "c5", "2M-f32")}
for i, fname in enumerate(("traffic.json", "valu.json", "kernel_us.json")):
    path = os.path.join(DST, fname)
    old = json.load(open(path)) if os.path.exists(path) else {}
    out = dict(main[i]) if (main and main[i]) else {k: v for k, v in old.items() if k not in ("_also", "_by_workload")}
    if also and also[i]:
        out["_also"] = [also[i]]
    elif "_also" in old and not (main and main[i]):
        out["_also"] = old["_also"]
    byw = dict(old.get("_by_workload") or {})
    for wl, res in others.items():
        if res and res[i]:
            byw[wl] = res[i]
    if byw:
        out["_by_workload"] = byw
    json.dump(out, open(path, "w"), indent=1)
    print(fname, "root:", (out.get("_collected_at") or {}), "also:", [c.get("_collected_at") for c in out.get("_also", [])],
          "by_workload:", {k: (v.get("_collected_at") or {}).get("tile_intersections") for k, v in byw.items()})
