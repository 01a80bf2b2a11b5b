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
    if ks:
        shutil.copy(ks[0], os.path.join(DST, f"{tag}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(ks[0])))
        by_stage = {}
        for r in rows:
            st = stage_of(r["Name"])
            if st:
                by_stage.setdefault(st, []).append((int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0]))
        ku = {}
        for st, lst in by_stage.items():
            primary = max(c for c, _, _ in lst)
            ku[st] = round(sum(c * us for c, us, _ in lst) / primary, 2)
        ku["_kernels"] = {st: [n for _, _, n in lst] for st, lst in by_stage.items()}
        ku["_collected_at"] = collected_at(run, "prof")
        ku["_note"] = (f"rocprofv3 --kernel-trace --stats of the bench command of workload {wl}: per stage, sum over its kernels of "
                       f"AverageNs x Calls / calls of the stage's most-launched kernel; source profiles/{tag}_kernel_stats.csv")
    tp = os.path.join(run, "pmc", "traffic_summary.json")
    if os.path.exists(tp):
        d = json.load(open(tp)).get("_per_stage", {})
        traffic = {st: v["hbm_bytes_per_launch_corrected"] for st, v in d.items()}
        traffic["_collected_at"] = collected_at(run, "pmc")
        traffic["_note"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_profiles_r05.sh WL={wl}), KB -> bytes, FETCH_SIZE doubled "
                            f"as MI355X_MICROARCH.md prescribes for gfx950, per launch of each stage's primary kernel; source profiles/{tag}_pmc_traffic.json")
    sp = os.path.join(run, "pmc", "sq_summary.json")
    if os.path.exists(sp):
        d = json.load(open(sp)).get("_per_stage", {})
        valu = {}
        for st, c in d.items():
            if "SQ_INSTS_VALU" not in c:
                continue
            valu[st] = {"wave_instructions": c["SQ_INSTS_VALU"], "waves": c.get("SQ_WAVES"),
                        "active_lane_fraction": (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                                                 if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None),
                        # share of the kernel's SIMD-cycles in which a VALU instruction was executing: SQ_ACTIVE_INST_VALU counts
                        # quad-cycles (x 4) over SQ_BUSY_CYCLES per SE ... reported as measured, see profiles/README.md
                        "valu_active_quad_cycles": c.get("SQ_ACTIVE_INST_VALU"), "busy_cycles": c.get("SQ_BUSY_CYCLES"),
                        "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_inst_any": c.get("SQ_WAIT_INST_ANY")}
        valu["_collected_at"] = collected_at(run, "pmc")
        valu["_note"] = (f"rocprofv3 --pmc SQ_* (two passes, tools/gpu_profiles_r05.sh WL={wl}), per launch of each stage's primary kernel; "
                         f"active_lane_fraction = exec-mask lanes per issued VALU instruction / 64; source profiles/{tag}_sq_counters.json")
        if wl in ("c2", "c2d"):
            valu["_useful_lane_fraction_model"] = {"so_rasterize_bwd": 0.393, "so_rasterize_fwd": 0.393,
                                                   "source": "tools/passsim.py mcmc (profiles/r02_experiments.json)"}
    print(wl, "->", {k: (v if not isinstance(v, dict) else "...") for k, v in (ku or {}).items() if not k.startswith("_")},
          (traffic or {}).get("_collected_at"))
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
