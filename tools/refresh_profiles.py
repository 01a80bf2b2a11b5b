"""Copies the summaries of the last gpurun measurement calls into profiles/ (tracked) and regenerates
profiles/traffic.json (entry point -> corrected HBM bytes per launch, read by bench.py).

    python tools/refresh_profiles.py r02      # after tools/gpu_profiles_r02.sh and tools/gpu_c4_r02.sh
"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def first(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


pairs = [(f"{tag}/prof/**/bench_kernel_stats.csv", f"{tag}_engine_kernel_stats.csv"),
         (f"{tag}/pmc/traffic_summary.json", f"{tag}_engine_pmc_traffic.json"),
         (f"{tag}/bench.json", f"{tag}_engine_bench.json"),
         (f"{tag}/other_lines.jsonl", f"{tag}_other_bench_lines.jsonl"),
         (f"{tag}/pmc/sq_summary.json", f"{tag}_engine_sq_counters.json"),
         (f"{tag}/hip/hip_api_diff.json", f"{tag}_mcmc_refine_hip_api_calls.json"),
         (f"c4{tag}/bench.json", f"{tag}_c4_densify_bench.json"),
         (f"c4{tag}/**/c4_kernel_stats.csv", f"{tag}_c4_densify_kernel_stats.csv"),
         (f"c4{tag}/hip_api_diff.json", f"{tag}_c4_refine_hip_api_calls.json"),
         (f"parity_{tag}.json", f"parity_{tag}.json")]
for a, b in pairs:
    p = first(a)
    if p and b.startswith("parity_") and os.path.exists(os.path.join(dst, b)):
        # a test run records only the cases it ran: merge into the tracked file, newest entry per case wins
        old, new = json.load(open(os.path.join(dst, b))), json.load(open(p))
        old.update(new)
        json.dump(old, open(os.path.join(dst, b), "w"), indent=1, sort_keys=True)
        print("merged", os.path.relpath(p, ROOT), "->", b, f"({len(new)} cases of {len(old)})")
    elif p:
        shutil.copy(p, os.path.join(dst, b))
        print("copied", os.path.relpath(p, ROOT), "->", b)
    else:
        print("MISSING", a)
names = {"so_rasterize_bwd": "void so::k_rasterize_bwd<3, 16, false, true, true>", "so_rasterize_fwd": "void so::k_rasterize_fwd<3, 16, true>",
         "so_adam_step_dev": "so::k_adam_dev", "so_ssim_l1_fwd": "void so::k_ssim_l1_fwd<3>", "so_ssim_l1_bwd": "void so::k_ssim_l1_bwd<3>",
         "so_ssim_l1_fused": "void so::k_ssim_l1_fused<3>",
         "so_preprocess_fwd": "void so::k_preprocess_fwd<3, so::AttrSoA, false>",
         "so_preprocess_bwd": "void so::k_preprocess_bwd<3, so::AttrSoA, true, true, false>",
         "so_isect_fill": "void so::k_tile_sort_waves<256, 2048>"}


def collected_at(run, kind):
    """tile intersections of the profiled bench command itself (its JSON line is in the pass's stdout file)"""
    vals = []
    for f in glob.glob(os.path.join(src, run, kind, "stdout*.txt")):
        for line in open(f):
            if line.startswith("{") and "tile_intersections" in line:
                try:
                    j = json.loads(line)
                    vals.append((j["config"]["tile_intersections"], j["steps"], j["warmup"]))
                except Exception:
                    pass
    if not vals:
        return None
    return {"tile_intersections": int(sum(v[0] for v in vals) / len(vals)), "steps": vals[0][1], "warmup": vals[0][2], "runs": len(vals)}


def collections(run, label):
    """(traffic, valu, kernel_us) of one counter collection: `run` = its directory under gpurun_out/, `label` = the prefix of
    its copies in profiles/"""
    import csv
    traffic = valu = ku = None
    tp = os.path.join(dst, f"{label}_engine_pmc_traffic.json")
    if os.path.exists(tp):
        d = json.load(open(tp))
        traffic = {k: d[v]["hbm_bytes_per_launch_corrected"] for k, v in names.items() if v in d}
        traffic["_collected_at"] = collected_at(run, "pmc")
        traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_profiles_%s.sh) of the bench command (c2, 8 ring "
                            "views cycled), KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; source "
                            "profiles/%s_engine_pmc_traffic.json" % (tag, label))
    # the VALU side: wave-instructions and exec-mask lane cycles per launch (bench.py roofline.valu)
    sqp = os.path.join(dst, f"{label}_engine_sq_counters.json")
    if os.path.exists(sqp):
        sq = json.load(open(sqp))
        valu = {}
        for k, v in names.items():
            c = sq.get(v)
            if c and "SQ_INSTS_VALU" in c:
                valu[k] = {"wave_instructions": c["SQ_INSTS_VALU"], "waves": c.get("SQ_WAVES"),
                           "active_lane_fraction": (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                                                    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None)}
        valu["_collected_at"] = collected_at(run, "pmc")
        valu["_note"] = ("rocprofv3 --pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU (tools/gpu_profiles_%s.sh), means per launch of the "
                         "bench command (c2); active_lane_fraction = exec-mask lanes per issued VALU instruction / 64; source "
                         "profiles/%s_engine_sq_counters.json" % (tag, label))
        # lanes that carry a contributing (pixel, Gaussian) pair in a pass of the rasteriser kernels: tools/passsim.py on the c2 front view
        valu["_useful_lane_fraction_model"] = {"so_rasterize_bwd": 0.393, "so_rasterize_fwd": 0.393, "source": "tools/passsim.py mcmc (profiles/r02_experiments.json)"}
    # rocprofv3's per-kernel averages of the bench command, keyed by entry point (bench.py roofline.kernel_us_rocprof)
    ks = os.path.join(dst, f"{label}_engine_kernel_stats.csv")
    if os.path.exists(ks):
        rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(ks))}
        ku = {k: round(rows[v], 2) for k, v in names.items() if v in rows}
        ku["_collected_at"] = collected_at(run, "prof")
        ku["_note"] = "rocprofv3 --kernel-trace --stats of the bench command, AverageNs per kernel; source profiles/%s_engine_kernel_stats.csv" % label
    return traffic, valu, ku


main = collections(tag, tag)
# a second collection at the round-end driver's command line (`--steps 20 --warmup 5`: RTAG=<tag>d tools/gpu_profiles_<tag>.sh):
# kept under "_also"; bench.py uses whichever collection was made at (nearly) its own tile-intersection count
also = None
if os.path.isdir(os.path.join(src, tag + "d")):
    for a, b in [(f"{tag}d/prof/**/bench_kernel_stats.csv", f"{tag}d_engine_kernel_stats.csv"),
                 (f"{tag}d/pmc/traffic_summary.json", f"{tag}d_engine_pmc_traffic.json"),
                 (f"{tag}d/pmc/sq_summary.json", f"{tag}d_engine_sq_counters.json"),
                 (f"{tag}d/bench.json", f"{tag}d_engine_bench.json")]:
        p = first(a)
        if p:
            shutil.copy(p, os.path.join(dst, b))
            print("copied", os.path.relpath(p, ROOT), "->", b)
    also = collections(tag + "d", tag + "d")
for i, fname in enumerate(("traffic.json", "valu.json", "kernel_us.json")):
    if main[i] is None:
        continue
    out = dict(main[i])
    if also is not None and also[i] is not None:
        out["_also"] = [also[i]]
    json.dump(out, open(os.path.join(dst, fname), "w"), indent=1)
    print(fname, {k: v for k, v in out.items() if not k.startswith("_")}, out.get("_collected_at"),
          [c.get("_collected_at") for c in out.get("_also", [])])
