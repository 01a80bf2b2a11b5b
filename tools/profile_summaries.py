"""Summaries of one workload's rocprofv3 passes (tools/gpu_profiles_r05.sh, run on the GPU box): per KERNEL the mean per launch
of FETCH_SIZE / WRITE_SIZE (KB) with the corrected HBM bytes (FETCH_SIZE x 2 on gfx950 for wide streaming reads, WRITE_SIZE as
it is -- /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section) and of every SQ counter; and per STAGE of the training
step (entry-point name, as bench.py's roofline_by_kernel keys) the totals per training iteration, summed over the kernels a
stage launches (the per-tile sort is up to four kernels; the backward rasteriser one of two)."""
import collections
import csv
import glob
import json
import os

root = os.environ["GRAFT_REPO_ROOT"]
out = os.path.join(root, "gpurun_out", "r05_" + os.environ.get("WL", "c2"))

STAGES = [("so_preprocess_fwd", ("k_preprocess_fwd",)), ("so_preprocess_bwd", ("k_preprocess_bwd",)),
          ("so_rasterize_fwd", ("k_rasterize_fwd",)), ("so_rasterize_bwd", ("k_rasterize_bwd",)),
          ("so_ssim_l1_fused", ("k_ssim_l1_fused",)), ("so_isect_fill", ("k_tile_sort", "k_tile_order", "k_isect", "k_scan_tiles", "k_tile_merge")),
          ("so_step_inputs", ("k_step_inputs",)), ("so_adam_step_dev", ("k_adam",)), ("refine", ("k_refine", "k_reset_opacity", "k_attr_pack"))]


def stage_of(kernel):
    base = kernel.split("(")[0]
    for st, pres in STAGES:
        if any(("so::" + p) in base for p in pres):
            return st
    return None


def per_kernel(files, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


for r in sorted(glob.glob(out + "/prof/**/bench_kernel_stats.csv", recursive=True))[:1]:
    for row in list(csv.DictReader(open(r)))[:16]:
        print(f"{row['Name'][:84]:84s} calls {row['Calls']:>6s} avg_us {float(row['AverageNs']) / 1e3:10.1f} pct {row['Percentage']}")

def stage_totals(acc):
    """{stage: {counter: total per launch of the stage's PRIMARY kernel (the one launched most often)}, "_kernels": [...]}: a
    stage that is one kernel gets that kernel's mean per launch; the per-tile sort (up to four kernels per iteration) and the
    backward rasteriser (one of two) get the sum over their kernels."""
    by_stage = collections.defaultdict(dict)
    for k, v in acc.items():
        st = stage_of(k)
        if st:
            by_stage[st][k] = v
    res = {}
    for st, ks in by_stage.items():
        primary = max(max(len(x) for x in v.values()) for v in ks.values())
        tot = collections.defaultdict(float)
        for k, v in ks.items():
            for c, x in v.items():
                tot[c] += sum(x) / primary
        res[st] = dict(tot)
        res[st]["_launches"] = primary
        res[st]["_kernels"] = sorted(ks)
    return res


traffic = per_kernel(glob.glob(out + "/pmc/**/pmc_*_counter_collection.csv", recursive=True), ("FETCH_SIZE", "WRITE_SIZE"))
summary = {}
for k, v in traffic.items():
    f, w = v.get("FETCH_SIZE", [0.0]), v.get("WRITE_SIZE", [0.0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    summary[k] = {"FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm, "hbm_bytes_per_launch_corrected": 2 * fm * 1024 + wm * 1024, "launches": len(f)}
st_t = stage_totals(traffic)
for st, v in st_t.items():
    v["hbm_bytes_per_launch_corrected"] = 2 * v.get("FETCH_SIZE", 0.0) * 1024 + v.get("WRITE_SIZE", 0.0) * 1024
summary["_per_stage"] = st_t
summary["_note"] = ("_per_stage: per launch of the stage's primary kernel, summed over the kernels the stage launches (KB for FETCH_SIZE / "
                    "WRITE_SIZE; corrected bytes = 2 x FETCH + WRITE)")
json.dump(summary, open(out + "/pmc/traffic_summary.json", "w"), indent=1)
for k, v in sorted(st_t.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"]):
    print(f"{k:22s} launches {v['_launches']:5d}  HBM bytes / launch {v['hbm_bytes_per_launch_corrected'] / 1e6:9.2f} MB  (FETCH {v.get('FETCH_SIZE', 0):.0f} KB x2, WRITE {v.get('WRITE_SIZE', 0):.0f} KB)")

sq = per_kernel(glob.glob(out + "/pmc/**/sq_*_counter_collection.csv", recursive=True),
                ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                 "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
                 "SQ_THREAD_CYCLES_VALU"))
sqs = {k: {c: round(sum(x) / len(x)) for c, x in sorted(v.items())} for k, v in sq.items() if "so::" in k}
sqs["_per_stage"] = {st: {c: (round(x) if isinstance(x, float) else x) for c, x in sorted(v.items())} for st, v in stage_totals(sq).items()}
json.dump(sqs, open(out + "/pmc/sq_summary.json", "w"), indent=1)
for k, v in sqs["_per_stage"].items():
    if "SQ_INSTS_VALU" in v:
        print(f"{k:22s} VALU insts {v['SQ_INSTS_VALU']:>12.0f}  waves {v.get('SQ_WAVES', 0):>8.0f}  valu-active quad-cycles {v.get('SQ_ACTIVE_INST_VALU', 0):>12.0f}  "
              f"busy cycles {v.get('SQ_BUSY_CYCLES', 0):>12.0f}  wave cycles {v.get('SQ_WAVE_CYCLES', 0):>14.0f}")
