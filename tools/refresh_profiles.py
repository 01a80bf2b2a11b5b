"""Copies the summaries of the last gpurun profile calls into profiles/ (tracked) and regenerates
profiles/traffic.json (entry point -> corrected HBM bytes per launch, read by bench.py)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
pairs = [("prof/bench_kernel_stats.csv", f"{tag}_engine_kernel_stats.csv"),
         ("pmc/traffic_summary.json", f"{tag}_engine_pmc_traffic.json"),
         ("bench.json", f"{tag}_engine_bench.json")]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, b))
        print("copied", a, "->", b)
d = json.load(open(os.path.join(dst, f"{tag}_engine_pmc_traffic.json")))
names = {"so_rasterize_bwd": "void so::k_rasterize_bwd<3, 16, false, true>", "so_rasterize_fwd": "void so::k_rasterize_fwd<3, 16, true>",
         "so_adam_step_dev": "so::k_adam_dev", "so_ssim_l1_fwd": "void so::k_ssim_l1_fwd<3>", "so_ssim_l1_bwd": "void so::k_ssim_l1_bwd<3>",
         "so_preprocess_fwd": "void so::k_preprocess_fwd<3, so::AttrSoA, false>",
         "so_preprocess_bwd": "void so::k_preprocess_bwd<3, so::AttrSoA, true, true, false>"}
out = {k: d[v]["hbm_bytes_per_launch_corrected"] for k, v in names.items() if v in d}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_pmc.sh), KB -> bytes, FETCH_SIZE doubled as "
                "MI355X_MICROARCH.md prescribes for gfx950; source profiles/%s_engine_pmc_traffic.json" % tag)
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(out)
