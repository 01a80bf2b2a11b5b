# round 5, call ad: LARGE images -- 4K and a full-resolution panorama's tile count: does everything hold, and at what rates?
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 40 --warmup 10"
for wl in "4k_1M:--width 3840 --height 2160 --gaussians 1000000" "5760x2880_1M:--width 5760 --height 2880 --gaussians 1000000" "4k_100k_ref:--width 3840 --height 2160 --gaussians 100000 --regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/ad_$name.json 2> gpurun_out/ad_$name.err || { echo "$name failed"; tail -5 gpurun_out/ad_$name.err; continue; }
  python3 - gpurun_out/ad_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], {k: v["us"] for k, v in rk.items()}, j["config"].get("backward_rasteriser"), "fwd Mpix/s", j.get("forward_mpix_per_s"), "void", j.get("void_steps"), "hbm_iter", j.get("hbm_iter_fraction"))
PY
done
