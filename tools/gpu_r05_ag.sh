# round 5, call ag: the same step through the other camera models (the reference's default is spherical)
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "pinhole_c2:" "fisheye_c2:--camera-model fisheye" "spherical_1920x960_100k:--camera-model spherical --width 1920 --height 960" "spherical_1920x960_100k_ref:--camera-model spherical --width 1920 --height 960 --regime ref" "spherical_1440x720_1M:--camera-model spherical --width 1440 --height 720 --gaussians 1000000"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/ag_$name.json 2> gpurun_out/ag_$name.err || { echo "$name failed"; tail -5 gpurun_out/ag_$name.err; continue; }
  python3 - gpurun_out/ag_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "V", j["config"]["visible_gaussians"], {k: v["us"] for k, v in rk.items()}, j["config"].get("backward_rasteriser"), "void", j.get("void_steps"), j["config"].get("binned_lists"), j["config"].get("bin_capacity"))
PY
done
