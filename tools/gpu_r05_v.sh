# round 5, call v: small images, long lists (what the reference's default data_factor = 4 gives): where does k_preprocess_fwd stand?
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "1440x720_1M_ref03:--width 1440 --height 720 --gaussians 1000000 --regime ref --init-scale 0.3"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 300 python3 $B $flags > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err || { echo "$name failed"; tail -3 gpurun_out/v_$name.err; continue; }
  python3 - gpurun_out/v_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
I = j["config"]["tile_intersections"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", I, "V", j["config"]["visible_gaussians"], {k: v["us"] for k, v in rk.items()},
      "binning atomics G/s ~ %.1f" % (I / rk["so_preprocess_fwd"]["us"] / 1e3), j["config"].get("backward_rasteriser"), j["config"].get("binned_lists"), j["config"].get("bin_capacity"))
PY
done
