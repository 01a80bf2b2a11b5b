# round 5, call ai: the tile-count-scaled unevenness rule for one wave per tile: which kernel does each regime get now, and at what rate?
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 60"
for wl in "sph_1440x720_1M:--camera-model spherical --width 1440 --height 720 --gaussians 1000000" "1440x720_100k_ref:--width 1440 --height 720 --gaussians 100000 --regime ref" "c2_ref:--regime ref" "c4:--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100" "sph_1920x960_ref:--camera-model spherical --width 1920 --height 960 --regime ref" "skew02:--cloud-scale 0.2" "2M:--gaussians 2000000"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/ai_$name.json 2> gpurun_out/ai_$name.err || { echo "$name failed"; tail -5 gpurun_out/ai_$name.err; continue; }
  python3 - gpurun_out/ai_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], j["config"].get("backward_rasteriser"), "|", j["config"].get("tile_order"), "void", j.get("void_steps"))
PY
done
