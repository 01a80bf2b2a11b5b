# round 5, call aj: tile waves (policy) against quadrant waves (forced) where the policy picks tile waves: 2M, a 1920 x 960 panorama with dense lists, c4's size without densification
cd $GRAFT_REPO_ROOT
for wl in "2M:--gaussians 2000000" "sph_1920x960_ref:--camera-model spherical --width 1920 --height 960 --regime ref" "1M_1440p:--gaussians 1000000 --width 2560 --height 1440"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MODE in policy quadrant; do
    python3 - $MODE --no-cpu-baseline --no-operator-path --no-other-configs --steps 60 $flags > gpurun_out/aj_${name}_$MODE.json 2> gpurun_out/aj_${name}_$MODE.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "quadrant":
    lp.MIN_TILES_FOR_TILE_WAVES = 10 ** 9
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
    python3 - gpurun_out/aj_${name}_$MODE.json $name $MODE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], j["config"].get("backward_rasteriser"), "bins", j["config"].get("bin_capacity"))
PY
  done
done
