# round 5, call ak: longest-list-first for the QUADRANT-wave kernels at moderate unevenness (policy: only beyond 8x the mean)
cd $GRAFT_REPO_ROOT
for wl in "sph_1440x720_1M:--camera-model spherical --width 1440 --height 720 --gaussians 1000000" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "c2:" "c3:--gaussians 500000"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MODE in policy lpt; do
    python3 - $MODE --no-cpu-baseline --no-operator-path --no-other-configs --steps 100 $flags > gpurun_out/ak_${name}_$MODE.json 2> gpurun_out/ak_${name}_$MODE.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "lpt":
    lp.pick_tile_order = lambda now, impl, mean_list, fullest: True
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
    python3 - gpurun_out/ak_${name}_$MODE.json $name $MODE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], "order", rk.get("so_tile_order", {}).get("us"), j["config"].get("backward_rasteriser"), "|", j["config"].get("tile_order"))
PY
  done
done
