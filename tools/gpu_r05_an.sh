# round 5, call an: where is the crossover between quadrant waves and tile waves by MEAN list length, on large tile counts?
cd $GRAFT_REPO_ROOT
for wl in "1M_1440p:--gaussians 1000000 --width 2560 --height 1440 --steps 60" "c3_500k:--gaussians 500000 --steps 100" "1M_1080p:--gaussians 1000000 --steps 60" "2M_1440p:--gaussians 2000000 --width 2560 --height 1440 --steps 40" "4k_1M:--gaussians 1000000 --width 3840 --height 2160 --steps 40"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MODE in policy tile; do
    python3 - $MODE --no-cpu-baseline --no-operator-path --no-other-configs $flags > gpurun_out/an_${name}_$MODE.json 2> gpurun_out/an_${name}_$MODE.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "tile":
    lp.pick_raster_impl = lambda now, mean_list, fullest, tile16, absgrad, first=False, n_tiles=1 << 30: 1
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
    python3 - gpurun_out/an_${name}_$MODE.json $name $MODE <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); sys.exit(0)
rk = j["roofline_by_kernel"]
import re
m = re.search(r"(\d+)x(\d+)", j["config"]["workload"]); M = ((int(m.group(1)) + 15) // 16) * ((int(m.group(2)) + 15) // 16)
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "mean list %.0f" % (j["config"]["tile_intersections"] / M), "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], "|", j["config"].get("backward_rasteriser"))
PY
  done
done
