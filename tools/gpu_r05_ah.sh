# round 5, call ah: skewed long lists on a mid-size panorama (1440 x 720, 1M Gaussians, cameras inside the cloud): tile waves or quadrant waves?
cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-operator-path --no-other-configs --steps 100 --camera-model spherical --width 1440 --height 720 --gaussians 1000000"
for MODE in policy quadrant policy quadrant; do
  python3 - $MODE $ARGS > gpurun_out/ah_$MODE.json 2> gpurun_out/ah_$MODE.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "quadrant":
    lp.MIN_TILES_FOR_TILE_WAVES = 10 ** 9
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
  python3 - gpurun_out/ah_$MODE.json $MODE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], {k: v["us"] for k, v in rk.items()}, j["config"].get("backward_rasteriser"), j["config"].get("tile_order"))
PY
done
