# round 5, call al: k_tile_order with its counter loads batched: what does the table cost now, and does c2 gain from longest-first?
cd $GRAFT_REPO_ROOT
for wl in "c2:" "c3:--gaussians 500000" "ref:--regime ref --steps 60" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MODE in policy lpt; do
    python3 - $MODE --no-cpu-baseline --no-operator-path --no-other-configs $flags > gpurun_out/al_${name}_$MODE.json 2> gpurun_out/al_${name}_$MODE.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "lpt":
    lp.pick_tile_order = lambda now, impl, mean_list, fullest: True
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
    python3 - gpurun_out/al_${name}_$MODE.json $name $MODE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], "order", rk.get("so_tile_order", {}).get("us"), "|", j["config"].get("tile_order"))
PY
  done
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "tile_order" 2>&1 | tail -2
