# round 5, call am: the tile table built beside the sort on a forked stream (SPLAT_ONE_AMD_ORDER_FORK=0/1): tests, then c2 / c3 / ref / small images
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py -x -q -m gpu > gpurun_out/am_pytest.txt 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/am_pytest.txt
for wl in "c2:" "c3:--gaussians 500000" "ref:--regime ref --steps 60" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MODE in "policy 1" "lpt 0" "lpt 1"; do
    set -- $MODE
    SPLAT_ONE_AMD_ORDER_FORK=$2 python3 - $1 --no-cpu-baseline --no-operator-path --no-other-configs $flags > gpurun_out/am_${name}_$1_$2.json 2> gpurun_out/am_${name}_$1_$2.err <<'PY'
import runpy, sys
mode = sys.argv[1]
import splat_one_amd.list_policy as lp
if mode == "lpt":
    lp.pick_tile_order = lambda now, impl, mean_list, fullest: True
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
    python3 - gpurun_out/am_${name}_$1_$2.json $name "$MODE" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); sys.exit(0)
rk = j["roofline_by_kernel"]
print(sys.argv[2], "(order, fork) =", sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], "|", j["config"].get("tile_order")[:20])
PY
  done
done
