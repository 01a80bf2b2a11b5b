# round 5, call az: tile tables kept per view (SPLAT_ONE_AMD_ORDER_CACHE=0/1): tests, then the named workloads
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -2
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "c2d:--steps 20 --warmup 5" "c3:--gaussians 500000 --steps 100" "ref:--regime ref --steps 60" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref --steps 100" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for CACHE in 0 1; do
    SPLAT_ONE_AMD_ORDER_CACHE=$CACHE timeout -k 10 300 python3 $B $flags > gpurun_out/az_${name}_$CACHE.json 2> gpurun_out/az_${name}_$CACHE.err || { echo "$name $CACHE failed"; tail -3 gpurun_out/az_${name}_$CACHE.err; continue; }
    python3 - gpurun_out/az_${name}_$CACHE.json $name $CACHE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "kept tables", sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], "order", rk.get("so_tile_order", {}).get("us"), "|", (j["config"].get("tile_order") or "")[:24])
PY
  done
done
