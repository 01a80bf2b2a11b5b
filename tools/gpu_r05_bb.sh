# round 5, call bb: kept tables where the per-step table already paid (tile waves, long lists): SPLAT_ONE_AMD_ORDER_CACHE=0/1 on one box
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c4:--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100" "2M:--gaussians 2000000 --steps 60" "c4:--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100"; do
  name=${wl%%:*}; flags=${wl#*:}
  for CACHE in 0 1; do
    SPLAT_ONE_AMD_ORDER_CACHE=$CACHE timeout -k 10 300 python3 $B $flags > gpurun_out/bb_${name}_$CACHE.json 2> gpurun_out/bb_${name}_$CACHE.err || { echo "$name $CACHE failed"; continue; }
    python3 - gpurun_out/bb_${name}_$CACHE.json $name $CACHE <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "kept tables", sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"])
PY
  done
done
