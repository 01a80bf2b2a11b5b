# round 5, call y: one wave per tile (raster_impl 1) on images of FEW tiles -- 1024 tiles are one wave per SIMD
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "768_100k_ref:--width 768 --height 768 --gaussians 100000 --regime ref" "960x540_100k_ref:--width 960 --height 540 --gaussians 100000 --regime ref" "1440x720_100k_ref:--width 1440 --height 720 --gaussians 100000 --regime ref" "1080p_100k_ref:--regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  for NO in 0 1; do
    SPLAT_ONE_AMD_NO_TILE_WAVES=$NO timeout -k 10 300 python3 $B $flags > gpurun_out/y_${name}_$NO.json 2> gpurun_out/y_${name}_$NO.err || { echo "$name $NO failed"; tail -3 gpurun_out/y_${name}_$NO.err; continue; }
    python3 - gpurun_out/y_${name}_$NO.json $name $NO <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "no_tile_waves", sys.argv[3], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], j["config"].get("backward_rasteriser"), j["config"].get("tile_order"))
PY
  done
done
