# round 5, call ab: 8x8 tiles on small images (one wave per tile, four times the tiles, shorter lists)?
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "512_60k:--width 512 --height 512 --gaussians 60000" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "960x540_100k_ref:--width 960 --height 540 --gaussians 100000 --regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  for TS in 16 8; do
    SPLAT_ONE_AMD_TILE_SIZE=$TS timeout -k 10 300 python3 $B $flags > gpurun_out/ab_${name}_$TS.json 2> gpurun_out/ab_${name}_$TS.err || { echo "$name $TS failed"; tail -3 gpurun_out/ab_${name}_$TS.err; continue; }
    python3 - gpurun_out/ab_${name}_$TS.json $name $TS <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "tile", sys.argv[3], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], {k: v["us"] for k, v in rk.items()})
PY
  done
done
