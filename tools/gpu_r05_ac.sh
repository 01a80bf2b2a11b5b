# round 5, call ac: replicated bin counters on the OPERATOR path (so_rasterization_fwd): tests, then R = 1 against auto
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "raster_op or rasterization or fuzz or packed or threads or runner_io or trains" > gpurun_out/ac_pytest.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/ac_pytest.txt
B="bench.py --operator-path --no-cpu-baseline --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "c2_ref:--regime ref --steps 40" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for R in auto; do
    if [ $R = auto ]; then unset SPLAT_ONE_AMD_BIN_REPLICAS; else export SPLAT_ONE_AMD_BIN_REPLICAS=$R; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/ac_${name}_$R.json 2> gpurun_out/ac_${name}_$R.err || { echo "$name $R failed"; tail -3 gpurun_out/ac_${name}_$R.err; continue; }
    python3 - gpurun_out/ac_${name}_$R.json $name $R <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "R", sys.argv[3], "operator path it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"])
PY
  done
done
