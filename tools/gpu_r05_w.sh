# round 5, call w: replicated bin counters (SPLAT_ONE_AMD_BIN_REPLICAS = 1 / 4 / 8 / auto) on images of few tiles, then the whole GPU suite
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "512_60k_mcmc:--width 512 --height 512 --gaussians 60000" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "1440x720_1M:--width 1440 --height 720 --gaussians 1000000" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for R in 1 8 4 auto; do
    if [ $R = auto ]; then unset SPLAT_ONE_AMD_BIN_REPLICAS; else export SPLAT_ONE_AMD_BIN_REPLICAS=$R; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/w_${name}_$R.json 2> gpurun_out/w_${name}_$R.err || { echo "$name R=$R failed"; tail -3 gpurun_out/w_${name}_$R.err; continue; }
    python3 - gpurun_out/w_${name}_$R.json $name $R <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "R", sys.argv[3], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "pp_fwd", rk["so_preprocess_fwd"]["us"], "sort(+gather)", rk["so_isect_fill"]["us"], "void", j.get("void_steps"))
PY
  done
done
unset SPLAT_ONE_AMD_BIN_REPLICAS
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/w_pytest.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/w_pytest.txt
