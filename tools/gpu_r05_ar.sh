# round 5, call ar: the big-rectangle binning path with four 8x8 tile blocks in flight (product) against one (variant -DSO_PP_BIG_PIPELINE=0)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_raster_op.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/ar_pytest.txt 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/ar_pytest.txt
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
V=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_nobigpipe.so
for wl in "c2:" "c2_s1.2:--scale-spread 1.2" "c2_s0.7:--scale-spread 0.7" "ref:--regime ref --steps 40" "1M_s1.0:--gaussians 1000000 --scale-spread 1.0 --steps 40" "960x540_1M_s1.0:--width 960 --height 540 --gaussians 1000000 --scale-spread 1.0" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in variant product; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$V; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/ar_${name}_$LIB.json 2> gpurun_out/ar_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/ar_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "one block in flight" if sys.argv[3] == "variant" else "four blocks in flight", "it/s %.1f" % j["value"], "pp_fwd", rk["so_preprocess_fwd"]["us"], "I", j["config"]["tile_intersections"])
PY
  done
done
unset SPLAT_ONE_AMD_LIB
