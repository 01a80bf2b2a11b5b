# round 5, call as: lane-walked rectangles up to 36 tiles (12 at a time) in the binned path: product against variants own12 (the old limit) and own60
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_raster_op.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/as_pytest.txt 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/as_pytest.txt
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "c2:" "c2_s1.2:--scale-spread 1.2" "c2_s0.7:--scale-spread 0.7" "ref:--regime ref --steps 40" "1M_s1.0:--gaussians 1000000 --scale-spread 1.0 --steps 40" "960x540_1M_s1.0:--width 960 --height 540 --gaussians 1000000 --scale-spread 1.0" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "c4n:--gaussians 1000000 --width 2560 --height 1440 --steps 40"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in own12 product own60; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$LIB.so; [ -f $SPLAT_ONE_AMD_LIB ] || continue; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/as_${name}_$LIB.json 2> gpurun_out/as_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/as_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "pp_fwd", rk["so_preprocess_fwd"]["us"], "I", j["config"]["tile_intersections"])
PY
  done
done
unset SPLAT_ONE_AMD_LIB
