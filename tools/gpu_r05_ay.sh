# round 5, call ay: the forward rasteriser with two list entries per trip (product) against one (variant -DSO_FWD_PAIRS=0)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_rasterization.py tests/test_gpu_raster_op.py -x -q -m gpu 2>&1 | tail -2
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
V=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_nofwdpairs.so
for wl in "c2:" "skew02:--cloud-scale 0.2" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "ref:--regime ref --steps 40" "c4n:--gaussians 1000000 --width 2560 --height 1440 --steps 40" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in variant product; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$V; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/ay_${name}_$LIB.json 2> gpurun_out/ay_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/ay_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "one entry per trip" if sys.argv[3] == "variant" else "two entries per trip", "it/s %.1f" % j["value"], "rfwd", rk["so_rasterize_fwd"]["us"], "fwd Mpix/s %.0f" % (j.get("forward_mpix_per_s") or 0))
PY
  done
done
unset SPLAT_ONE_AMD_LIB
