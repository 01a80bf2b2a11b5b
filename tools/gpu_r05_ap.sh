# round 5, call ap: the default tile order with the tile rows taken centre-out (variant build, -DSO_TILE_ROWS_CENTRE_OUT=1)
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
V=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_centreout.so
for wl in "c2:" "c2_noise:--targets noise" "spherical:--camera-model spherical --width 1920 --height 960" "c2:" ; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in product centreout; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$V; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/ap_${name}_$LIB.json 2> gpurun_out/ap_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/ap_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"])
PY
  done
done
unset SPLAT_ONE_AMD_LIB
