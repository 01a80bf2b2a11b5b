# round 5, call aw: the fused loss kernel's compile-time shape once more on the final library: 256-thread workgroups, tap groups of 4 / 11
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "c4n:--gaussians 1000000 --width 2560 --height 1440 --steps 40" "512:--width 512 --height 512 --gaussians 60000"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in product ssim256 ssimtap4 ssimtap11 product; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$LIB.so; [ -f $SPLAT_ONE_AMD_LIB ] || continue; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/aw_${name}_$LIB.json 2> gpurun_out/aw_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/aw_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "ssim us", j["roofline_by_kernel"]["so_ssim_l1_fused"]["us"])
PY
  done
done
unset SPLAT_ONE_AMD_LIB
