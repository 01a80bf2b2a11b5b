# round 5, call bc: the backward rasteriser's staging batch (SO_BWD_STAGE = 256 product / 128 / 64) now that c2 runs longest-first
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "c3:--gaussians 500000 --steps 100" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in product bwdstage128 bwdstage64; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$LIB.so; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/bc_${name}_$LIB.json 2> gpurun_out/bc_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/bc_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "rbwd", j["roofline_by_kernel"]["so_rasterize_bwd"]["us"])
PY
  done
done
unset SPLAT_ONE_AMD_LIB
