# Where the binning counters live (so_common.hpp bin_counter_index, SO_BIN_GROUP_LOG2): plain order / single counters spread /
# runs of 2, 4, 8 spread -- on the uniform headline scene, the dense `ref` regime (large splats: a wave bumps a block of
# neighbouring tiles) and the gathered clouds.  Variant libraries: tools/build_lib_variant.sh bin_<name> -DSO_BIN_GROUP_LOG2=<k>
cd $GRAFT_REPO_ROOT
for ARGS in "" "--regime ref" "--cloud-scale 0.2" "--cloud-scale 0.2 --n 2000000 --steps 50" "--cloud-scale 0.4 --n 400000"; do
  for V in plain g1 g2 g4 g8; do
    echo "=== $V  bench.py $ARGS"
    SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_bin_$V.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --kernel-table $ARGS 2> gpurun_out/skew_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline_by_kernel']
print(round(d['value'],1),'it/s  I',d['config']['tile_intersections'],' void',d.get('void_steps'),' '.join(f'{n[3:]} {v[\"us\"]}' for n,v in k.items()))" || exit 1
  done
done
