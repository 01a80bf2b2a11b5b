# tile -> XCD order of the rasteriser kernels (rasterize_common.hpp SO_TILE_ORDER 0 / 1 / 2), on the uniform c2 scene and
# on scenes whose splats gather in the middle of the image (bench.py --cloud-scale)
cd $GRAFT_REPO_ROOT
for ARGS in "" "--cloud-scale 0.4" "--cloud-scale 0.4 --n 400000" "--regime ref --cloud-scale 0.5"; do
  for ORD in 0 1 2; do
    echo "=== order $ORD  bench.py $ARGS"
    SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_order$ORD.so timeout 600 python3 bench.py --no-cpu-baseline --kernel-table --steps 100 $ARGS 2> gpurun_out/tileorder_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline_by_kernel']
print(round(d['value'],1),'it/s  I',d['config']['tile_intersections'],' fwd',k['so_rasterize_fwd']['us'],' bwd',k['so_rasterize_bwd']['us'],' sort',k['so_isect_fill']['us'],' ppfwd',k['so_preprocess_fwd']['us'])"
  done
done
