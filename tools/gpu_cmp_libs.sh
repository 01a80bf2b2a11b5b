cd $GRAFT_REPO_ROOT
for L in "" build/variants/libsplat_one_amd_order2.so build/variants/libsplat_one_amd_order0.so "" build/variants/libsplat_one_amd_order2.so; do
  if [ -n "$L" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/$L; else unset SPLAT_ONE_AMD_LIB; fi
  echo "=== lib ${L:-product}"
  timeout 600 python3 bench.py --no-cpu-baseline --kernel-table --steps 100 2> gpurun_out/cmp_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline_by_kernel']
print(round(d['value'],1),'it/s  I',d['config']['tile_intersections'],' fwd',k['so_rasterize_fwd']['us'],' bwd',k['so_rasterize_bwd']['us'])"
done
