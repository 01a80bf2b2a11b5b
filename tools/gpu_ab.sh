cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for impl in 0 1; do
  echo "=== raster-impl $impl"
  timeout 600 python bench.py --steps 200 --warmup 20 --kernel-table --no-cpu-baseline --raster-impl $impl 2>&1 | grep -v amdgpu.ids | cut -c1-160
done
