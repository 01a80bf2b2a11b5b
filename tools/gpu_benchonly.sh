cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 100 --warmup 10 --kernel-table --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-200
