cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 50 --warmup 10 --kernel-table --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-400
