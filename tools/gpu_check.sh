# quick correctness + bench pass:  bash tools/gpu_check.sh "<pytest -k expr>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -15
timeout 600 python bench.py --steps 200 --warmup 20 --kernel-table --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-230
