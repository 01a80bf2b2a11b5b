set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0)); import os; print(os.cpu_count())" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
