cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "it/s", "n_gpus": 1, "steps": 200, "warmup": 20, "ms_per_step": [0-9.]*'; done
