set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 200 --warmup 20 --kernel-table "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -30 gpurun_out/bench.err
cat gpurun_out/bench.json
