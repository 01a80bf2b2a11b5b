# fused loss kernel: its parity test, the engine/trainer suites that run through it, and the step with 1 vs 2 loss kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "loss" > gpurun_out/fl_tests.log 2>&1 &&
timeout -k 10 700 python -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py tests/test_gpu_fuzz.py -x -q -m gpu >> gpurun_out/fl_tests.log 2>&1 &&
timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-table > gpurun_out/fl_bench1.json 2> gpurun_out/fl_bench1.err &&
timeout -k 10 200 python bench.py --no-cpu-baseline --kernel-table --loss-kernels 2 > gpurun_out/fl_bench2.json 2> gpurun_out/fl_bench2.err
echo rc=$?
tail -5 gpurun_out/fl_tests.log
