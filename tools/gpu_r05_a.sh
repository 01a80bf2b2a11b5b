# round 5, first GPU call: the new tests, the default bench (teacher targets), the training demo sweep
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_trainer.py -x -q -m gpu -k "overflow or mcmc" > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
timeout 600 python3 bench.py --kernel-table --no-other-configs > $OUT/bench.json 2> $OUT/bench_stderr.txt; echo "bench rc $?"
cut -c1-400 $OUT/bench.json
timeout 600 python3 bench.py --kernel-table --no-other-configs --no-cpu-baseline --no-operator-path --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver_stderr.txt; echo "bench rc $?"
cut -c1-400 $OUT/bench_driver.json
timeout 900 python3 tools/train_demo.py --sweep > $OUT/train_sweep.jsonl 2> $OUT/train_sweep_stderr.txt; echo "sweep rc $?"
cut -c1-300 $OUT/train_sweep.jsonl
timeout 900 python3 tools/train_demo.py > $OUT/train_demo.jsonl 2> $OUT/train_demo_stderr.txt; echo "demo rc $?"
cut -c1-300 $OUT/train_demo.jsonl
