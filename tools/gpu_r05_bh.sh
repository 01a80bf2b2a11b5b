# round 5, call bh: the whole GPU suite and the driver's command line on the round's last commit (after the per-grid bins)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05bh
mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; RC=$?; echo "pytest exit $RC"; tail -3 $OUT/pytest.txt
[ $RC -eq 0 ] || exit $RC
timeout -k 10 100 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
cut -c1-300 $OUT/bench_driver.json
