# round 5, call o: the whole GPU suite, smoke, the default bench line and the driver's command line on the current library
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05o
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-operator-path > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-300 $OUT/bench_driver.json
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-operator-path | cut -c1-200
