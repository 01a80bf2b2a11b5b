# round 5, call bd: final validation on the committed library -- the whole GPU suite, smoke, the default bench line, the driver's
# command line, and the two-rank gloo rehearsal of bench.py's N > 1 flow (last run before the tile tables were kept per view)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05bd
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; RC=$?; echo "pytest exit $RC"; tail -3 $OUT/pytest.txt
[ $RC -eq 0 ] || exit $RC
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 || { tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
cut -c1-300 $OUT/bench_default.json
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
cut -c1-300 $OUT/bench_driver.json
bash tools/gpu_bench_2ranks_gloo.sh > $OUT/gloo2.txt 2>&1; echo "gloo rehearsal exit $?"; grep "^rc=" $OUT/gloo2.txt
