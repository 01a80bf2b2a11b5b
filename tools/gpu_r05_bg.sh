# round 5, call bg: bins of rasterization() kept per grid (C, tile_w, tile_h) instead of per tile count -- the regression test, the
# operator-path test files, and the fuzz run that found it (2201..2600) again
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_raster_op.py tests/test_gpu_rasterization.py -x -q -m gpu > gpurun_out/bg_pytest.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/bg_pytest.txt
timeout -k 10 400 python tools/dbg_fuzz_report.py $(seq 2201 2600) > gpurun_out/fuzz_r05e_operator.log 2>&1; echo "operator rc $?"
grep -c " ok " gpurun_out/fuzz_r05e_operator.log || true
grep "FAIL\|ERROR" gpurun_out/fuzz_r05e_operator.log | cut -c1-400 || true
