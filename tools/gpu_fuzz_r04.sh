# Seeds 100-500 of both seeded fuzzers (tests/test_gpu_fuzz.py: rasterization() and the fused engine against the oracle on the
# device's discrete decisions, 1e-3 everywhere) on the final round-4 library; the engine cases also with the one-wave-per-tile
# backward forced (SPLAT_ONE_AMD_FUZZ_IMPL=1 -> so_step_desc.raster_impl = 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 900 python tools/dbg_fuzz_report.py $(seq 100 500) > gpurun_out/fuzz_r04_operator.log 2>&1
timeout -k 10 900 python tools/dbg_fuzz_report.py --engine $(seq 100 500) > gpurun_out/fuzz_r04_engine.log 2>&1
grep -c " ok " gpurun_out/fuzz_r04_operator.log gpurun_out/fuzz_r04_engine.log
grep "FAIL\|ERROR" gpurun_out/fuzz_r04_operator.log gpurun_out/fuzz_r04_engine.log | cut -c1-400
