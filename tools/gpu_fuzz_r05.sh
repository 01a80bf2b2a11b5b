# seeds 100-700 of both seeded fuzzers on the round-5 library (uniform 1e-3 bar), then the soak (18 configurations x 800 iterations)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 500 python tools/dbg_fuzz_report.py $(seq 100 700) > gpurun_out/fuzz_r05_operator.log 2>&1
timeout -k 10 500 python tools/dbg_fuzz_report.py --engine $(seq 100 700) > gpurun_out/fuzz_r05_engine.log 2>&1
grep -c " ok " gpurun_out/fuzz_r05_operator.log gpurun_out/fuzz_r05_engine.log
grep "FAIL\|ERROR" gpurun_out/fuzz_r05_operator.log gpurun_out/fuzz_r05_engine.log | cut -c1-400
timeout -k 10 900 python tools/dbg_soak.py 800 > gpurun_out/soak_r05.log 2>&1; echo "soak rc $?"
tail -22 gpurun_out/soak_r05.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "tile_order or prologue" 2>&1 | tail -3
