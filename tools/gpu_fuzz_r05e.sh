# NEW seeds (2601-3200) of both seeded fuzzers after the per-grid bins (uniform 1e-3 bar)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 300 python tools/dbg_fuzz_report.py $(seq 2601 3200) > gpurun_out/fuzz_r05f_operator.log 2>&1; echo "operator rc $?"
timeout -k 10 400 python tools/dbg_fuzz_report.py --engine $(seq 2601 3200) > gpurun_out/fuzz_r05f_engine.log 2>&1; echo "engine rc $?"
grep -c " ok " gpurun_out/fuzz_r05f_operator.log gpurun_out/fuzz_r05f_engine.log || true
grep "FAIL\|ERROR" gpurun_out/fuzz_r05f_operator.log gpurun_out/fuzz_r05f_engine.log | cut -c1-500 || true
