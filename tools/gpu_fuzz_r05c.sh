# NEW seeds (1301-1900) of both seeded fuzzers on the round's final library (uniform 1e-3 bar), then the soak
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 500 python tools/dbg_fuzz_report.py $(seq 1301 1900) > gpurun_out/fuzz_r05c_operator.log 2>&1
timeout -k 10 500 python tools/dbg_fuzz_report.py --engine $(seq 1301 1900) > gpurun_out/fuzz_r05c_engine.log 2>&1
grep -c " ok " gpurun_out/fuzz_r05c_operator.log gpurun_out/fuzz_r05c_engine.log || true
grep "FAIL\|ERROR" gpurun_out/fuzz_r05c_operator.log gpurun_out/fuzz_r05c_engine.log | cut -c1-500 || true
timeout -k 10 900 python tools/dbg_soak.py 800 > gpurun_out/soak_r05c.log 2>&1; echo "soak rc $?"
tail -22 gpurun_out/soak_r05c.log | cut -c1-200
