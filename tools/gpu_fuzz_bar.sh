# What a uniform 1e-3 gradient bar would flag in the seeded fuzz tests (fisheye 3e-3 / spherical 5e-3).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
timeout -k 10 500 python tools/dbg_fuzz_report.py $(seq 0 119) > gpurun_out/fuzz_bar_operator.log 2>&1
timeout -k 10 500 python tools/dbg_fuzz_report.py --engine $(seq 0 119) > gpurun_out/fuzz_bar_engine.log 2>&1
grep -c " ok " gpurun_out/fuzz_bar_operator.log gpurun_out/fuzz_bar_engine.log
grep "FAIL\|ERROR" gpurun_out/fuzz_bar_operator.log gpurun_out/fuzz_bar_engine.log | cut -c1-400
