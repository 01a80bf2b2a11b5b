# history stress of the rasterization() fuzzer after the per-grid bins: 400 old seeds in a shuffled order (another history for every case)
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
SEEDS=$(python3 -c "import random; s=list(range(2201,2801)); random.Random(5).shuffle(s); print(' '.join(map(str,s[:400])))")
timeout -k 10 150 python tools/dbg_fuzz_report.py $SEEDS > gpurun_out/fuzz_r05g_operator_shuffled.log 2>&1; echo "operator rc $?"
grep -c " ok " gpurun_out/fuzz_r05g_operator_shuffled.log || true
grep "FAIL\|ERROR" gpurun_out/fuzz_r05g_operator_shuffled.log | cut -c1-400 || true
