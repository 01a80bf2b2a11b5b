# fixed-state stage times with SPLAT_ONE_AMD_BWD_TILE = 0 / 1 (wave-per-quadrant vs wave-per-tile backward rasteriser)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/bwd_ab.jsonl
for T in 0 1; do
  export SPLAT_ONE_AMD_BWD_TILE=$T
  for CFG in "100000 1920 1080 mcmc" "1000000 2560 1440 mcmc" "100000 1920 1080 ref" "500000 1920 1080 mcmc"; do
    echo "{\"tile_waves\": $T}" >> gpurun_out/bwd_ab.jsonl
    timeout -k 10 200 python tools/dbg_bwd_fixed.py $CFG >> gpurun_out/bwd_ab.jsonl 2> gpurun_out/bwd_ab.err || exit 1
  done
done
cat gpurun_out/bwd_ab.jsonl
