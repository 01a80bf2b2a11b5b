# fixed-state stage times: product vs variant libraries (build/variants/libsplat_one_amd_NAME.so); args: variant names
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/bwd_ablate.jsonl
for V in "" "$@"; do
  if [ -n "$V" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$V.so; else unset SPLAT_ONE_AMD_LIB; fi
  for CFG in "100000 1920 1080 mcmc" "1000000 2560 1440 mcmc" "100000 1920 1080 ref"; do
    timeout -k 10 200 python tools/dbg_bwd_fixed.py $CFG >> gpurun_out/bwd_ablate.jsonl 2> gpurun_out/bwd_ablate.err || exit 1
  done
done
cat gpurun_out/bwd_ablate.jsonl
