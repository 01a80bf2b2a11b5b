cd $GRAFT_REPO_ROOT
for W in c4 2M-f32 c5; do WL=$W bash tools/gpu_profiles_r05.sh > gpurun_out/prof_$W.log 2>&1 || { tail -20 gpurun_out/prof_$W.log; exit 1; }; tail -25 gpurun_out/prof_$W.log; done
