cd $GRAFT_REPO_ROOT
for W in c2 c2d c2-ref; do WL=$W bash tools/gpu_profiles_r05.sh > gpurun_out/prof_$W.log 2>&1 || { tail -20 gpurun_out/prof_$W.log; exit 1; }; tail -25 gpurun_out/prof_$W.log; done
