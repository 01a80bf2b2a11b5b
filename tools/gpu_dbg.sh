cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for args in "100000 1920 1080 graph"; do
  echo "=== $args"
  timeout 120 python tools/dbg_engine.py $args > gpurun_out/dbg.log 2>&1
  tail -12 gpurun_out/dbg.log
  if grep -q "Memory access fault" gpurun_out/dbg.log; then echo FAULT; exit 1; fi
done
