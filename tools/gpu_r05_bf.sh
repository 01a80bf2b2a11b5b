# round 5, call bf: seed 2344 of the rasterization() fuzzer failed its forward bar (7.3e-4) inside the run 2201..2600 and passes alone,
# in pairs and in the run 2300..2345 -- is it the history (a re-used buffer) or a race?  The same history twice.
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_FUZZ_BAR=1e-3
for i in 1 2; do
  timeout -k 10 200 python tools/dbg_fuzz_report.py $(seq 2201 2346) > gpurun_out/bf_$i.log 2>&1
  echo "run $i: ok $(grep -c ' ok ' gpurun_out/bf_$i.log)"; grep "FAIL\|ERROR" gpurun_out/bf_$i.log | cut -c1-300
done
echo done
