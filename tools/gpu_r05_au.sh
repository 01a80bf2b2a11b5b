# round 5, call au: counter copies where one image region is hot (policy), the new stage accounting, engine tests
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_raster_op.py -x -q -m gpu > gpurun_out/au_pytest.txt 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/au_pytest.txt
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "skew02:--cloud-scale 0.2" "skew04_400k:--cloud-scale 0.4 --gaussians 400000" "c2:" "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 300 python3 $B $flags > gpurun_out/au_$name.json 2> gpurun_out/au_$name.err || { echo "$name failed"; tail -3 gpurun_out/au_$name.err; continue; }
  python3 - gpurun_out/au_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], {k: v["us"] for k, v in rk.items()}, "bins", j["config"].get("bin_capacity"), "void", j.get("void_steps"))
PY
done
