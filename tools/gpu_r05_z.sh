# round 5, call z: after the tile-count rule for one wave per tile + replicated counters: small-image lines, the long schedule, the whole GPU suite
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "960x540_1M:--width 960 --height 540 --gaussians 1000000" "960x540_100k_ref:--width 960 --height 540 --gaussians 100000 --regime ref" "c2:" "ref:--regime ref"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 300 python3 $B $flags > gpurun_out/z_$name.json 2> gpurun_out/z_$name.err || { echo "$name failed"; tail -3 gpurun_out/z_$name.err; continue; }
  python3 - gpurun_out/z_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], {k: v["us"] for k, v in rk.items()}, j["config"].get("backward_rasteriser"))
PY
done
timeout -k 10 600 python3 tools/train_demo.py --long --res 512 --teacher-n 50000 --student-n 50000 --train-views 32 > gpurun_out/z_train_long.jsonl 2> gpurun_out/z_train_long.err
python3 - <<'PY'
import json
for l in open("gpurun_out/z_train_long.jsonl"):
    j = json.loads(l); print(j["strategy"], j["n_final"], round(j["psnr_heldout"], 2), j["wall_seconds"], j["void_steps"], [(b["steps"], b["gaussians"], b["it_s"]) for b in j["blocks"]])
PY
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/z_pytest.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/z_pytest.txt
