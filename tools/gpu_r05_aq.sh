# round 5, call aq: anisotropic splats of very different sizes (a trained scene's statistics): --scale-spread
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "c2_s0.7:--scale-spread 0.7" "c2_s1.2:--scale-spread 1.2" "1M_s1.0:--gaussians 1000000 --scale-spread 1.0" "960x540_1M_s1.0:--width 960 --height 540 --gaussians 1000000 --scale-spread 1.0" "ref_s0.7:--regime ref --scale-spread 0.7 --steps 40"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/aq_$name.json 2> gpurun_out/aq_$name.err || { echo "$name failed"; tail -5 gpurun_out/aq_$name.err; continue; }
  python3 - gpurun_out/aq_$name.json $name <<'PY'
import json, sys, re
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
c = j["config"]
m = re.search(r"(\d+)x(\d+)", c["workload"]); M = ((int(m.group(1)) + 15) // 16) * ((int(m.group(2)) + 15) // 16)
print(sys.argv[2], "it/s %.1f" % j["value"], "I", c["tile_intersections"], "mean %.0f" % (c["tile_intersections"] / M), "bins", c.get("bin_capacity"), "binned", c.get("binned_lists"), {k: v["us"] for k, v in rk.items()}, c.get("backward_rasteriser"), "|", (c.get("tile_order") or "")[:12], "void", j.get("void_steps"))
PY
  grep -i "warn" gpurun_out/aq_$name.err | head -2 | cut -c1-200
done
