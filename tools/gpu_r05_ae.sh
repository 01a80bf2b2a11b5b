# round 5, call ae: FEW Gaussians (the floor of a step): 10k at 1080p, c1 (10k at 256 x 256), 2k at 512 x 512
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 200"
for wl in "1080p_10k:--gaussians 10000" "c1_256_10k:--width 256 --height 256 --gaussians 10000" "c1_256_10k_ref:--width 256 --height 256 --gaussians 10000 --regime ref" "512_2k:--width 512 --height 512 --gaussians 2000"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/ae_$name.json 2> gpurun_out/ae_$name.err || { echo "$name failed"; tail -5 gpurun_out/ae_$name.err; continue; }
  python3 - gpurun_out/ae_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], "I", j["config"]["tile_intersections"], {k: v["us"] for k, v in rk.items()}, "sum", round(sum(v["us"] for v in rk.values()), 1))
PY
done
