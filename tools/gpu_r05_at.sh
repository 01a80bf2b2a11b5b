# round 5, call at: counter copies at 1080p when the lists are long / uneven (hot counters): R = 1 / 4 / 8
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "c2_s1.2:--scale-spread 1.2" "1M_s1.0:--gaussians 1000000 --scale-spread 1.0 --steps 40" "ref:--regime ref --steps 40" "2M:--gaussians 2000000 --steps 40" "skew02:--cloud-scale 0.2"; do
  name=${wl%%:*}; flags=${wl#*:}
  for R in 1 4 8; do
    SPLAT_ONE_AMD_BIN_REPLICAS=$R timeout -k 10 300 python3 $B $flags > gpurun_out/at_${name}_$R.json 2> gpurun_out/at_${name}_$R.err || { echo "$name $R failed"; continue; }
    python3 - gpurun_out/at_${name}_$R.json $name $R <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "R", sys.argv[3], "it/s %.1f" % j["value"], "pp_fwd", rk["so_preprocess_fwd"]["us"], "sort(+gather)", rk["so_isect_fill"]["us"], "bins", j["config"].get("bin_capacity"), "void", j.get("void_steps"))
PY
  done
done
