# round 5, call x: more counter copies on very small images? (R = 8 / 16 / 32 at 512 x 512 and 256 x 256)
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "256_20k_ref:--width 256 --height 256 --gaussians 20000 --regime ref" "768_200k:--width 768 --height 768 --gaussians 200000"; do
  name=${wl%%:*}; flags=${wl#*:}
  for R in 8 16 32 8; do
    export SPLAT_ONE_AMD_BIN_REPLICAS=$R
    timeout -k 10 300 python3 $B $flags > gpurun_out/x_${name}_$R.json 2> gpurun_out/x_${name}_$R.err || { echo "$name R=$R failed"; tail -3 gpurun_out/x_${name}_$R.err; continue; }
    python3 - gpurun_out/x_${name}_$R.json $name $R <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "R", sys.argv[3], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "pp_fwd", rk["so_preprocess_fwd"]["us"], "sort(+gather)", rk["so_isect_fill"]["us"], "void", j.get("void_steps"), "bins", j["config"].get("bin_capacity"))
PY
  done
done
