# round 5, call af: what do the idle workgroups of the capacity-sized grids cost?  c2 with capacities 2^20 (default) / 2^18 / 2^17 / 100 352
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for cap in 1048576 262144 131072 100352 1048576 100352; do
  timeout -k 10 300 python3 $B --max-gaussians $cap > gpurun_out/af_$cap.json 2> gpurun_out/af_$cap.err || { echo "$cap failed"; tail -3 gpurun_out/af_$cap.err; continue; }
  python3 - gpurun_out/af_$cap.json $cap <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print("capacity", sys.argv[2], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], {k: v["us"] for k, v in rk.items()})
PY
done
