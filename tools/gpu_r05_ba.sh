# round 5, call ba: how often a kept tile table is rebuilt (SPLAT_ONE_AMD_ORDER_REFRESH = 4 / 8 / 16 / 32 / 64), c2 and c3
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 400"
for wl in "c2:" "c3:--gaussians 500000 --steps 200"; do
  name=${wl%%:*}; flags=${wl#*:}
  for R in 4 8 16 32 64; do
    SPLAT_ONE_AMD_ORDER_REFRESH=$R timeout -k 10 300 python3 $B $flags > gpurun_out/ba_${name}_$R.json 2> gpurun_out/ba_${name}_$R.err || { echo "$name $R failed"; continue; }
    python3 - gpurun_out/ba_${name}_$R.json $name $R <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "rebuilt every", sys.argv[3], "visits: it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"])
PY
  done
done
