# round 5, call ao: the total-work rule for one wave per tile: named workloads + the whole GPU suite
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "ref:--regime ref --steps 60" "c3:--gaussians 500000 --steps 100" "c4:--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100" "2M:--gaussians 2000000 --steps 60" "c5:--gaussians 2000000 --attr-dtype f16 --steps 60" "1M_1440p:--gaussians 1000000 --width 2560 --height 1440 --steps 60" "4k_1M:--gaussians 1000000 --width 3840 --height 2160 --steps 40" "1M_1080p:--gaussians 1000000 --steps 60"; do
  name=${wl%%:*}; flags=${wl#*:}
  timeout -k 10 400 python3 $B $flags > gpurun_out/ao_$name.json 2> gpurun_out/ao_$name.err || { echo "$name failed"; tail -5 gpurun_out/ao_$name.err; continue; }
  python3 - gpurun_out/ao_$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "it/s %.1f" % j["value"], "I", j["config"]["tile_intersections"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], j["config"].get("backward_rasteriser"), "hbm_iter", round(j.get("hbm_iter_fraction") or 0, 4), "void", j.get("void_steps"))
PY
done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/ao_pytest.txt 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/ao_pytest.txt
