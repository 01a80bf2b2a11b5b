# round 5, call p: binned counters bumped in pairs by 64-bit atomics (SPLAT_ONE_AMD_PP_PAIRS=0/1): tests, then A/B per regime
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05p
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "engine or cabi or configs or spher or raster or step or trains" > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.txt
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "ref:--regime ref --steps 60" "c4n:--gaussians 1000000 --width 2560 --height 1440 --steps 60" "2M:--gaussians 2000000 --steps 60" "skew:--cloud-scale 0.2"; do
  name=${wl%%:*}; flags=${wl#*:}
  for b in 0 1 0 1; do
    SPLAT_ONE_AMD_PP_PAIRS=$b timeout -k 10 200 python3 $B $flags > $OUT/${name}_$b.stdout 2> $OUT/${name}_$b.stderr
    python3 - $OUT/${name}_$b.stdout $name $b <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "no line"); sys.exit(0)
j = json.loads(l[-1])
print(sys.argv[2], "pairs", sys.argv[3], "it/s %.1f" % j["value"], "pp_fwd us", j["roofline_by_kernel"]["so_preprocess_fwd"]["us"], "I", j["config"]["tile_intersections"])
PY
  done
done
