# round 5, call n3: k_preprocess_bwd 256 vs 128 threads, alternating on one box (box-to-box differences are ~3 %)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05n3
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "2M:--gaussians 2000000 --steps 60" "c4n:--gaussians 1000000 --width 2560 --height 1440 --steps 60" "c3:--gaussians 500000 --steps 100" "c5:--gaussians 2000000 --attr-dtype f16 --steps 60"; do
  name=${wl%%:*}; flags=${wl#*:}
  for b in 256 128 256 128; do
    SPLAT_ONE_AMD_PPB_BLOCK=$b timeout -k 10 200 python3 $B $flags > $OUT/${name}_$b.stdout 2> $OUT/${name}_$b.stderr
    python3 - $OUT/${name}_$b.stdout $name $b <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "no line"); sys.exit(0)
j = json.loads(l[-1])
print(sys.argv[2], "block", sys.argv[3], "it/s %.1f" % j["value"], "pp_bwd us", j["roofline_by_kernel"]["so_preprocess_bwd"]["us"], "frac", j["roofline_by_kernel"]["so_preprocess_bwd"]["frac"])
PY
  done
done
