# round 5, call ax: the backward rasteriser in list segments (SPLAT_ONE_AMD_BWD_SEGMENTS = 1 / 2 / 4 / 8): tests, then chain-bound regimes
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "segments" 2>&1 | tail -2
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512_60k_ref:--width 512 --height 512 --gaussians 60000 --regime ref" "skew02:--cloud-scale 0.2" "skew04_400k:--cloud-scale 0.4 --gaussians 400000" "c2:" "c2_s1.2:--scale-spread 1.2"; do
  name=${wl%%:*}; flags=${wl#*:}
  for S in 1 policy; do
    env $( [ $S = policy ] && echo XX=1 || echo SPLAT_ONE_AMD_BWD_SEGMENTS=$S ) timeout -k 10 300 python3 $B $flags > gpurun_out/ax_${name}_$S.json 2> gpurun_out/ax_${name}_$S.err || { echo "$name $S failed"; tail -3 gpurun_out/ax_${name}_$S.err; continue; }
    python3 - gpurun_out/ax_${name}_$S.json $name $S <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rk = j["roofline_by_kernel"]
print(sys.argv[2], "segments", sys.argv[3], "it/s %.1f" % j["value"], "rbwd", rk["so_rasterize_bwd"]["us"], "rfwd", rk["so_rasterize_fwd"]["us"], j["config"].get("backward_rasteriser")[:22])
PY
  done
done
