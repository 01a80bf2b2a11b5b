# round 5, call aa: the fused loss kernel on small images -- fewer rows per workgroup (more workgroups, more halo)
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100"
for wl in "512:--width 512 --height 512 --gaussians 60000" "960x540:--width 960 --height 540 --gaussians 100000" "256:--width 256 --height 256 --gaussians 20000" "1440x720:--width 1440 --height 720 --gaussians 100000" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for MR in 24 16 12 8 4; do
    SPLAT_ONE_AMD_SSIM_MIN_ROWS=$MR timeout -k 10 300 python3 $B $flags > gpurun_out/aa_${name}_$MR.json 2> gpurun_out/aa_${name}_$MR.err || { echo "$name $MR failed"; continue; }
    python3 - gpurun_out/aa_${name}_$MR.json $name $MR <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "min rows", sys.argv[3], "it/s %.1f" % j["value"], "ssim us", j["roofline_by_kernel"]["so_ssim_l1_fused"]["us"])
PY
  done
done
