# round 5, call be: the LLVM AMDGPU scheduler's options, whole library (never tried before): -amdgpu-sched-strategy=max-ilp /
# max-memory-clause / iterative-minreg, -amdgpu-use-amdgpu-trackers, -amdgpu-schedule-metric-bias=0,
# -amdgpu-disable-unclustered-high-rp-reschedule, -amdgpu-schedule-relaxed-occupancy (iterative-ilp crashes the compiler).
# xnackoff: --offload-arch=gfx950:xnack- (the generic target must also run with XNACK on).
# Scheduling does not change a floating-point result; per-kernel stage times tell which file would take which option.
cd $GRAFT_REPO_ROOT
B="bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for wl in "c2:" "ref:--regime ref --steps 60" "c2:"; do
  name=${wl%%:*}; flags=${wl#*:}
  for LIB in product maxilp maxclause itminreg trackers bias0 norp relaxocc xnackoff; do
    if [ $LIB = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$LIB.so; fi
    timeout -k 10 300 python3 $B $flags > gpurun_out/be_${name}_$LIB.json 2> gpurun_out/be_${name}_$LIB.err || { echo "$name $LIB failed"; continue; }
    python3 - gpurun_out/be_${name}_$LIB.json $name $LIB <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = j["roofline_by_kernel"]
print(sys.argv[2], "%-9s" % sys.argv[3], "it/s %.1f" % j["value"], "ms %.4f" % j["ms_per_step"], " ".join("%s %.1f" % (n.replace("so_", ""), v["us"]) for n, v in k.items()))
PY
  done
done
unset SPLAT_ONE_AMD_LIB
