set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $OUT
/usr/bin/time -v timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default_stderr.txt; echo "rc $?"
grep "Elapsed (wall" $OUT/bench_default_stderr.txt
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05f/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["ms_per_step"], d["config"]["tile_intersections"], d["config"]["tile_intersections_timed_region"], d["hbm_iter_fraction"], d.get("operator_path_it_s"))
print({k:v["us"] for k,v in d["roofline_by_kernel"].items()})
print(json.dumps(d.get("other_configs"), indent=1)[:3000])
print(d.get("ref_regime")); print(d.get("cpu_baseline"))
PY
for i in 1 2 3; do timeout 600 python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-operator-path --step-trace 2>> $OUT/driver_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-like', round(d['value'],1), d['ms_per_step'])"; done
grep step-trace $OUT/driver_stderr.txt | cut -c1-400
for S in 0.2; do timeout 600 python3 bench.py --cloud-scale $S --no-other-configs --no-cpu-baseline --no-operator-path --kernel-table 2>> $OUT/cloud_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cloud', round(d['value'],1), d['config']['tile_order'], d['config']['tile_intersections'], {k:v['us'] for k,v in d['roofline_by_kernel'].items()})"; SPLAT_ONE_AMD_TILE_ORDER=0 timeout 600 python3 bench.py --cloud-scale $S --no-other-configs --no-cpu-baseline --no-operator-path --kernel-table 2>> $OUT/cloud_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cloud no-lpt', round(d['value'],1), d['config']['tile_order'], d['config']['tile_intersections'], {k:v['us'] for k,v in d['roofline_by_kernel'].items()})"; done
