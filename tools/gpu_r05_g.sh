set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $OUT
rm -f $OUT/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_raster_op.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-operator-path --no-other-configs --kernel-table"
for L in "" noearly "" noearly; do
  if [ -n "$L" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$L.so; else unset SPLAT_ONE_AMD_LIB; fi
  for ARGS in "" "--n 2000000 --steps 50" "--cloud-scale 0.2"; do
    echo "# lib=${L:-product} $ARGS" >> $OUT/ab.jsonl
    timeout 600 python3 bench.py $B $ARGS >> $OUT/ab.jsonl 2>> $OUT/ab_stderr.txt || exit 1
  done
done
unset SPLAT_ONE_AMD_LIB
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05g/ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print(round(d["value"],1), d["config"]["tile_intersections"], {k:v["us"] for k,v in d["roofline_by_kernel"].items()})
PY
S=$(date +%s)
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default_stderr.txt; echo "rc $? wall $(( $(date +%s) - S )) s"
python3 - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05g/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["ms_per_step"], d["config"]["tile_intersections"], d["config"]["tile_intersections_timed_region"], d["hbm_iter_fraction"], d.get("operator_path_it_s"))
print({k:v["us"] for k,v in d["roofline_by_kernel"].items()})
print(json.dumps(d.get("other_configs"), indent=1)[:3500])
print(d.get("ref_regime")); print(d.get("cpu_baseline"))
PY
