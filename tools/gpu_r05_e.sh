set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $OUT
rm -f $OUT/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_trains.py tests/test_gpu_trainer.py tests/test_gpu_attr_f16.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-operator-path --no-other-configs --kernel-table"
for L in "" sw1 sw3 sw12 "" sw1; do
  if [ -n "$L" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$L.so; else unset SPLAT_ONE_AMD_LIB; fi
  for ARGS in "" "--n 2000000 --steps 50"; do
    echo "# lib=${L:-product(sw6)} $ARGS" >> $OUT/ab.jsonl
    timeout 600 python3 bench.py $B $ARGS >> $OUT/ab.jsonl 2>> $OUT/ab_stderr.txt || exit 1
  done
done
unset SPLAT_ONE_AMD_LIB
timeout 600 python3 bench.py $B --steps 20 --warmup 5 --step-trace > $OUT/trace20.json 2> $OUT/trace20_stderr.txt
grep step-trace $OUT/trace20_stderr.txt
timeout 600 python3 bench.py $B --step-trace > $OUT/trace200.json 2> $OUT/trace200_stderr.txt
grep step-trace $OUT/trace200_stderr.txt
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05e/ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print(round(d["value"],1), d["config"]["tile_intersections"], {k:v["us"] for k,v in d["roofline_by_kernel"].items()})
PY
