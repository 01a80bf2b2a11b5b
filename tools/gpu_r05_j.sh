set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05j
mkdir -p $OUT
rm -f $OUT/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_refine.py tests/test_gpu_threads.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-operator-path --no-other-configs --kernel-table"
for F in 1 0 1 0; do
  for ARGS in "" "--gaussians 500000 --steps 100"; do
    echo "# FOLD=$F $ARGS" >> $OUT/ab.jsonl
    SPLAT_ONE_AMD_SORT_FOLD=$F timeout 600 python3 bench.py $B $ARGS >> $OUT/ab.jsonl 2>> $OUT/ab_stderr.txt || exit 1
  done
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05j/ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print(round(d["value"],1), d["ms_per_step"], d["config"]["tile_intersections"], {k:v["us"] for k,v in d["roofline_by_kernel"].items()}, round(d["forward_mpix_per_s"]))
PY
