set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $OUT
rm -f $OUT/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_trains.py tests/test_gpu_trainer.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-operator-path --no-other-configs --kernel-table"
for DS in 1 0 1 0; do
  echo "# SSIM_DSTEP=$DS" >> $OUT/ab.jsonl
  SPLAT_ONE_AMD_SSIM_DSTEP=$DS timeout 600 python3 bench.py $B >> $OUT/ab.jsonl 2>> $OUT/ab_stderr.txt || exit 1
done
for LPT in 1 0; do
  for ARGS in "--regime ref --steps 50" "--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100"; do
    echo "# LPT=$LPT $ARGS" >> $OUT/ab.jsonl
    SPLAT_ONE_AMD_TILE_ORDER=$LPT timeout 600 python3 bench.py $B $ARGS >> $OUT/ab.jsonl 2>> $OUT/ab_stderr.txt || exit 1
  done
done
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05d/ab.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print(round(d["value"],1), d["config"]["tile_intersections"], {k:v["us"] for k,v in d["roofline_by_kernel"].items()})
PY
