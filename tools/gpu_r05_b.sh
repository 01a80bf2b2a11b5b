set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $OUT
timeout 900 python3 tools/train_demo.py --sweep2 > $OUT/train_sweep2.jsonl 2> $OUT/train_sweep2_stderr.txt; echo "sweep rc $?"
cut -c1-420 $OUT/train_sweep2.jsonl
tail -3 $OUT/train_sweep2_stderr.txt
