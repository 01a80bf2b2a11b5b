# round 5, call m: how many untimed iterations does the GPU need after the read-backs before a 20-step region runs at the long-run rate?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05m
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 20 --warmup 5"
for r in 16 64 128 256 16 64 128 256; do
  echo "== rewarm $r"
  timeout -k 10 150 python3 $B --rewarm-steps $r --step-trace 2> $OUT/t_$r.stderr | cut -c1-150
  grep "consecutive" $OUT/t_$r.stderr | cut -c1-260
done
