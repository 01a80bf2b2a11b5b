# round 5, call r: where does the HOST time of the operator path go (cProfile of bench.py --operator-path)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout -k 10 300 python3 -m cProfile -o $OUT/op.prof bench.py --operator-path --no-cpu-baseline --no-other-configs --steps 400 > $OUT/stdout.txt 2> $OUT/stderr.txt
cut -c1-160 $OUT/stdout.txt | tail -1
python3 - <<'PY'
import pstats
p = pstats.Stats("gpurun_out/r05r/op.prof")
p.sort_stats("cumulative").print_stats(45)
p.sort_stats("tottime").print_stats(30)
PY
