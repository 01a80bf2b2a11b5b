# rocprofv3 kernel trace of the bench (per-kernel durations); summary copied to gpurun_out/
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
ls -R $OUT | head -30
python3 - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/prof"
for f in glob.glob(out+"/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows))
    for r in rows[:40]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:12.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
