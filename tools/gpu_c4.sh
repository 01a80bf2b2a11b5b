# BASELINE.json configs[3]: 1M Gaussians, SH degree 3, 2560x1440, densify/prune on -- bench line + rocprofv3 kernel trace
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4
mkdir -p $OUT
timeout 600 python3 bench.py --n 1000000 --width 2560 --height 1440 --densify 100 --steps 200 --warmup 20 --no-cpu-baseline --kernel-table > $OUT/bench.json 2> $OUT/bench_stderr.txt
tail -12 $OUT/bench_stderr.txt; cut -c1-900 $OUT/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --n 1000000 --width 2560 --height 1440 --densify 100 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
python3 - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/c4"
for f in glob.glob(out+"/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows))
    for r in rows[:30]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:12.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
