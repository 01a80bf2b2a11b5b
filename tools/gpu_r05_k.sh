cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05k
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o op -- python3 $GRAFT_REPO_ROOT/bench.py --operator-path --no-cpu-baseline --no-other-configs --steps 200 > $OUT/stdout.txt 2> $OUT/stderr.txt
python3 - <<'PY'
import csv, glob, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05k"
for f in glob.glob(out+"/prof/**/op_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:30]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.1f} pct {r['Percentage']}")
PY
find $OUT -name "*_kernel_trace.csv" -delete
cut -c1-200 $OUT/stdout.txt | tail -2
