# Where does an operator-path iteration go?  (1) host issue time vs wall time + cProfile (tools/dbg_host.py --operator),
# (2) rocprofv3 kernel trace of the same loop: GPU-busy time per iteration.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/opprof
timeout -k 10 300 python tools/dbg_host.py --operator > gpurun_out/opprof/host.txt 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/opprof
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o op -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --operator-path --no-operator-path > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/opprof"
for f in glob.glob(out+"/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, len(rows), "total ms", tot/1e6)
    for r in rows[:32]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:12.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
head -70 gpurun_out/opprof/host.txt
