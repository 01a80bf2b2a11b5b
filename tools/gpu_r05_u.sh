# round 5, call u: kernel statistics of the reference's 7000-iteration schedule (tools/train_demo.py --long, engine, both strategies)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05u
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o long -- python3 $GRAFT_REPO_ROOT/tools/train_demo.py --long --res 512 --teacher-n 50000 --student-n 50000 --train-views 32 > $OUT/stdout.txt 2> $OUT/stderr.txt
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05u"
f = glob.glob(out + "/prof/**/long_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f s over %d kernels" % (tot / 1e9, len(rows)))
for r in rows[:28]:
    print(f"{r['Name'].split('(')[0][-64:]:64s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.1f} {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
find $OUT -name "*_kernel_trace.csv" -delete
python3 - <<'PY'
import json, os
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05u/stdout.txt"):
    if l.startswith("{"):
        j = json.loads(l); print(j["strategy"], j["wall_seconds"], [(b["steps"], b["gaussians"], b["it_s"]) for b in j["blocks"]])
PY
