cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05av; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100 --width 512 --height 512 --gaussians 60000 --regime ref > $OUT/stdout.txt 2> $OUT/stderr.txt
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05av"
f = glob.glob(out + "/prof/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
find $OUT -name "*_kernel_trace.csv" -delete
