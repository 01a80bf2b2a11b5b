# rocprofv3 --kernel-trace --stats of the default bench: the ten longest kernels (name, calls, average us)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path "$@" > $OUT/out.txt 2> $OUT/err.txt || exit 1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kstats/**/ks_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f}")
PY
cut -c1-160 $OUT/out.txt
