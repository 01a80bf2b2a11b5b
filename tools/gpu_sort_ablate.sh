cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for V in ""; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/kst_$V; rm -rf $OUT; mkdir -p $OUT
  if [ -n "$V" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$V.so; fi
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --cloud-scale 0.2 --n 2000000 --steps 20 > $OUT/out.txt 2> $OUT/err.txt ) || { tail -3 $OUT/err.txt; exit 1; }
  echo "== ${V:-product}"; python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/ks_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tile_sort" in r["Name"]: print(f"{r['Name'][:56]:56s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f}")
PY
done
