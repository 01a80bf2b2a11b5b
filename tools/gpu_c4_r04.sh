# BASELINE.json configs[3] in round 4: 1M Gaussians, 2560x1440, DefaultStrategy on the DEVICE every 100 iterations.
#  (1) the bench line (N grows inside the timed region, one refinement timed with HIP events)
#  (2) rocprofv3 kernel stats of the same command
#  (3) HIP API call counts of a plain training loop at 400 and 600 iterations: what two more refinements cost in
#      host-side synchronisation
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4r04
mkdir -p $OUT
ARGS="--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 300 --no-cpu-baseline"
timeout 600 python3 bench.py $ARGS --kernel-table > $OUT/bench.json 2> $OUT/bench_stderr.txt
grep -v amdgpu $OUT/bench_stderr.txt | tail -12; cut -c1-1500 $OUT/bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_stdout.txt 2> $OUT/prof_stderr.txt
for S in 400 600; do
  timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d $OUT -o hip$S -- python3 $GRAFT_REPO_ROOT/tools/dbg_refine_sync.py --steps $S > $OUT/hip${S}_stdout.txt 2> $OUT/hip${S}_stderr.txt
  tail -2 $OUT/hip${S}_stdout.txt
done
python3 - <<'PY'
import csv, glob, os, json
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/c4r04"
for f in glob.glob(out+"/**/c4_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:24]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:12.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
counts={}
for S in (400,600):
    for f in glob.glob(out+f"/**/hip{S}_hip_api_stats.csv", recursive=True):
        counts[S]={r['Name']: int(r['Calls']) for r in csv.DictReader(open(f))}
if len(counts)==2:
    names=sorted(set(counts[400])|set(counts[600]))
    diff={n: counts[600].get(n,0)-counts[400].get(n,0) for n in names}
    keep={n:(counts[400].get(n,0),counts[600].get(n,0),d) for n,d in diff.items() if ("Sync" in n or "Memcpy" in n or "Malloc" in n or "Free" in n or "Graph" in n or d)}
    json.dump({"calls_400_600_diff": keep}, open(out+"/hip_api_diff.json","w"), indent=1)
    for n,v in keep.items(): print(n, v)
PY
