# Round-2 measurement set of the default bench (c2, 8 ring views cycled) with the final library:
#  (1) the bench line incl. CPU baseline            -> gpurun_out/r02/bench.json
#  (2) rocprofv3 --kernel-trace --stats             -> gpurun_out/r02/prof/bench_kernel_stats.csv
#  (3) rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no other trace domain) -> traffic_summary.json
#  (4) the other regimes / sizes                    -> gpurun_out/r02/other_lines.jsonl
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $OUT/prof $OUT/pmc
timeout 600 python3 bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_stderr.txt || exit 1
cut -c1-600 $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/prof/stdout.txt 2> $OUT/prof/stderr.txt || exit 1
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -o pmc_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --no-cpu-baseline > $OUT/pmc/stdout_$CTR.txt 2> $OUT/pmc/stderr_$CTR.txt || exit 1
done
python3 - <<'PY'
import csv, glob, os, json, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r02"
for f in glob.glob(out+"/prof/**/bench_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
res=collections.defaultdict(dict)
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob(out+f"/pmc/**/pmc_{ctr}_counter_collection.csv", recursive=True)
    if not fs: print("no file for",ctr); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]==ctr:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        res[k][ctr]=sum(v)/len(v); res[k]["calls"]=len(v)
summary={}
for k,v in res.items():
    f=v.get("FETCH_SIZE",0.0); w=v.get("WRITE_SIZE",0.0)
    # gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream -> x2; WRITE_SIZE exact
    summary[k]={"FETCH_SIZE_KB":f,"WRITE_SIZE_KB":w,"hbm_bytes_per_launch_corrected":2*f*1024+w*1024,"launches":v.get("calls",0)}
for k,v in sorted(summary.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch_corrected"])[:14]:
    print(f"{k[:70]:70s} {v}")
json.dump(summary, open(out+"/pmc/traffic_summary.json","w"), indent=1)
PY
cd $GRAFT_REPO_ROOT
: > $OUT/other_lines.jsonl
for ARGS in "--regime ref" "--operator-path" "--n 500000" "--n 2000000 --steps 50" "--n 2000000 --attr-dtype f16 --steps 50" "--views 1" "--loss-kernels 2"; do
  echo "# bench.py $ARGS" >> $OUT/other_lines.jsonl
  timeout 600 python3 bench.py --no-cpu-baseline --kernel-table $ARGS >> $OUT/other_lines.jsonl 2> $OUT/other_stderr.txt || exit 1
done
cut -c1-400 $OUT/other_lines.jsonl
