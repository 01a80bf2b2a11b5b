# Round-4 measurement set of the default bench (c2, 8 ring views cycled) with the final library:
#  (1) the bench line incl. operator-path rate and CPU baseline     -> gpurun_out/r04/bench.json
#  (2) rocprofv3 --kernel-trace --stats                             -> gpurun_out/r04/prof/bench_kernel_stats.csv
#  (3) rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)    -> gpurun_out/r04/pmc/traffic_summary.json
#  (4) rocprofv3 --pmc SQ_* (two passes)                            -> gpurun_out/r04/pmc/sq_summary.json
#  (5) HIP API calls of 100 more iterations with one MCMC refinement in them (device-side strategy: +0 syncs / copies)
#  (6) the other regimes / sizes, incl. clouds gathered in the middle of the image (--cloud-scale 0.2: skewed tile load)                                    -> gpurun_out/r04/other_lines.jsonl
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export RTAG
# RTAG=r04d BARGS="--steps 20 --warmup 5": the same counter passes at the state the round-end driver benches (its command
# line: 25 iterations in all, ~308k tile intersections instead of ~372k) -- bench.py picks the collection that matches its run
RTAG=${RTAG:-r04}
BARGS=${BARGS:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$RTAG
mkdir -p $OUT/prof $OUT/pmc $OUT/hip
if [ "${PART:-A}" = "A" ]; then
timeout 600 python3 bench.py --kernel-table $BARGS > $OUT/bench.json 2> $OUT/bench_stderr.txt || exit 1
cut -c1-600 $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path $BARGS > $OUT/prof/stdout.txt 2> $OUT/prof/stderr.txt || exit 1
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -o pmc_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path $BARGS > $OUT/pmc/stdout_$CTR.txt 2> $OUT/pmc/stderr_$CTR.txt || exit 1
done
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc -o sq_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path $BARGS > $OUT/pmc/stdout_$tag.txt 2> $OUT/pmc/stderr_$tag.txt || exit 1
done
for S in $( [ -z "$BARGS" ] && echo 150 250 ); do
  timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hip -o hip$S -- python3 $GRAFT_REPO_ROOT/tools/dbg_refine_sync.py --strategy mcmc --n 100000 --width 1920 --height 1080 --steps $S > $OUT/hip/hip${S}_stdout.txt 2> $OUT/hip/hip${S}_stderr.txt || exit 1
  tail -1 $OUT/hip/hip${S}_stdout.txt
done
python3 - <<'PY'
import csv, glob, os, json, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ.get("RTAG","r04")
for f in glob.glob(out+"/prof/**/bench_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
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
    # gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream -> x2; WRITE_SIZE exact (MI355X_MICROARCH.md)
    summary[k]={"FETCH_SIZE_KB":f,"WRITE_SIZE_KB":w,"hbm_bytes_per_launch_corrected":2*f*1024+w*1024,"launches":v.get("calls",0)}
for k,v in sorted(summary.items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch_corrected"])[:12]:
    print(f"{k[:70]:70s} {v}")
json.dump(summary, open(out+"/pmc/traffic_summary.json","w"), indent=1)
sq=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc/**/sq_*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "so::" in k:
            sq[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
sqs={k:{c: round(sum(x)/len(x)) for c,x in sorted(v.items())} for k,v in sq.items()}
json.dump(sqs, open(out+"/pmc/sq_summary.json","w"), indent=1)
for k,v in sqs.items():
    if "SQ_INSTS_VALU" in v: print(k[:60], v["SQ_INSTS_VALU"], v.get("SQ_THREAD_CYCLES_VALU"), v.get("SQ_ACTIVE_INST_VALU"))
counts={}
for S in (150,250):
    for f in glob.glob(out+f"/hip/**/hip{S}_hip_api_stats.csv", recursive=True):
        counts[S]={r['Name']: int(r['Calls']) for r in csv.DictReader(open(f))}
if len(counts)==2:
    names=sorted(set(counts[150])|set(counts[250]))
    diff={n: counts[250].get(n,0)-counts[150].get(n,0) for n in names}
    keep={n:(counts[150].get(n,0),counts[250].get(n,0),d) for n,d in diff.items() if ("Sync" in n or "Memcpy" in n or "Malloc" in n or "Free" in n or "Graph" in n or d)}
    json.dump({"_what": "rocprofv3 --hip-trace --stats of tools/dbg_refine_sync.py --strategy mcmc (100k Gaussians, 1080p, refinement every 100 iterations, noise every iteration) at 150 and 250 iterations: [calls at 150, calls at 250, difference] -- the 100 extra iterations hold ONE MCMC refinement (step 200)",
               "calls_150_250_diff": keep}, open(out+"/hip/hip_api_diff.json","w"), indent=1)
    for n,v in keep.items(): print(n, v)
PY
fi   # PART A
[ "${PART:-A}" = "B" ] || exit 0
cd $GRAFT_REPO_ROOT
: > $OUT/other_lines.jsonl
for ARGS in "--regime ref" "--n 500000" "--n 2000000 --steps 50" "--n 2000000 --attr-dtype f16 --steps 50" "--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 300" "--cloud-scale 0.2" "--cloud-scale 0.2 --n 2000000 --steps 50"; do
  echo "# bench.py $ARGS" >> $OUT/other_lines.jsonl
  timeout 600 python3 bench.py --no-cpu-baseline --no-operator-path --kernel-table $ARGS >> $OUT/other_lines.jsonl 2> $OUT/other_stderr.txt || exit 1
done
cut -c1-300 $OUT/other_lines.jsonl
