# HBM traffic per kernel from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots: FETCH_SIZE needs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o pmc_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/stdout_$CTR.txt 2> $OUT/stderr_$CTR.txt
done
ls $OUT
python3 - <<'PY'
import csv, glob, os, json, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc"
res=collections.defaultdict(dict)
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob(out+f"/**/pmc_{ctr}_counter_collection.csv", recursive=True)
    if not fs: print("no file for",ctr); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]==ctr:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        res[k][ctr]=sum(v)/len(v); res[k]["calls"]=len(v)
rows=[]
for k,v in res.items():
    f=v.get("FETCH_SIZE",0.0); w=v.get("WRITE_SIZE",0.0)
    # gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream -> x2; WRITE_SIZE exact
    rows.append((2*f*1024+w*1024, k, f, w, v.get("calls",0)))
rows.sort(reverse=True)
summary={}
for tot,k,f,w,c in rows[:25]:
    print(f"{k[:70]:70s} calls {c:4d} FETCH_KB {f:12.1f} WRITE_KB {w:12.1f} hbm_bytes_corrected {tot:14.0f}")
    summary[k]={"FETCH_SIZE_KB":f,"WRITE_SIZE_KB":w,"hbm_bytes_per_launch_corrected":tot,"launches":c}
json.dump(summary, open(out+"/traffic_summary.json","w"), indent=1)
PY
