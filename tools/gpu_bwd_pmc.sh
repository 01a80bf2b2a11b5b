# SQ counters of the backward rasteriser in both mappings (fixed model state): rocprofv3 --pmc on tools/dbg_bwd_fixed.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bwd_pmc
rm -rf $OUT; mkdir -p $OUT
for T in 0 1; do
  export SPLAT_ONE_AMD_BWD_TILE=$T
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"; do
    tag=t${T}_$(echo $SET | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -o $tag -- python3 $GRAFT_REPO_ROOT/tools/dbg_bwd_fixed.py ${CFG:-100000 1920 1080 mcmc} > $OUT/stdout_$tag.txt 2> $OUT/stderr_$tag.txt) || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os, json, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/bwd_pmc"
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*_counter_collection.csv", recursive=True):
    t=os.path.basename(f).split("_")[0]
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "rasterize_bwd" in k:
            res[t+" "+k][r["Counter_Name"]].append(float(r["Counter_Value"]))
s={k:{c: round(sum(x)/len(x)) for c,x in sorted(v.items())} for k,v in res.items()}
json.dump(s, open(out+"/summary.json","w"), indent=1)
for k,v in s.items(): print(k, v)
PY
