cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/counters.txt | sort -u | tr '\n' ' ' | head -c 6000
echo
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -o pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stdout_$tag.txt 2> $OUT/stderr_$tag.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc2"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "so::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k[:60])
    print("   ", {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
