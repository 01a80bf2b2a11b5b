# PMC counters of one probe binary:  bash tools/gpu_pmc_bin.sh tools/probes/ssim_64x32.bin [kernel-substring]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/$1
PAT=${2:-ssim}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bin
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT -o set$i -- $BIN > $OUT/stdout_$i.txt 2> $OUT/stderr_$i.txt
done
python3 - "$PAT" <<'PY'
import csv, glob, os, collections, sys
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_bin"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if sys.argv[1] in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k[:60])
    for c,x in sorted(v.items()):
        print(f"    {c:28s} {sum(x)/len(x):16.0f}")
PY
tail -3 $OUT/stderr_4.txt
