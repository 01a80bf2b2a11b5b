# Round-5 measurement set, one WORKLOAD per call (counters beyond c2):
#   WL=c2      the default bench (100k Gaussians, 1080p, mcmc preset)                 BARGS=""
#   WL=c2-ref  the reference's default preset (--regime ref)
#   WL=c4      BASELINE configs[3]: 1M Gaussians, 1440p, DefaultStrategy every 100 iterations
#   WL=2M-f32  2M Gaussians, 1080p
#   WL=c5      2M Gaussians, float16 attribute rows
# per workload: (1) the bench line, (2) rocprofv3 --kernel-trace --stats, (3) --pmc FETCH_SIZE / WRITE_SIZE (separate passes, the
# unit and gfx950 corrections of MI355X_MICROARCH.md applied by the summary below), (4) --pmc SQ_* in two passes.  Counters are
# collected with --kernel-trace only (no hip / hsa / memory-copy traces next to --pmc).
#   -> gpurun_out/r05_$WL/{bench.json, prof/, pmc/{traffic_summary.json, sq_summary.json}}; then python tools/refresh_profiles_r05.py
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WL=${WL:-c2}
case $WL in
  c2) BARGS="" ;;
  c2d) BARGS="--steps 20 --warmup 5" ;;
  c2-ref) BARGS="--regime ref --steps 50" ;;
  c4) BARGS="--gaussians 1000000 --width 2560 --height 1440 --densify 100 --steps 100" ;;
  2M-f32) BARGS="--gaussians 2000000 --steps 50" ;;
  c5) BARGS="--gaussians 2000000 --attr-dtype f16 --steps 50" ;;
  c3) BARGS="--gaussians 500000 --steps 100" ;;
  *) echo "unknown WL $WL"; exit 2 ;;
esac
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_$WL
rm -rf $OUT
mkdir -p $OUT/prof $OUT/pmc
Q="--no-cpu-baseline --no-operator-path --no-other-configs"
timeout 600 python3 bench.py --kernel-table $Q $BARGS > $OUT/bench.json 2> $OUT/bench_stderr.txt || exit 1
cut -c1-300 $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py $Q $BARGS > $OUT/prof/stdout.txt 2> $OUT/prof/stderr.txt || exit 1
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc -o pmc_$CTR -- python3 $GRAFT_REPO_ROOT/bench.py $Q $BARGS > $OUT/pmc/stdout_$CTR.txt 2> $OUT/pmc/stderr_$CTR.txt || exit 1
done
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc -o sq_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $Q $BARGS > $OUT/pmc/stdout_$tag.txt 2> $OUT/pmc/stderr_$tag.txt || exit 1
done
cd $GRAFT_REPO_ROOT
WL=$WL python3 tools/profile_summaries.py
# the trace files themselves are large: keep the summaries
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_counter_collection.csv" -delete
du -sh $OUT
