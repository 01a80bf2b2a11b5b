# bench.py --kernel-table on the product library and on variant builds (build/variants/libsplat_one_amd_NAME.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
for V in "" "$@"; do
  if [ -n "$V" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$V.so; else unset SPLAT_ONE_AMD_LIB; fi
  for REP in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-operator-path --kernel-table > gpurun_out/variants/bench_${V:-product}_$REP.json 2> gpurun_out/variants/bench_${V:-product}_$REP.err || exit 1
    python -c "
import json,sys; d=json.load(open('gpurun_out/variants/bench_${V:-product}_$REP.json')); k=d['roofline_by_kernel']
print('${V:-product}', $REP, 'it/s %.0f' % d['value'], {a: k[a]['us'] for a in k})"
  done
done
