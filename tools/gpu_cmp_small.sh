# product vs a variant library on the small configuration (c1: 10k Gaussians, 256x256): latency-bound, few tiles
cd $GRAFT_REPO_ROOT
for V in "" "$@"; do
  if [ -n "$V" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$V.so; else unset SPLAT_ONE_AMD_LIB; fi
  for REP in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-operator-path --kernel-table --gaussians 10000 --width 256 --height 256 --steps 400 > gpurun_out/small.json 2> gpurun_out/small.err || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/small.json')); k=d['roofline_by_kernel']
print('${V:-product}', $REP, 'it/s %.0f' % d['value'], {a: k[a]['us'] for a in k})"
  done
done
