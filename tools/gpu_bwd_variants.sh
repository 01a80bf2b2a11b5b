# fixed-state stage times of variant libraries (tile-wave backward on), c2 and 500k only
cd $GRAFT_REPO_ROOT
export SPLAT_ONE_AMD_BWD_TILE=1
for V in "" "$@"; do
  if [ -n "$V" ]; then export SPLAT_ONE_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libsplat_one_amd_$V.so; else unset SPLAT_ONE_AMD_LIB; fi
  for CFG in "100000 1920 1080 mcmc" "500000 1920 1080 mcmc" "100000 1920 1080 ref"; do
    timeout -k 10 200 python tools/dbg_bwd_fixed.py $CFG 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${V:-product}', d['N'], d['regime'], 'bwd', d['us']['so_rasterize_bwd'])" || exit 1
  done
done
