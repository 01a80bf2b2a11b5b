# round 5, call l: (1) what inside k_step_inputs costs its 4.7 us (ablation variants of step.hip: SO_SI_NO_STATUS / _CAM / _SCHED),
# (2) the moment-row touches of k_preprocess_bwd (SO_PP_PREFETCH=1/2), (3) the driver's 20-step region step by step, (4) phase stamps
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05l
mkdir -p $OUT
V=$GRAFT_REPO_ROOT/build/variants
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --no-other-configs"
for v in product si_nostatus si_nocam si_nosched si_none pf1 pf2; do
  if [ $v = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$V/libsplat_one_amd_$v.so; [ -f $SPLAT_ONE_AMD_LIB ] || continue; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o b -- python3 $B --steps 200 > $OUT/$v.stdout 2> $OUT/$v.stderr
  echo "== $v exit $?"
  find $OUT/prof_$v -name "*_kernel_trace.csv" -delete
done
unset SPLAT_ONE_AMD_LIB
python3 - <<'PY'
import csv, glob, json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05l"
res = {}
for v in ("product", "si_nostatus", "si_nocam", "si_nosched", "si_none", "pf1", "pf2"):
    fs = glob.glob(out + f"/prof_{v}/**/b_kernel_stats.csv", recursive=True)
    if not fs:
        continue
    r = {}
    for row in csv.DictReader(open(fs[0])):
        n = row["Name"].split("(")[0]
        for key in ("k_step_inputs", "k_preprocess_bwd", "k_preprocess_fwd"):
            if key in n and int(row["Calls"]) > 100:
                r[key] = round(float(row["AverageNs"]) / 1e3, 2)
    try:
        line = [l for l in open(out + f"/{v}.stdout") if l.startswith("{")][-1]
        r["it_s"] = json.loads(line)["value"]
    except Exception as e:      # noqa: BLE001
        r["it_s"] = repr(e)
    res[v] = r
    print(v, r)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
echo "== plain (no profiler) product / pf1 / pf2, default flags"
for v in product pf1 pf2 product; do
  if [ $v = product ]; then unset SPLAT_ONE_AMD_LIB; else export SPLAT_ONE_AMD_LIB=$V/libsplat_one_amd_$v.so; [ -f $SPLAT_ONE_AMD_LIB ] || continue; fi
  timeout -k 10 150 python3 $B 2> /dev/null | cut -c1-120
done
unset SPLAT_ONE_AMD_LIB
echo "== step trace, driver flags"
timeout -k 10 150 python3 $B --steps 20 --warmup 5 --step-trace 2> $OUT/trace20.stderr | cut -c1-160
grep "step-trace" $OUT/trace20.stderr | cut -c1-700
timeout -k 10 150 python3 $B --steps 80 --warmup 5 --step-trace 2> $OUT/trace80.stderr | cut -c1-160
grep "step-trace" $OUT/trace80.stderr | cut -c1-700
if [ -f $V/libsplat_one_amd_ppstamps.so ]; then
  echo "== stamps"
  SPLAT_ONE_AMD_LIB=$V/libsplat_one_amd_ppstamps.so timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/dbg_ppstamps.py > $OUT/stamps.txt 2>&1
  tail -22 $OUT/stamps.txt
fi
