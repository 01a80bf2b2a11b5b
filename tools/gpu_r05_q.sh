# round 5, call q: where are the gaps between the kernels of a c2 step (rocprofv3 kernel trace, start / end stamps of consecutive kernels)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05q
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-path --no-other-configs --steps 100 > $OUT/stdout.txt 2> $OUT/stderr.txt
python3 - <<'PY'
import csv, glob, os, collections, json
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05q"
f = glob.glob(out + "/prof/**/b_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("so::", "")
    return n.split("<")[0]
# the timed region: the last 100 k_step_inputs launches
idx = [i for i, r in enumerate(rows) if "k_step_inputs" in r["Kernel_Name"]]
lo = idx[-101] if len(idx) > 101 else idx[0]
hi = idx[-1]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows[lo:hi], rows[lo + 1:hi + 1]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    durs[short(a["Kernel_Name"])].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
tot_gap = 0.0
res = {}
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    v2 = sorted(v)
    print("%-28s -> %-28s n %4d  gap mean %6.2f  p50 %6.2f  p90 %6.2f us" % (k[0], k[1], len(v), sum(v) / len(v), v2[len(v) // 2], v2[int(len(v) * 0.9)]))
    tot_gap += sum(v) / 100.0
    res[k[0] + " -> " + k[1]] = {"n": len(v), "mean_us": round(sum(v) / len(v), 2), "p50_us": round(v2[len(v) // 2], 2)}
print("gaps per step: %.1f us;  kernels per step: %.1f us;  span per step %.1f us" % (tot_gap, sum(sum(v) for v in durs.values()) / 100.0,
      (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e5))
json.dump(res, open(out + "/gaps.json", "w"), indent=1)
PY
find $OUT -name "*_kernel_trace.csv" -delete
cut -c1-200 $OUT/stdout.txt | tail -1
