# Step time against the capacity of the device-resident model (every per-Gaussian launch takes its grid from the capacity
# and clamps to the live count on the device): default max(2N, 2^20) against tighter capacities at N = 100k.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/capacity.jsonl
for CAP in 0 131072 262144 524288 2097152; do
  A=""; [ $CAP -gt 0 ] && A="--max-gaussians $CAP"
  echo "# bench.py $A" >> gpurun_out/capacity.jsonl
  timeout -k 10 300 python bench.py --no-cpu-baseline --kernel-table $A >> gpurun_out/capacity.jsonl 2> gpurun_out/capacity_$CAP.err || exit 1
done
python3 - <<'PY'
import json
for l in open("gpurun_out/capacity.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print("   ", round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms")
PY
grep -h "so_preprocess" gpurun_out/capacity_0.err gpurun_out/capacity_131072.err gpurun_out/capacity_2097152.err
