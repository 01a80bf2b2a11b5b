set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05i
mkdir -p $OUT
timeout 600 python3 tools/dbg_oppath_host.py 2>&1 | grep -v amdgpu.ids | tail -16
timeout 900 python3 tools/train_demo.py --steps 800 --train-views 32 > $OUT/train_demo.jsonl 2> $OUT/train_demo_stderr.txt; echo "demo rc $?"
python3 - <<'PY'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05i/train_demo.jsonl"):
    d=json.loads(l); o=d.pop("oracle",None); bm=d.pop("loss_block_means")
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items()}); print([round(x,4) for x in bm]); print(o)
PY
