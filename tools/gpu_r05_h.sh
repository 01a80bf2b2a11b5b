set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h
mkdir -p $OUT
for i in 1 2 3; do timeout 600 python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-operator-path --step-trace 2>> $OUT/driver_stderr.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-like', round(d['value'],1), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['traffic'], (d['roofline']['valu'] or {}).get('frac'), d['roofline']['profile_notes'])"; done
grep step-trace $OUT/driver_stderr.txt | cut -c1-330
timeout 600 python3 bench.py --regime ref --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --no-operator-path | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref', round(d['value'],1), d['roofline']['kernel'], d['roofline']['traffic'], (d['roofline']['valu'] or {}), d['roofline']['profile_notes'])"
timeout 1200 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_all.txt
tail -5 $OUT/pytest_all.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 300 python3 tools/cpu_c1.py > $OUT/cpu_c1.json 2> $OUT/cpu_c1_stderr.txt; cut -c1-600 $OUT/cpu_c1.json
