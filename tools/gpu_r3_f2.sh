# round 3: the production-size bit-identity test + default bench line (traffic key)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
SECONDS=0
timeout 1200 python -m pytest tests/test_nets_gpu.py -m gpu -x -q -k "bit_identical" --durations=3 2>&1 | tail -8 | cut -c1-200
echo "elapsed $SECONDS s"
python bench.py --no-batch-max 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value'],3), round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['traffic'], d['warp']['roofline']['traffic'], d['cpu_baseline']['value'])
"
