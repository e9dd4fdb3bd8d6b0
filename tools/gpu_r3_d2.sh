# round 3: (1) oracle fp64 formulations on the box's host cores, (2) deferred batched wgrad reduce
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
SE3DS_ORACLE_F64_DIV=kernel python tools/oracle_f64_time.py 2>&1 | grep "torch.float"
SE3DS_ORACLE_F64_DIV=output python tools/oracle_f64_time.py 2>&1 | grep "torch.float"
SE3DS_ORACLE_F64_DIV=kernel python tools/oracle_f64_time.py 2>&1 | grep "float64"
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))
"; }
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_blocks_gpu.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
for rep in 1 2; do
  SE3DS_WGRAD_DEFER=0 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "defer=0"
  SE3DS_WGRAD_DEFER=1 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "defer=1"
done
