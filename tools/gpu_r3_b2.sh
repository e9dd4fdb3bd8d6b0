# round 3: block counts of the norm statistics / element-wise kernels after the conversion fix
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))
"; }
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "default(2048,8)"
SE3DS_NORM_STAT_BLOCKS=1024 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "stat=1024"
SE3DS_NORM_STAT_BLOCKS=512 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "stat=512"
SE3DS_NORM_STAT_BLOCKS=4096 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "stat=4096"
SE3DS_NORM_EW_BLOCKS=4 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "ew=4"
SE3DS_NORM_EW_BLOCKS=16 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "ew=16"
python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "default(2048,8)"
