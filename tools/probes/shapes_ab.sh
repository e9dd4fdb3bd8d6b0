# per-shape conv times inside the instrumented step (SE3DS_BENCH_SHAPES=1) under env variants
# usage: shapes_ab.sh TAG ENV=v [ENV=v ...]   -> gpurun_out/shapes_TAG_<i>_<rep>.log
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=$1; shift
for rep in 1 2; do
  i=0
  for v in "$@"; do
    env $v SE3DS_BENCH_SHAPES=all timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 2>/dev/null > gpurun_out/shapes_${tag}_${i}_${rep}.log
    python - gpurun_out/shapes_${tag}_${i}_${rep}.log "$v rep$rep" <<'PY'
import sys, json
lines = open(sys.argv[1]).read().splitlines()
sh = [l.split() for l in lines if l.startswith('SHAPE')]
t128 = sum(float(s[1]) for s in sh if '128->128' in ' '.join(s))
t11 = sum(float(s[1]) for s in sh if '1x1s' in ' '.join(s))
tall = sum(float(s[1]) for s in sh)
d = json.loads([l for l in lines if l.startswith('{')][-1])
r = d['roofline']
print('%-28s ms/step %.2f conv_ms %.2f (listed shapes %.2f, 3x3 128->128 %.2f, 1x1 %.2f) frac %.4f' % (
    sys.argv[2], d['ms_per_step'], r['conv_ms_per_step'], tall, t128, t11, r['frac']))
PY
    i=$((i+1))
  done
done
