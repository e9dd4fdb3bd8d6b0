cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in SE3DS_NOP=1 SE3DS_NORM_CG=0; do
  echo "== $v"
  env $v timeout 900 python bench.py --batch 24 --steps 3 --warmup 1 --no-cpu-baseline --no-batch-max --no-warp --no-shipped --no-fp32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.1f value %.2f peak GiB %.1f' % (d['ms_per_step'], d['value'], d['hbm_gib_peak']))"
done
done
