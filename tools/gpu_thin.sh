cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 600 python -m pytest tests/test_nets_gpu.py -x -q -m gpu -k "thin or conv_fwd_bwd or macro" 2>&1 | tail -4
SE3DS_BENCH_SHAPES=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep SHAPE | grep -- "->3 \|->1 \|4->\|5->" 
for v in SE3DS_NO_THIN_FWD=1 SE3DS_NOP=1 SE3DS_NO_THIN_FWD=1 SE3DS_NOP=1; do
  echo "== $v"
  env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.2f value %.3f' % (d['ms_per_step'], d['value']))"
done
