# warp parity tests + A/B of the warp bench under env switches (one GPU call):
#   gpu_warp_ab.sh "VAR=1" "VAR2=1" ...   (each variant runs random and room depth)
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_warp_gpu.py -x -q -m gpu 2>&1 | tail -2
run() {
  echo "== $1"
  for d in random random room room; do
    env $1 timeout 300 python bench.py --workload warp --warp-depth $d --steps 20 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$d ms/step %.4f  splat ms %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac']))"
  done
}
run "SE3DS_NOP=1"
for v in "$@"; do run "$v"; done
