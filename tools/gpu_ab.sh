# A/B of the default bench under env switches (one GPU call): gpu_ab.sh "VAR=1" ...
cd $GRAFT_REPO_ROOT
ulimit -c 0
run() {
  echo "== $1"
  env $1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.2f value %.3f conv frac %.4f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))"
}
for rep in 1 2; do
  run "SE3DS_NOP=1"
  for v in "$@"; do run "$v"; done
done
