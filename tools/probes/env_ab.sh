# parity tests by -k expression, then tools/conv_bench.py under env variants (2 reps)
# usage: env_ab.sh '<pytest -k expr>' '<grep pattern of conv_bench lines>' ENV=v [ENV=v ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
k="$1"; pat="$2"; shift; shift
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -x -q -k "$k" > gpurun_out/env_ab_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/env_ab_tests.log | cut -c1-220
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)"
    env $v N=8 timeout 600 python tools/conv_bench.py 2>&1 | grep "$pat" | cut -c1-170
  done
done
