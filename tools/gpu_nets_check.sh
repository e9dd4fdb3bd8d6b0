cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_nets_gpu.py -m gpu -q 2>&1 | tail -60 > gpurun_out/pytest_nets.log
cat gpurun_out/pytest_nets.log
