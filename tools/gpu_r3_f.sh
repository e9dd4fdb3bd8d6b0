cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python tools/step_compare.py 128 2 4 > gpurun_out/r3_f_cmp.log 2>&1
cat gpurun_out/r3_f_cmp.log | cut -c1-260
timeout 900 python tools/step_compare.py 512 8 4 > gpurun_out/r3_f_cmp512.log 2>&1
cat gpurun_out/r3_f_cmp512.log | cut -c1-260
