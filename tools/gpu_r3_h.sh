cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 2400 python tools/step_compare.py 512 8 14 > gpurun_out/r3_h_cmp512.log 2>&1
grep -v "^STEP" gpurun_out/r3_h_cmp512.log | cut -c1-200
