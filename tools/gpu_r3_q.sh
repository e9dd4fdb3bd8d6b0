# round 3: device idle time between kernels (serial schedule and default schedule), then the
# configs tests with the faster fp64 oracle
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
for mode in serial default; do
  rm -rf gpurun_out/prof_gap
  if [ $mode = serial ]; then export SE3DS_DUAL_STREAM=0 SE3DS_SEGMENT_OPTIMIZER=0; else unset SE3DS_DUAL_STREAM SE3DS_SEGMENT_OPTIMIZER; fi
  rocprofv3 --kernel-trace -d gpurun_out/prof_gap -o gap -- python tools/step_times.py > gpurun_out/r3_q_steps_$mode.log 2>&1
  tail -3 gpurun_out/r3_q_steps_$mode.log
  echo "== $mode"
  python tools/gap_analysis.py gpurun_out/prof_gap/gap_results.db 0.6 | tee gpurun_out/r3_q_gaps_$mode.log
done
unset SE3DS_DUAL_STREAM SE3DS_SEGMENT_OPTIMIZER
rm -rf gpurun_out/prof_gap
SECONDS=0
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_nets_gpu.py -m gpu -x -q -s --durations=8 2>&1 | grep -v "^$" | tail -40 | cut -c1-200
echo "configs+nets elapsed $SECONDS s"
