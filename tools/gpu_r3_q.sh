# round 3: device idle time between kernels (serial schedule and default schedule)
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
rm -rf gpurun_out/prof_gap
