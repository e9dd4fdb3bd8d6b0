# gpurun -- 'bash tools/probes/grid_barrier.sh'   (cost of one grid-wide barrier, cross-XCD visibility, launch overheads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/probes/build/grid_barrier > gpurun_out/grid_barrier.txt 2>&1
echo "rc=$?"; cat gpurun_out/grid_barrier.txt
