cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/prof_warp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp -o warp -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_warp.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_warp/warp_results.db gpurun_out/warp_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline"
