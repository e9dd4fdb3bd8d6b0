cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan -o gan -- python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline > gpurun_out/prof_gan.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan/gan_results.db gpurun_out/gan_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline  (4 train_g_d steps: 1 warmup incl. EMA fwd, 2 timed, 1 instrumented)"
tail -1 gpurun_out/prof_gan.log | cut -c1-300
