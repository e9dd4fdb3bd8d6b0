cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gan
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan -o gan -- python bench.py > gpurun_out/prof_gan.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan/gan_results.db gpurun_out/gan_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py   (default: 3 warm-up + 10 timed + 1 instrumented train_g_d step = 14 steps, the first one with the step-0 EMA forward; model build kernels included)"
tail -1 gpurun_out/prof_gan.log | cut -c1-400
