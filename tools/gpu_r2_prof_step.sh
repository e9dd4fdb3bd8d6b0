# kernel-trace stats of the default bench only (see gpu_r2_prof.sh for the PMC passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gan_r2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan_r2 -o gan -- python bench.py --no-cpu-baseline > gpurun_out/prof_gan_r2.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan_r2/gan_results.db gpurun_out/r02_gan_step_b8_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (3 warm-up + 10 timed + 1 instrumented train_g_d step = 14 steps, then the cfg5 warp block; model build kernels included)"
tail -1 gpurun_out/prof_gan_r2.log | cut -c1-300
rm -rf gpurun_out/prof_gan_r2
