# round-2 profiles: kernel-trace stats of the default bench, PMC passes (MFMA busy, HBM traffic) of the
# dominant conv shapes and of the warp.  Counters in their own runs (--kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gan_r2 gpurun_out/pmc_r2_*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan_r2 -o gan -- python bench.py --no-cpu-baseline > gpurun_out/prof_gan_r2.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan_r2/gan_results.db gpurun_out/r02_gan_step_b8_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (3 warm-up + 10 timed + 1 instrumented train_g_d step = 14 steps, then the cfg5 warp block; model build kernels included)"
tail -1 gpurun_out/prof_gan_r2.log | cut -c1-300
head -24 gpurun_out/r02_gan_step_b8_kernel_stats.csv | cut -c1-150
for SHAPE in "1024 1024 3 1 32 64 1 8" "128 128 3 1 256 512 1 8" "512 2048 1 1 32 64 0 8"; do
  TAG=$(echo $SHAPE | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_r2_sq_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_r2_f_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_r2_w_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
  echo "== conv $SHAPE"
  python tools/pmc_summary.py gpurun_out/r02_conv_pmc_$TAG.json "gpurun_out/pmc_r2_sq_$TAG/*.db" "gpurun_out/pmc_r2_f_$TAG/*.db" "gpurun_out/pmc_r2_w_$TAG/*.db" 'igemm|wgrad'
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_r2_f_warp -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_r2_w_warp -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
echo "== warp"
python tools/pmc_summary.py gpurun_out/r02_warp_pmc.json "gpurun_out/pmc_r2_f_warp/*.db" "gpurun_out/pmc_r2_w_warp/*.db" 'splat|unproject'
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp_r2b -o warp -- python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/prof_warp_r2b/warp_results.db gpurun_out/r02_warp_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --steps 200 --warmup 20 --no-cpu-baseline"
head -8 gpurun_out/r02_warp_kernel_stats.csv | cut -c1-150
# the raw databases are large (the merge back is limited to 64 MiB): keep the summaries only
rm -rf gpurun_out/prof_gan_r2 gpurun_out/prof_warp_r2b gpurun_out/pmc_r2_*
du -sh gpurun_out
