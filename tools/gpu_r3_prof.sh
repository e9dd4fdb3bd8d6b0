# round-3 profiles: kernel-trace stats of the default bench and of the warp, PMC passes (HBM traffic)
# of the warp and of the dominant conv shape.  Counters in their own runs (--kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/prof_gan_r3 gpurun_out/pmc_r3_*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gan_r3 -o gan -- python bench.py --no-cpu-baseline --no-batch-max > gpurun_out/prof_gan_r3.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gan_r3/gan_results.db gpurun_out/r03_gan_step_b8_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-batch-max   (3 warm-up + 10 timed + 1 instrumented train_g_d step = 14 steps, then the cfg5 warp block; model build kernels included; decoders on two streams, optimiser on a side stream)"
tail -1 gpurun_out/prof_gan_r3.log | cut -c1-300
head -30 gpurun_out/r03_gan_step_b8_kernel_stats.csv | cut -c1-150
SHAPE="1024 1024 3 1 32 64 1 8"
TAG=$(echo $SHAPE | tr ' ' '_')
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_r3_sq_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_r3_f_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_r3_w_$TAG -o pmc -- python tools/one_conv.py $SHAPE > /dev/null 2>&1
echo "== conv $SHAPE"
python tools/pmc_summary.py gpurun_out/r03_conv_pmc_$TAG.json --source=se3ds_amd/csrc/conv.hip "gpurun_out/pmc_r3_sq_$TAG/*.db" "gpurun_out/pmc_r3_f_$TAG/*.db" "gpurun_out/pmc_r3_w_$TAG/*.db" 'igemm|wgrad'
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_r3_f_warp -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_r3_w_warp -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
echo "== warp"
python tools/pmc_summary.py gpurun_out/r03_warp_pmc.json --source=se3ds_amd/csrc/geom.hip "gpurun_out/pmc_r3_f_warp/*.db" "gpurun_out/pmc_r3_w_warp/*.db" 'splat|unproject'
for d in random room; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_warp_r3 -o warp -- python bench.py --workload warp --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
  python tools/rocpd_summary.py gpurun_out/prof_warp_r3/warp_results.db gpurun_out/r03_warp_${d}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload warp --warp-depth $d --steps 200 --warmup 20 --no-cpu-baseline"
  head -9 gpurun_out/r03_warp_${d}_kernel_stats.csv | cut -c1-150
  rm -rf gpurun_out/prof_warp_r3
done
rm -rf gpurun_out/prof_gan_r3 gpurun_out/pmc_r3_*
du -sh gpurun_out
