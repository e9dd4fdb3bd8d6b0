# stall-reason counters of the sorted splat kernels (two PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tmp_*
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc_tmp_a -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_query.py 'gpurun_out/pmc_tmp_a/*.db' 'splat_sort'
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d gpurun_out/pmc_tmp_b -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_query.py 'gpurun_out/pmc_tmp_b/*.db' 'splat_sort'
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS -d gpurun_out/pmc_tmp_c -o pmc -- python bench.py --workload warp --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_query.py 'gpurun_out/pmc_tmp_c/*.db' 'splat_sort'
rm -rf gpurun_out/pmc_tmp_*
