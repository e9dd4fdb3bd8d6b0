cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ARGS="${SHAPE:-1024 1024 3 1 32 64 1 8}"
rm -rf gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d gpurun_out/pmcA -o pmc -- python tools/one_conv.py $ARGS > gpurun_out/pmcA.log 2>&1
python tools/pmc_query.py 'gpurun_out/pmcA/*.db'
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE -d gpurun_out/pmcB -o pmc -- python tools/one_conv.py $ARGS > gpurun_out/pmcB.log 2>&1
python tools/pmc_query.py 'gpurun_out/pmcB/*.db'
tail -3 gpurun_out/pmcB.log
