# LDS-side counters of the dominant conv kernels (own pass: --kernel-trace + --pmc only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_lds*
SHAPE="1024 1024 3 1 32 64 1 8"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_lds1 -o pmc -- python tools/one_conv.py $SHAPE > gpurun_out/pmc_lds1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/pmc_lds2 -o pmc -- python tools/one_conv.py $SHAPE > gpurun_out/pmc_lds2.log 2>&1
tail -3 gpurun_out/pmc_lds1.log gpurun_out/pmc_lds2.log
python tools/pmc_summary.py gpurun_out/r02_conv_lds_pmc.json "gpurun_out/pmc_lds1/*.db" "gpurun_out/pmc_lds2/*.db" 'igemm|wgrad'
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_conv_lds_pmc.json'))
for k,v in d.items():
    print(k)
    for c,x in v.items():
        if isinstance(x,dict): print('   ',c, x['avg'])
P
rm -rf gpurun_out/pmc_lds1 gpurun_out/pmc_lds2
