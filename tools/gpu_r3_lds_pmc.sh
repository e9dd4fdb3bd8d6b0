# round 3: LDS / VALU / MFMA counters of the 128-channel full-resolution layer (the kernel the
# end-of-round experiments found bound by K-loop data movement), for the next round's redesign
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_l128*
SHAPE="128 128 3 1 512 1024 1 8"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_l128a -o pmc -- python tools/one_conv.py $SHAPE > gpurun_out/pmc_l128a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/pmc_l128b -o pmc -- python tools/one_conv.py $SHAPE > gpurun_out/pmc_l128b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES -d gpurun_out/pmc_l128c -o pmc -- python tools/one_conv.py $SHAPE > gpurun_out/pmc_l128c.log 2>&1
python tools/pmc_summary.py gpurun_out/r03_conv_pmc_128_128_3_1_512_1024_1_8.json --source=se3ds_amd/csrc/conv.hip "gpurun_out/pmc_l128a/*.db" "gpurun_out/pmc_l128b/*.db" "gpurun_out/pmc_l128c/*.db" 'igemm|wgrad'
python - <<'P'
import json
d=json.load(open('gpurun_out/r03_conv_pmc_128_128_3_1_512_1024_1_8.json'))
for k,v in d.items():
    if k.startswith('_'): continue
    print(k)
    for c,x in v.items():
        if isinstance(x,dict): print('   ',c, x['avg'])
        else: print('   ',c, x)
P
rm -rf gpurun_out/pmc_l128a gpurun_out/pmc_l128b gpurun_out/pmc_l128c
