# HBM traffic of the conv kernels of one layer shape (two PMC passes): gpu_pmc_conv.sh [cin cout k s h w pad n]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
ARGS="${*:-1024 1024 3 1 32 64 1 8}"
rm -rf gpurun_out/pmcF gpurun_out/pmcW
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcF -o pmc -- python tools/one_conv.py $ARGS > gpurun_out/pmcF.log 2>&1
python tools/pmc_kernels.py 'gpurun_out/pmcF/*.db' 'igemm|wgrad|thin'
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcW -o pmc -- python tools/one_conv.py $ARGS > gpurun_out/pmcW.log 2>&1
python tools/pmc_kernels.py 'gpurun_out/pmcW/*.db' 'igemm|wgrad|thin'
