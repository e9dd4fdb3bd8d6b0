# HBM traffic of the warp kernels from the L2's fabric-side counters (two PMC passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
rm -rf gpurun_out/pmcF gpurun_out/pmcW
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcF -o pmc -- python bench.py --workload warp --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcF.log 2>&1
python tools/pmc_kernels.py 'gpurun_out/pmcF/*.db' 'splat|unproject'
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcW -o pmc -- python bench.py --workload warp --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcW.log 2>&1
python tools/pmc_kernels.py 'gpurun_out/pmcW/*.db' 'splat|unproject'
