cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0
python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "elapsed $SECONDS s"
tail -1 gpurun_out/bench_default.log | cut -c1-3000
tail -3 gpurun_out/bench_default.err
