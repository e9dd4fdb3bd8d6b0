cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 2 --warmup 1 --batch 2 --no-cpu-baseline > gpurun_out/bench_gan_b2.log 2>&1
tail -3 gpurun_out/bench_gan_b2.log | cut -c1-2500
