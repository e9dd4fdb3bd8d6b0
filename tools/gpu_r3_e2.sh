# round 3: host thread count vs the oracle's speed on the GPU box (the GPU suite is bound by it)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
nproc; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket" | head -5
for t in 16 32 64 128; do
  OMP_NUM_THREADS=$t MKL_NUM_THREADS=$t python tools/oracle_f64_time.py 2>&1 | grep "torch.float"
done
