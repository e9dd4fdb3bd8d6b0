# round 3: host thread count vs the oracle's speed on the GPU box (the GPU suite is bound by it)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for t in 4 8 12 16 24; do
  OMP_NUM_THREADS=$t MKL_NUM_THREADS=$t python tools/oracle_f64_time.py 2>&1 | grep "torch.float"
done
python - <<'PY'
import torch, time
for t in (8, 16, 32):
  torch.set_num_threads(t)
  x = torch.randn(2, 1024, 8, 16); w = torch.randn(1024, 1024, 3, 3)
  t0 = time.time()
  for _ in range(5): torch.nn.functional.conv2d(x, w, padding=1)
  print('set_num_threads', t, 'conv x5', round(time.time() - t0, 3), flush=True)
PY
