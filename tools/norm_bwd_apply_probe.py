"""Fixed cost of norm_bwd_apply vs norm_apply: GPU time (events) and host issue time per call at
tiny and small row counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd import _lib
from se3ds_amd import hipops  # noqa
L = _lib.lib()
DEV = 'cuda:0'
def timeit(fn, reps=100):
  for _ in range(5): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
  for _ in range(reps): fn()
  e1.record(); th = (time.perf_counter() - t0) / reps * 1e6
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3, th
c = 512
for r in (64, 2048, 16384, 65536):
  t = lambda: torch.randn((r, c), device=DEV).bfloat16()
  x, dy, y, dx, dres = t(), t(), t(), t(), t()
  f = lambda: torch.rand(c, device=DEV) + 0.5
  mean, rstd, gamma, scale, shift = f(), f(), f(), f(), f()
  amask = torch.zeros(max(r * c // 8, 1), dtype=torch.uint8, device=DEV)
  sums = torch.zeros((1, 2, c), device=DEV)
  s = _lib.stream()
  a = timeit(lambda: L.se3ds_norm_apply(x.data_ptr(), 3, 1, r, c, scale.data_ptr(), shift.data_ptr(), None, None, 1, 0.0, y.data_ptr(), amask.data_ptr(), s))
  b = timeit(lambda: L.se3ds_norm_bwd_apply(dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), sums.data_ptr(), float(r), 1, 0.0, dx.data_ptr(), None, amask.data_ptr(), 0, 0.3, s))
  d = timeit(lambda: L.se3ds_affine_bwd(dy.data_ptr(), y.data_ptr(), 3, 1, r, c, scale.data_ptr(), 1, 0.0, dx.data_ptr(), None, amask.data_ptr(), s))
  print('rows %6d: apply gpu %.1f us host %.1f us | bwd_apply gpu %.1f host %.1f | affine_bwd gpu %.1f host %.1f' % (r, a[0], a[1], b[0], b[1], d[0], d[1]))
