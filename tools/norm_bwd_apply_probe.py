"""Which operand makes norm_bwd_apply slow on small tensors?  Variants on one shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd import _lib
from se3ds_amd import hipops  # noqa
L = _lib.lib()
DEV = 'cuda:0'
def timeit(fn, reps=50):
  for _ in range(5): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w, c) in [(8, 32, 64, 512), (8, 32, 64, 1024), (8, 64, 128, 256)]:
  r = n * h * w
  t = lambda: torch.randn((r, c), device=DEV).bfloat16()
  x, dy, y, dx, dres = t(), t(), t(), t(), t()
  f = lambda: torch.rand(c, device=DEV) + 0.5
  mean, rstd, gamma = f(), f(), f()
  amask = torch.zeros(r * c // 8, dtype=torch.uint8, device=DEV)
  sums = torch.zeros((1, 2, c), device=DEV)
  s = _lib.stream()
  def run(dres_p, amask_p, act, in_act):
    return timeit(lambda: L.se3ds_norm_bwd_apply(dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c,
                                                 mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                 sums.data_ptr(), float(r), act, 0.0, dx.data_ptr(),
                                                 dres_p, amask_p, in_act, 0.3, s))
  print((n, h, w, c), 'full %.1f | no dres %.1f | no amask (reads y) %.1f | act none %.1f | act none no dres %.1f us' % (
      run(dres.data_ptr(), amask.data_ptr(), 1, 0), run(None, amask.data_ptr(), 1, 0),
      run(dres.data_ptr(), None, 1, 0), run(dres.data_ptr(), None, 0, 0), run(None, None, 0, 0)))
  cp = timeit(lambda: dx.copy_(x))
  print('   torch copy (read + write one tensor): %.1f us' % cp)
