"""Per-kernel bandwidth of the norm family through the C ABI (bf16): apply, bwd_stats, bwd_apply.
Effective bytes = tensors read + written by the call (amask counts 1/8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd import _lib
from se3ds_amd import hipops  # noqa
L = _lib.lib()
DEV = 'cuda:0'
ACT = int(os.environ.get('ACT', '1'))   # activation kind of the norm (0 none, 1 relu): selects the kernel variants
shapes = [(8, 512, 1024, 128, False), (8, 256, 512, 128, True), (8, 128, 256, 128, True),
          (8, 64, 128, 256, True), (8, 32, 64, 512, True), (8, 32, 64, 1024, True), (8, 32, 64, 2048, True)]
def timeit(fn, reps=20):
  for _ in range(3): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e-3
for n, h, w, c, with_res in shapes:
  r = n * h * w
  t = lambda: torch.randn((r, c), device=DEV).bfloat16()
  x, dy, y, res, dx, dres = t(), t(), t(), t(), t(), t()
  f = lambda: torch.rand(c, device=DEV) + 0.5
  scale, shift, mean, rstd, gamma = f(), f(), f(), f(), f()
  amask = torch.empty(r * c // 8, dtype=torch.uint8, device=DEV)
  sums = torch.zeros((1, 2, c), device=DEV)
  ws = torch.empty(int(L.se3ds_norm_workspace_bytes(1, c)), dtype=torch.uint8, device=DEV)
  s = _lib.stream()
  nb = r * c * 2
  ta = timeit(lambda: L.se3ds_norm_apply(x.data_ptr(), 3, 1, r, c, scale.data_ptr(), shift.data_ptr(),
                                         res.data_ptr() if with_res else None, None, ACT, 0.0,
                                         y.data_ptr(), amask.data_ptr() if ACT else None, s))
  ts = timeit(lambda: L.se3ds_norm_bwd_stats(dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c,
                                             mean.data_ptr(), rstd.data_ptr(), ACT, 0.0, sums.data_ptr(),
                                             None, None, amask.data_ptr() if ACT else None, ws.data_ptr(), ws.numel(), s))
  tb = timeit(lambda: L.se3ds_norm_bwd_apply(dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c,
                                             mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                             sums.data_ptr(), float(r), ACT, 0.0, dx.data_ptr(),
                                             dres.data_ptr() if with_res else None, amask.data_ptr() if ACT else None,
                                             0, 0.0, s))
  ba = nb * (2 + (1 if with_res else 0)) + nb / 16
  bs = nb * 2 + nb / 16
  bb = nb * (3 + (1 if with_res else 0)) + nb / 16
  print('%-22s %6.1f MB | apply %7.1f us %5.2f TB/s | bwd_stats %7.1f us %5.2f TB/s | bwd_apply %7.1f us %5.2f TB/s'
        % ((n, h, w, c), nb / 1e6, ta * 1e6, ba / ta / 1e12, ts * 1e6, bs / ts / 1e12, tb * 1e6, bb / tb / 1e12))
