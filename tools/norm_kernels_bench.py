"""Per-kernel bandwidth of the norm family through the C ABI (bf16): apply, bwd_stats, bwd_apply.
Effective bytes = tensors read + written by the call (amask counts 1/8).
COLD=1 (round 6): every launch works on ANOTHER set of buffers, enough
sets that one pass over them exceeds 600 MB (the 256 MB MALL holds none of a set when its turn comes
again) -- the condition these kernels meet inside the step, where 20 back-to-back launches on one
34 MB buffer measured 5-6 TB/s and the step's profile says 2.5-3."""
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
COLD = int(os.environ.get('COLD', '0'))
def timeit(fns, reps=20):
  """fns: one closure per buffer set, called round robin."""
  k = len(fns)
  for i in range(max(3, k)): fns[i % k]()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = max(reps, 2 * k)
  torch.cuda.synchronize(); e0.record()
  for i in range(reps): fns[i % k]()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e-3
only = os.environ.get('SHAPES')   # e.g. SHAPES=4,5: rows of the table below
for si, (n, h, w, c, with_res) in enumerate(shapes):
  if only and str(si) not in only.split(','):
    continue
  r = n * h * w
  nb = r * c * 2
  sets = max(1, -(-600_000_000 // (4 * nb))) if COLD else 1
  t = lambda: torch.randn((r, c), device=DEV).bfloat16()
  f = lambda: torch.rand(c, device=DEV) + 0.5
  scale, shift, mean, rstd, gamma = f(), f(), f(), f(), f()
  sums = torch.zeros((1, 2, c), device=DEV)
  ws = torch.empty(int(L.se3ds_norm_workspace_bytes(1, c)), dtype=torch.uint8, device=DEV)
  s = _lib.stream()
  keep, fa, fs, fb, fc = [], [], [], [], []
  cg_ok = bool(L.se3ds_norm_bwd_cg_supported(3, r, c, ACT, 1 if ACT else 0, 0))
  cws = torch.empty(int(L.se3ds_norm_bwd_cg_workspace_bytes(c)), dtype=torch.uint8, device=DEV) if cg_ok else None
  dbeta, dgamma = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
  for _ in range(sets):
    x, dy, y, res, dx, dres = t(), t(), t(), t(), t(), t()
    amask = torch.empty(r * c // 8, dtype=torch.uint8, device=DEV)
    keep.append((x, dy, y, res, dx, dres, amask))
    fa.append(lambda x=x, y=y, res=res, amask=amask: L.se3ds_norm_apply(
        x.data_ptr(), 3, 1, r, c, scale.data_ptr(), shift.data_ptr(), res.data_ptr() if with_res else None,
        None, ACT, 0.0, y.data_ptr(), amask.data_ptr() if ACT else None, s))
    fs.append(lambda x=x, dy=dy, y=y, amask=amask: L.se3ds_norm_bwd_stats(
        dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c, mean.data_ptr(), rstd.data_ptr(), ACT, 0.0,
        sums.data_ptr(), None, None, amask.data_ptr() if ACT else None, ws.data_ptr(), ws.numel(), s))
    fb.append(lambda x=x, dy=dy, y=y, dx=dx, dres=dres, amask=amask: L.se3ds_norm_bwd_apply(
        dy.data_ptr(), y.data_ptr(), x.data_ptr(), 3, 1, r, c, mean.data_ptr(), rstd.data_ptr(),
        gamma.data_ptr(), sums.data_ptr(), float(r), ACT, 0.0, dx.data_ptr(),
        dres.data_ptr() if with_res else None, amask.data_ptr() if ACT else None, 0, 0.0, s))
    if cg_ok:   # what the step's backward takes for C >= 512: statistics partials + apply with prologue
      fc.append(lambda x=x, dy=dy, dx=dx, dres=dres, amask=amask: L.se3ds_norm_bwd_cg(
          dy.data_ptr(), x.data_ptr(), 3, r, c, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), float(r),
          ACT, 0.0, dx.data_ptr(), dres.data_ptr() if with_res else None, amask.data_ptr() if ACT else None,
          0, 0.0, dbeta.data_ptr(), dgamma.data_ptr(), None, None, None, None, cws.data_ptr(), cws.numel(), s))
  ta, ts, tb = timeit(fa), timeit(fs), timeit(fb)
  tc = timeit(fc) if cg_ok else 0.0
  ba = nb * (2 + (1 if with_res else 0)) + nb / 16
  bs = nb * 2 + nb / 16
  bb = nb * (3 + (1 if with_res else 0)) + nb / 16
  print('%-22s %6.1f MB x %2d sets | apply %7.1f us %5.2f TB/s | bwd_stats %7.1f us %5.2f TB/s | bwd_apply %7.1f us %5.2f TB/s'
        % ((n, h, w, c), nb / 1e6, sets, ta * 1e6, ba / ta / 1e12, ts * 1e6, bs / ts / 1e12, tb * 1e6, bb / tb / 1e12)
        + (' | bwd_cg (2 launches) %7.1f us %5.2f TB/s' % (tc * 1e6, (bs + bb) / tc / 1e12) if cg_ok else ''))
  del keep, fa, fs, fb
  torch.cuda.empty_cache()
