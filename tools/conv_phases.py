"""Where a conv kernel's time goes: the same launch with its epilogue or its K loop compiled out
(a -DSE3DS_PROBE build of conv.hip, tools/probes/conv_phases.sh; never the shipped library).
  SE3DS_PROBE_MODE=0 whole kernel, 1 no epilogue (prologue + K loop), 2 no K loop (prologue + epilogue)
Shapes: the 1x1 family of the encoder at batch 8 and, for scale, the dominant 3x3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
DEV = 'cuda:0'
N = int(os.environ.get('N', '8'))
shapes = [  # name, cin, cout, k, h, w, pad, kind
    ('1x1 512->2048 @32x64 partial_spectral', 512, 2048, 1, 32, 64, 0, 'partial_spectral'),
    ('1x1 2048->512 @32x64 partial_spectral', 2048, 512, 1, 32, 64, 0, 'partial_spectral'),
    ('1x1 512->2048 @32x64 plain', 512, 2048, 1, 32, 64, 0, 'plain'),
    ('1x1 256->1024 @64x128 partial_spectral', 256, 1024, 1, 64, 128, 0, 'partial_spectral'),
    ('1x1 1024->256 @64x128 partial_spectral', 1024, 256, 1, 64, 128, 0, 'partial_spectral'),
    ('1x1 128->512 @128x256 partial_spectral', 128, 512, 1, 128, 256, 0, 'partial_spectral'),
    ('3x3 1024->1024 @32x64 spectral', 1024, 1024, 3, 32, 64, 1, 'spectral'),
]
dtype = torch.bfloat16


def timed(fn, reps=20):
  for _ in range(3):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


for name, cin, cout, k, h, w, pad, kind in shapes:
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, 1, 'VALID', kind.startswith('partial'), kind)
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  sg.power_iteration(True)
  x = nn.Var(torch.randn((N, h, w, cin), device=DEV).to(dtype))
  mask = (torch.rand((N, h, w), device=DEV) < 0.9).float() if kind.startswith('partial') else None
  flops = 2.0 * N * h * w * cin * cout * k * k
  res = {}
  for what in ('fwd', 'dgrad'):
    for mode in ('0', '1', '2'):
      os.environ['SE3DS_PROBE_MODE'] = '0'
      ctx = nn.Ctx(DEV, dtype, training=True, record=True)
      out = nn.conv2d(ctx, x, layer, pad=pad, mask=mask)
      y = out[0] if isinstance(out, tuple) else out
      if what == 'fwd':
        def run():
          c2 = nn.Ctx(DEV, dtype, training=True, record=True)
          nn.conv2d(c2, x, layer, pad=pad, mask=mask)
      else:
        ctx.param_grads = False
        g = torch.randn(y.data.shape, device=DEV).to(dtype)
        tape = list(ctx.tape)
        def run():
          y.grad = g
          x.grad = None
          for fn, _, _ in reversed(tape):
            fn()
      os.environ['SE3DS_PROBE_MODE'] = mode
      res[(what, mode)] = timed(run)
  os.environ['SE3DS_PROBE_MODE'] = '0'
  print('%-42s ' % name + '  '.join(
      '%s: all %6.1f us (%4.0f TF/s)  no-epilogue %6.1f  no-K-loop %6.1f' % (
          wh, res[(wh, '0')], flops / res[(wh, '0')] / 1e6, res[(wh, '1')], res[(wh, '2')])
      for wh in ('fwd', 'dgrad')))
