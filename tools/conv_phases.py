"""Where a conv kernel's time goes: the same launch with its epilogue or its K loop compiled out
(a -DSE3DS_PROBE build of conv.hip, tools/probes/conv_phases.sh; never the shipped library).
  SE3DS_PROBE_MODE=0 whole kernel, 1 no epilogue (prologue + K loop), 2 no K loop (prologue + epilogue),
  3 whole kernel with non-temporal output stores
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


from se3ds_amd import _lib
L = _lib.lib()
for name, cin, cout, k, h, w, pad, kind in shapes:
  store = nn.ParamStore()
  partial = kind.startswith('partial')
  layer = nn.ConvLayer(store, 'c', cin, cout, k, 1, 'VALID', partial, kind)
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  sg.power_iteration(True)
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  x = torch.randn((N, h, w, cin), device=DEV).to(dtype)
  mask = (torch.rand((N, h, w), device=DEV) < 0.9).float() if partial else None
  wt, wn = layer.operands(ctx)
  ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
  ratio = um = None
  if partial:
    ratio, um, ru, bu = nn.mask_window(ctx, mask, N, h, w, ho, wo, k, 1, pad, pad, False, True)
  scale = layer.sn['sig'][1:] if kind == 'spectral' else None
  y = torch.empty((N, ho, wo, cout), device=DEV, dtype=dtype)
  dx = torch.empty_like(x)
  dy = torch.randn((N, ho, wo, cout), device=DEV).to(dtype)
  rows = int(L.se3ds_conv2d_fwd_stats_rows(3, N, cin, ho, wo, cout, k, k, 1, 1 if partial else 0, 1))
  stats = torch.empty((max(rows, 1), 2, cout), device=DEV)
  s_ = _lib.stream()
  def fwd():   # the C entry points directly: ~3 us of host time per call, far below the kernels
    if rows > 0:
      L.se3ds_conv2d_fwd_stats(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 3, N, h, w, cin, ho, wo, cout,
                               k, k, 1, pad, pad, 0, _lib.ptr(mask), 1, _lib.ptr(scale), _lib.ptr(layer.bias),
                               _lib.ptr(ratio), _lib.ptr(um if partial else None), 0, 0.0, stats.data_ptr(), s_)
    else:
      L.se3ds_conv2d_fwd(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 3, N, h, w, cin, ho, wo, cout, k, k, 1,
                         pad, pad, 0, _lib.ptr(mask), 1, _lib.ptr(scale), _lib.ptr(layer.bias),
                         _lib.ptr(ratio), _lib.ptr(um if partial else None), 0, 0.0, s_)
  def dgrad():   # (a partial conv's dy is pre-scaled: no row scale in the data gradient)
    L.se3ds_conv2d_dgrad(dy.data_ptr(), wn.data_ptr(), dx.data_ptr(), 3, N, h, w, cin, ho, wo, cout, k, k,
                         1, pad, pad, 0, None, _lib.ptr(scale), None, _lib.ptr(mask), 0, 0.0, s_)
  def dgrad_acc():   # ... added into an existing gradient (the residual stream's)
    L.se3ds_conv2d_dgrad_acc(dy.data_ptr(), wn.data_ptr(), dx.data_ptr(), 3, N, h, w, cin, ho, wo, cout, k,
                             k, 1, pad, pad, 0, None, _lib.ptr(scale), None, _lib.ptr(mask), 0, 0.0,
                             dx.data_ptr(), s_)
  flops = 2.0 * N * h * w * cin * cout * k * k
  parts = []
  for what, fn in (('fwd', fwd), ('dgrad', dgrad), ('dgrad_acc', dgrad_acc)):
    t = {}
    for mode in ('0', '1', '2', '3'):
      os.environ['SE3DS_PROBE_MODE'] = mode
      t[mode] = timed(fn, 50)
    parts.append('%s: all %6.1f us (%4.0f TF/s) no-epilogue %6.1f no-K-loop %6.1f nt-stores %6.1f' % (
        what, t['0'], flops / t['0'] / 1e6, t['1'], t['2'], t['3']))
  os.environ['SE3DS_PROBE_MODE'] = '0'
  print('%-40s ' % name + ' | '.join(parts))
