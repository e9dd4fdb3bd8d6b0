"""Production-shape parity (SURVEY 8d dominant GEMM shapes, BASELINE cfg3 widths): every conv
layer shape that carries the bench step, at the real channel counts / resolutions, through the
DEFAULT dispatch (no env overrides: the kernels tested are the kernels the bench runs), forward +
data gradient + weight gradient vs the PyTorch-CPU oracle (oracle/nets_torch.py).

Inputs are bf16-representable, so one oracle run serves the fp32 and the bf16 path.  The oracle
runs in batch chunks (convolutions are per-sample; weight gradients add up), which bounds host
memory at the 512x1024 shapes.

Tolerances (max |a-b| / max |b| per tensor):
  fp32 path                       : 1e-4   (north_star: 1e-3; measured ~1e-6)
  bf16 path, bf16-stored tensors  : 6e-3   (one bf16 rounding step is 2^-9 = 2e-3 of the value)
  bf16 path, fp32-stored tensors  : 1e-4   (weight / bias gradients: exact bf16 products, fp32
                                            accumulation) -- except where the product rounds a
                                            per-pixel-scaled dy to bf16 first (partial convs with
                                            a real mask): 4e-3, stated per case.
"""
import numpy as np
import pytest
import torch

from oracle import nets_torch as O
from se3ds_amd.hipops import nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

TOL_F32 = 1e-4
TOL_BF16_STORED = 6e-3
TOL_BF16_F32OUT = 1e-4
TOL_BF16_F32OUT_SCALED_DY = 4e-3


def rel_err(a, b):
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def _bf(t):
  return t.bfloat16().float()


# name, kind, cin, cout, k, stride, padding, pad, bias, mask, h, w, batches, oracle chunk
# (h, w are INPUT sizes; cfg3 = 512x1024 panoramas, ResNet-101, gen_dims 128, dis_dims 128)
PROD_CONVS = [
    # decoder deconv1: 90 of these per generator forward, 59 % of the step's FLOPs
    ('deconv1 3x3 1024', 'spectral', 1024, 1024, 3, 1, 'VALID', 1, False, False, 32, 64, (2, 8), 8),
    # heads' 128->128 at full resolution (10.5 %)
    ('head 3x3 128 @512', 'spectral', 128, 128, 3, 1, 'VALID', 1, True, False, 512, 1024, (2, 8), 2),
    # decoder final_conv, plain Conv2D (7.9 %)
    ('final_conv 3x3 128 @256', 'plain', 128, 128, 3, 1, 'VALID', 1, False, False, 256, 512, (2, 8), 2),
    ('deconv4 3x3 128 @128', 'spectral', 128, 128, 3, 1, 'VALID', 1, False, False, 128, 256, (2, 8), 4),
    ('deconv2 3x3 512', 'spectral', 512, 512, 3, 1, 'VALID', 1, False, False, 32, 64, (2, 8), 8),
    ('deconv3 3x3 256', 'spectral', 256, 256, 3, 1, 'VALID', 1, False, False, 64, 128, (2, 8), 8),
    # encoder stack3 bottleneck (partial conv with a real mask; 3x3 and both 1x1)
    ('stack3 3x3 512 partial', 'partial_spectral', 512, 512, 3, 1, 'VALID', 1, True, True, 32, 64, (2, 8), 8),
    ('stack3 1x1 512->2048', 'partial_spectral', 512, 2048, 1, 1, 'SAME', 0, True, True, 32, 64, (2, 8), 8),
    ('stack3 1x1 2048->512', 'partial_spectral', 2048, 512, 1, 1, 'SAME', 0, True, True, 32, 64, (2, 8), 8),
    ('stack3 ds 1x1 s2 1024->2048', 'partial_spectral', 1024, 2048, 1, 2, 'SAME', 0, False, True, 64, 128, (2, 8), 8),
    ('stack2 3x3 s2 256 partial', 'partial_spectral', 256, 256, 3, 2, 'VALID', 1, True, True, 128, 256, (2, 8), 8),
    ('encoder final 3x3 4096->512', 'partial', 4096, 512, 3, 1, 'VALID', 1, True, True, 16, 32, (2, 8), 8),
    ('agent3 1x1 2048->512', 'partial_spectral', 2048, 512, 1, 1, 'SAME', 0, False, False, 32, 64, (8,), 8),
    ('context 3x3 512->1024', 'spectral', 512, 1024, 3, 1, 'VALID', 1, True, False, 16, 32, (8,), 8),
    ('upc 1x1 512->256', 'spectral', 512, 256, 1, 1, 'SAME', 0, True, False, 16, 32, (8,), 8),
    # first layer: 7x7 stride 2, 5 -> 128 with the projection mask, 512x1024 input
    ('conv1 7x7 s2 5->128', 'partial', 5, 128, 7, 2, 'VALID', 3, True, True, 512, 1024, (2, 8), 2),
    # heads' output convs at full resolution
    ('rgb head 3x3 128->3', 'spectral', 128, 3, 3, 1, 'VALID', 1, True, False, 512, 1024, (2, 8), 2),
    ('depth head 3x3 128->1', 'spectral', 128, 1, 3, 1, 'VALID', 1, True, False, 512, 1024, (2, 8), 2),
    # discriminator (batch = 2 x samples: [fake; real]); 4x4 stride 2, pad 2
    ('D g0 4x4 s2 4->128', 'plain', 4, 128, 4, 2, 'VALID', 2, True, False, 512, 1024, (4, 16), 4),
    ('D g1 4x4 s2 128->256', 'spectral', 128, 256, 4, 2, 'VALID', 2, True, False, 257, 513, (4, 16), 4),
    ('D g2 4x4 s2 256->512', 'spectral', 256, 512, 4, 2, 'VALID', 2, True, False, 129, 257, (4, 16), 4),
    ('D g3 4x4 s2 512->512', 'spectral', 512, 512, 4, 2, 'VALID', 2, True, False, 65, 129, (4, 16), 8),
    ('D g5 4x4 s1 512->512', 'spectral', 512, 512, 4, 1, 'VALID', 2, True, False, 17, 33, (16,), 16),
    ('D final 4x4 512->1', 'plain', 512, 1, 4, 1, 'SAME', 0, True, False, 18, 34, (16,), 16),
    ('D1 g0 4x4 s2 4->128 @256', 'plain', 4, 128, 4, 2, 'VALID', 2, True, False, 256, 512, (16,), 8),
    # cfg1 (128x256 panoramas, batch 2): the same channel counts on TINY maps -- fewer pixels than
    # one pixel tile / one reduction step (ragged tiles, single-step weight gradients)
    ('cfg1 deconv1 3x3 1024 @8x16', 'spectral', 1024, 1024, 3, 1, 'VALID', 1, False, False, 8, 16, (1, 2), 2),
    ('cfg1 deconv1 3x3 1024 @4x8', 'spectral', 1024, 1024, 3, 1, 'VALID', 1, False, False, 4, 8, (1,), 1),
    ('cfg1 deconv2 3x3 512 @4x8', 'spectral', 512, 512, 3, 1, 'VALID', 1, False, False, 4, 8, (1, 2), 2),
    ('cfg1 stack3 1x1 512->2048 @8x16', 'partial_spectral', 512, 2048, 1, 1, 'SAME', 0, True, True, 8, 16, (2,), 2),
    ('cfg1 stack3 1x1 2048->512 @8x16', 'partial_spectral', 2048, 512, 1, 1, 'SAME', 0, True, True, 8, 16, (2,), 2),
    ('cfg1 stack4 3x3 s2 1024 @8x16', 'partial_spectral', 1024, 1024, 3, 2, 'VALID', 1, True, True, 8, 16, (2,), 2),
    ('cfg1 stack4 1x1 1024->4096 @4x8', 'partial_spectral', 1024, 4096, 1, 1, 'SAME', 0, True, True, 4, 8, (2,), 2),
    ('cfg1 encoder final 3x3 4096->512 @4x8', 'partial', 4096, 512, 3, 1, 'VALID', 1, True, True, 4, 8, (2,), 2),
    ('cfg1 agent3 1x1 2048->512 @8x16', 'partial_spectral', 2048, 512, 1, 1, 'SAME', 0, False, False, 8, 16, (2,), 2),
]

_ORACLE_CACHE = {}


def _inputs(case, n):
  name, kind, cin, cout, k, stride, padding, pad, bias, use_mask, h, w, _, _ = case
  gen = torch.Generator().manual_seed(hash((cin, cout, k, h, w, n)) % (2 ** 31))
  x = _bf(torch.randn((n, h, w, cin), generator=gen))
  kern = _bf(nn.glorot_uniform((k, k, cin, cout), gen))
  b = (torch.randn(cout, generator=gen) * 0.1) if bias else None
  u = nn.truncated_normal_init((1, cout), gen)
  mask = None
  if use_mask:
    # random holes + a zeroed band (indoor_datasets.py:281-304) + a hole-free region
    mask = (torch.rand((n, h, w, 1), generator=gen) > 0.3).float()
    mask[:, h // 3:h // 3 + max(1, h // 8)] = 0
    mask[:, :, : w // 4] = 1
  ho = nn.conv_out_size(h, k, stride, padding, pad)[0]
  wo = nn.conv_out_size(w, k, stride, padding, pad)[0]
  gy = _bf(torch.randn((n, ho, wo, cout), generator=gen))
  return x, kern, b, u, mask, gy


def _oracle(case, n, inputs=None):
  """Chunked oracle: y, dx, dK, db, update_mask.  inputs: explicit (x, kern, b, u, mask, gy)
  instead of the seeded ones (golden fixtures)."""
  key = (case[0], n)
  if inputs is None and key in _ORACLE_CACHE:
    return _ORACLE_CACHE[key]
  name, kind, cin, cout, k, stride, padding, pad, bias, use_mask, h, w, _, chunk = case
  x, kern, b, u, mask, gy = inputs if inputs is not None else _inputs(case, n)
  ko = kern.clone().requires_grad_(True)
  p = {'c/kernel': ko, 'c/u': u}
  if bias:
    p['c/bias'] = b.clone().requires_grad_(True)
  ys, dxs, ums = [], [], []
  db64 = torch.zeros(cout, dtype=torch.float64) if (bias and kind.startswith('partial')) else None
  for c0 in range(0, n, chunk):
    xo = x[c0:c0 + chunk].clone().requires_grad_(True)
    net = O.Net(p, training=True)
    xin = O.pad_layer(xo, pad, circular_pad=False, training=True) if pad else xo
    if kind.startswith('partial'):
      m_in = None
      if mask is not None:
        m = mask[c0:c0 + chunk]
        m_in = O.pad_layer(m, pad, circular_pad=False, training=True) if pad else m
      yo, umo = net.partial_conv(xin, m_in, 'c', stride, padding, spectral=kind == 'partial_spectral')
      ums.append(umo.detach())
      if db64 is not None:
        # y = ((conv - b) * ratio + b) * um  =>  dy/db = (1 - ratio) * um (layers.py:199-203).  Where
        # the window is full, 1 - ratio is ~1e-6 and autograd's fp32 "sum(dy um) - sum(dy um ratio)"
        # cancels to noise (15 % off on the 1x1 bottlenecks); the derivative is evaluated in
        # binary64 from the oracle's own fp32 ratio / update_mask instead.
        ones = torch.ones((chunk if m_in is None else m_in.shape[0], xin.shape[1], xin.shape[2], 1))
        raw = O.tf_conv2d(ones if m_in is None else m_in, torch.ones((k, k, 1, 1)), stride, padding)
        ratio = (k * k) / (raw + 1e-6) * torch.clamp(raw, 0, 1)
        wgt = (1.0 - ratio.double()) * umo.detach().double()
        db64 += (gy[c0:c0 + chunk].double() * wgt).sum(dim=(0, 1, 2))
    elif kind == 'spectral':
      yo = net.spectral_conv(xin, 'c', stride, padding)
    else:
      yo = net.conv2d(xin, 'c', stride, padding)
    yo.backward(gy[c0:c0 + chunk])
    ys.append(yo.detach())
    dxs.append(xo.grad)
  res = dict(y=torch.cat(ys).numpy(), dx=torch.cat(dxs).numpy(), dk=ko.grad.numpy(),
             db=(db64.numpy() if db64 is not None else p['c/bias'].grad.numpy()) if bias else None,
             um=torch.cat(ums).numpy()[..., 0] if ums else None)
  if inputs is None:
    _ORACLE_CACHE.clear()   # keep one entry: the fp32 and bf16 variants of a case run back to back
    _ORACLE_CACHE[key] = res
  return res


def _hip(case, n, dtype, prior_grad=None, inputs=None):
  """prior_grad: an existing gradient of x (a ResNet block input that already received its
  residual-branch gradient): the data-gradient kernel then ADDS in its epilogue
  (se3ds_conv2d_dgrad_acc) instead of writing a fresh tensor."""
  name, kind, cin, cout, k, stride, padding, pad, bias, use_mask, h, w, _, _ = case
  x, kern, b, u, mask, gy = inputs if inputs is not None else _inputs(case, n)
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, stride, padding, bias, kind)
  store.finalize(DEV, None)
  d = {'c/kernel': kern.numpy()}
  if bias:
    d['c/bias'] = b.numpy()
  if layer.spectral:
    d['c/u'] = u.numpy()
  store.load_dict(d)
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  sg.power_iteration(training=False)
  xv = nn.Var(x.to(DEV).to(dtype), requires_grad=True)
  mdev = mask[..., 0].contiguous().to(DEV) if mask is not None else None
  res = nn.conv2d(ctx, xv, layer, pad=pad, wrap=False, mask=mdev)
  um = None
  if kind.startswith('partial'):
    yv, um = res
  else:
    yv = res
  y = yv.data.float().cpu().numpy()
  yv.grad = gy.to(DEV).to(dtype)
  if prior_grad is not None:
    xv.grad = prior_grad.to(DEV).to(dtype).clone()
  ctx.backward()
  sg.backward_fixup()
  torch.cuda.synchronize()
  return dict(y=y, dx=xv.grad.float().cpu().numpy(), dk=store.grad_views['c/kernel'].cpu().numpy(),
              db=store.grad_views['c/bias'].cpu().numpy() if bias else None,
              um=um.cpu().numpy() if um is not None else None)


def _check(case, n, dtype):
  ref = _oracle(case, n)
  got = _hip(case, n, dtype)
  use_mask = case[9]
  if dtype == torch.float32:
    t_act = t_par = TOL_F32
  else:
    t_act = TOL_BF16_STORED
    t_par = TOL_BF16_F32OUT_SCALED_DY if use_mask else TOL_BF16_F32OUT
  errs = {}
  if ref['um'] is not None:
    np.testing.assert_array_equal(got['um'], ref['um'])   # update mask is {0,1}: exact
  for key, t in (('y', t_act), ('dx', t_act), ('dk', t_par), ('db', t_par)):
    if ref[key] is None:
      continue
    errs[key] = rel_err(got[key], ref[key])
  print(f'{case[0]} n{n} {str(dtype)[6:]}: ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()))
  for key, t in (('y', t_act), ('dx', t_act), ('dk', t_par), ('db', t_par)):
    if key in errs:
      assert errs[key] < t, (case[0], n, dtype, key, errs[key], t)


def _case_ids():
  out = []
  for c in PROD_CONVS:
    for n in c[12]:
      out.append(pytest.param(c, n, id=f'{c[0]} n{n}'.replace(' ', '_')))
  return out


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case,n', _case_ids())
def test_prod_conv_fwd_dgrad_wgrad(case, n, dtype):
  if dtype == torch.float32 and n > case[12][0] and not case[0].startswith('cfg1'):
    pytest.skip('fp32 path: smallest batch only (fp32 MFMA peak is 16x lower; same kernels)')
  _check(case, n, dtype)


# ------------------------------------------------------------------------- transposed convs
# name, k, cin, cout, bias, h, w, batches, chunk   (decoder upsampling layers, cfg3 sizes)
PROD_CONVT = [
    ('deconv2 convT k3 512->256', 3, 512, 256, False, 32, 64, (2, 8), 8),
    ('deconv2 up convT k2 512->256', 2, 512, 256, False, 32, 64, (8,), 8),
    ('deconv3 convT k3 256->128', 3, 256, 128, False, 64, 128, (8,), 8),
    ('deconv4 convT k3 128->128', 3, 128, 128, False, 128, 256, (2, 8), 4),
    ('deconv4 up convT k2 128->128', 2, 128, 128, False, 128, 256, (8,), 4),
    ('final_deconv k2 128->128 bias', 2, 128, 128, True, 256, 512, (2, 8), 2),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case,n', [pytest.param(c, n, id=f'{c[0]} n{n}'.replace(' ', '_'))
                                    for c in PROD_CONVT for n in c[7]])
def test_prod_conv_transpose(case, n, dtype):
  name, k, cin, cout, bias, h, w, batches, chunk = case
  if dtype == torch.float32 and n > batches[0]:
    pytest.skip('fp32 path: smallest batch only')
  gen = torch.Generator().manual_seed(1000 + k * cin + h + n)
  x = _bf(torch.randn((n, h, w, cin), generator=gen))
  kern = _bf(nn.glorot_uniform((k, k, cout, cin), gen))
  b = (torch.randn(cout, generator=gen) * 0.1) if bias else None
  gy = _bf(torch.randn((n, 2 * h, 2 * w, cout), generator=gen))
  ko = kern.clone().requires_grad_(True)
  bo = b.clone().requires_grad_(True) if bias else None
  ys, dxs = [], []
  for c0 in range(0, n, chunk):
    xo = x[c0:c0 + chunk].clone().requires_grad_(True)
    yo = O.keras_conv2d_transpose(xo, ko, bo, 2)
    yo.backward(gy[c0:c0 + chunk])
    ys.append(yo.detach())
    dxs.append(xo.grad)
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, 2, 'SAME', bias, 'plain', transpose=True)
  store.finalize(DEV, None)
  d = {'c/kernel': kern.numpy()}
  if bias:
    d['c/bias'] = b.numpy()
  store.load_dict(d)
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  xv = nn.Var(x.to(DEV).to(dtype), requires_grad=True)
  yv = nn.conv_transpose2d(ctx, xv, layer)
  yh = yv.data.float().cpu().numpy()
  yv.grad = gy.to(DEV).to(dtype)
  ctx.backward()
  t_act = TOL_F32 if dtype == torch.float32 else TOL_BF16_STORED
  t_par = TOL_F32 if dtype == torch.float32 else TOL_BF16_F32OUT
  e = dict(y=rel_err(yh, torch.cat(ys).numpy()),
           dx=rel_err(xv.grad.float().cpu().numpy(), torch.cat(dxs).numpy()),
           dk=rel_err(store.grad_views['c/kernel'].cpu().numpy(), ko.grad.numpy()))
  if bias:
    e['db'] = rel_err(store.grad_views['c/bias'].cpu().numpy(), bo.grad.numpy())
  print(f'{name} n{n} {str(dtype)[6:]}: ' + ' '.join(f'{k_}={v:.2e}' for k_, v in e.items()))
  assert e['y'] < t_act and e['dx'] < t_act and e['dk'] < t_par, (name, n, dtype, e)
  if bias:
    assert e['db'] < t_par, (name, e)


ACC_CASES = ['cfg1 deconv1 3x3 1024 @8x16', 'cfg1 stack3 1x1 512->2048 @8x16', 'cfg1 deconv2 3x3 512 @4x8',
             'deconv1 3x3 1024', 'stack3 3x3 512 partial', 'stack3 1x1 512->2048',
             'stack3 1x1 2048->512', 'stack2 3x3 s2 256 partial', 'final_conv 3x3 128 @256',
             'D g2 4x4 s2 256->512']


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('name', ACC_CASES)
def test_prod_conv_dgrad_accumulates_into_existing_gradient(name, dtype):
  """se3ds_conv2d_dgrad_acc at production shapes: the input already holds a gradient (the residual
  branch of a ResNet block), the data-gradient epilogue adds to it in place."""
  case = next(c for c in PROD_CONVS if c[0] == name)
  n = case[12][0]
  ref = _oracle(case, n)
  g0 = _bf(torch.randn(ref['dx'].shape, generator=torch.Generator().manual_seed(5)))
  got = _hip(case, n, dtype, prior_grad=g0)
  want = ref['dx'] + g0.numpy()
  e = rel_err(got['dx'], want)
  print(f'{name} n{n} {str(dtype)[6:]}: dx(acc)={e:.2e}')
  # bf16: the sum is rounded once more when it is stored
  assert e < (TOL_F32 if dtype == torch.float32 else 2 * TOL_BF16_STORED), (name, e)


# ------------------------------------------------------------------------------ norm layers
# kind, n, h, w, c, act, with_res  (cfg3 / cfg1 tensor shapes of the generator's batch norms and the
# discriminator's instance norms; training mode = batch statistics)
PROD_NORMS = [
    ('batch', 2, 8, 16, 1024, 1, True),      # cfg1 deconv1 (256 samples per channel)
    ('batch', 2, 8, 16, 2048, 1, True),      # cfg1 stack3 bn3
    ('batch', 2, 4, 8, 4096, 1, True),       # cfg1 stack4 bn3 (64 samples per channel)
    ('batch', 8, 32, 64, 1024, 1, True),     # cfg3 deconv1
    ('batch', 2, 128, 256, 128, 1, True),    # cfg1 final_conv / cfg3 deconv4 class
    ('batch', 2, 512, 1024, 128, 0, False),  # cfg3 head batch norm (no activation)
    ('batch', 2, 64, 128, 512, 2, False),    # upc-style LeakyReLU
    ('instance', 4, 129, 257, 256, 2, False),  # discriminator group 1
    ('instance', 4, 17, 33, 512, 2, False),    # discriminator group 4
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', PROD_NORMS, ids=[f'{c[0]}_{c[1]}x{c[2]}x{c[3]}x{c[4]}' for c in PROD_NORMS])
def test_prod_norm_fwd_bwd(case, dtype):
  kind, n, h, w, c, act, with_res = case
  store = nn.ParamStore()
  layer = nn.NormLayer(store, 'n', c, kind)
  store.finalize(DEV, None)
  gen = torch.Generator().manual_seed(3 + c + h)
  store.load_dict({'n/gamma': (torch.rand(c, generator=gen) + 0.5).numpy(),
                   'n/beta': (torch.randn(c, generator=gen) * 0.2).numpy()})
  # channel means far from zero relative to the spread (as behind a ReLU + residual stream)
  x = _bf(torch.randn((n, h, w, c), generator=gen) * 0.7 + torch.randn(c, generator=gen) * 2.0)
  r = _bf(torch.randn((n, h, w, c), generator=gen))
  gy = _bf(torch.randn((n, h, w, c), generator=gen))
  alpha = 0.2
  p = {k: v.cpu().clone() for k, v in store.views.items()}
  p['n/gamma'].requires_grad_(True)
  p['n/beta'].requires_grad_(True)
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  xv = nn.Var(x.to(DEV).to(dtype))
  rv = nn.Var(r.to(DEV).to(dtype)) if with_res else None
  yv = nn.norm_act(ctx, xv, layer, act=act, alpha=alpha, res=rv)
  y_h = yv.data.float().cpu()
  xo, ro = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
  net = O.Net(p, training=True)
  pre = net.sync_bn(xo, 'n') if kind == 'batch' else net.instance_norm(xo, 'n')
  if with_res:
    pre = pre + ro
  # The derivative of ReLU / LeakyReLU is a DECISION on the sign of the pre-activation.  Two correct
  # fp32 evaluations of var = E[x^2] - E[x]^2 differ by ~1e-6 relative, so on a 16 M-element tensor a
  # few dozen pre-activations straddle zero and the two backward passes differ there by a whole
  # |dy| (0.25 in max-norm at 8x32x64x1024 -- measured, both implementations correct).  The test
  # therefore (1) bounds where decisions may differ -- only inside the forward error band around
  # zero, which is a theorem for correct arithmetic -- and (2) runs the oracle's backward under the
  # HIP path's decisions, after which everything is linear algebra and must agree tightly.
  if act == 0:
    yo_own = pre
    yo = pre
    flips = 0
  else:
    slope = 0.0 if act == 1 else alpha
    pos_h = y_h > 0
    pos_o = pre.detach() > 0
    yo_own = torch.where(pos_o, pre.detach(), pre.detach() * slope)
    yo = torch.where(pos_h, pre, pre * slope)
    flip = pos_h != pos_o
    flips = int(flip.sum())
    band = float((y_h - yo_own).abs().max()) * 1.01 + 1e-30
    assert not bool((flip & (pre.detach().abs() > band)).any()), 'sign decision outside the error band'
    assert flips <= 2e-4 * flip.numel(), flips
  yo.backward(gy)
  t_act = TOL_F32 if dtype == torch.float32 else 2 * TOL_BF16_STORED
  t_par = 2e-4 if dtype == torch.float32 else 1e-2
  e = dict(y=rel_err(y_h.numpy(), yo_own.detach().numpy()))
  if kind == 'batch':
    for nm in ('moving_mean', 'moving_variance'):
      assert rel_err(store['n/' + nm].cpu().numpy(), net.updates['n/' + nm].numpy()) < 1e-5, nm
  yv.grad = gy.to(DEV).to(dtype)
  ctx.backward()
  e['dx'] = rel_err(xv.grad.float().cpu().numpy(), xo.grad.numpy())
  e['dgamma'] = rel_err(store.grad_views['n/gamma'].cpu().numpy(), p['n/gamma'].grad.numpy())
  e['dbeta'] = rel_err(store.grad_views['n/beta'].cpu().numpy(), p['n/beta'].grad.numpy())
  if with_res:
    e['dres'] = rel_err(rv.grad.float().cpu().numpy(), ro.grad.numpy())
  print(f'{case} {str(dtype)[6:]}: flips={flips} ' + ' '.join(f'{k}={v:.2e}' for k, v in e.items()))
  assert e['y'] < t_act and e['dx'] < 2 * t_act, e
  assert e['dgamma'] < t_par and e['dbeta'] < t_par, e
  if with_res:
    assert e['dres'] < t_act, e
