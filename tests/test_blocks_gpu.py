"""Block-level forward + backward parity at PRODUCTION width through the default dispatch.

One residual block (or decoder stage, or head) is short enough to be well conditioned even with
batch statistics, so -- unlike the 200-layer cfg1 step, whose fp32 gradients are only comparable
through an fp64 yardstick -- every tensor is held DIRECTLY to north_star's bar:

  fp32 path : max|hip - oracle| <= 1e-3 * max|oracle|  for the output, every input gradient and
              every parameter gradient (kernel, bias, gamma, beta; spectral fix-up applied)
  bf16 path : <= 2e-2 on the same tensors (operands rounded to bf16, fp32 accumulation); 5e-2 on
              per-channel parameter gradients (bias, gamma, beta: sums of N*H*W cancelling terms)

Blocks (reference lines; widths and map sizes are those of highres.gin at 512 x 1024, batch 2):
  Bottleneck       layers.py:220-272  stack3 block0: 1024 -> 512 -> 2048, stride 2, real mask,
                                      1x1 strided downsample (64x128 -> 32x64)
  TransBasicBlock  layers.py:400-455  deconv1 block: 1024 @32x64 (two 3x3 1024 -> 1024)
  upsampling block layers.py:400-455,472-480  deconv2 last block: 512 -> 256, ConvT k3 s2 +
                                      ConvT k2 s2 residual (32x64 -> 64x128)
  upc + agent4, agent3 + add   image_models.py:351-441,455-462
  head             image_models.py:79-104  128 -> 128 -> 128 -> 3 at 128x256, then (tanh+1)/2
  patch discriminator  image_models.py:492-561  one scale of the discriminator, dis_dims 128 x 6
                                      layers, on a 256x512 RGB-D input (input gradient included)

ReLU / LeakyReLU derivatives are sign DECISIONS: two correct evaluations of a pre-activation
differ by ~1e-6 relative (fp32), so on a 4 M-element tensor one or two elements straddle zero and
the backward passes then differ by a whole |dy| there although both are right
(tests/test_prod_shapes_gpu.py::test_prod_norm_fwd_bwd measured it).  As there, the oracle's
backward runs under the device path's decisions (oracle.nets_torch.Net.decisions); the test
proves that the decisions differ from the oracle's own only inside the forward error band around
zero and only on a vanishing fraction of the elements.
"""
import numpy as np
import pytest
import torch

from oracle import nets_torch as O
from se3ds_amd.hipops import nn
from se3ds_amd.hipops.nn import ACT_LRELU, ACT_RELU
from se3ds_amd.models import image_models, layers

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}
FLIP_FRAC = {torch.float32: 2e-5, torch.bfloat16: 1e-2}


def _bf(t):
  return t.to(torch.bfloat16).to(torch.float32)


def _mask(n, h, w, gen):
  m = (torch.rand((n, h, w), generator=gen) < 0.7).float()
  m[:, h // 3:h // 3 + max(1, h // 6)] = 0      # a band of holes as the projected masks have
  m[:, :, w // 2:w // 2 + w // 8] *= (torch.rand((n, h, w // 8), generator=gen) < 0.2).float()
  return m


def _randomise(store, gen):
  upd = {}
  for nme in store.trainable_names:
    shape = store[nme].shape
    if nme.endswith('gamma'):
      upd[nme] = (torch.rand(shape, generator=gen) + 0.5).numpy()
    elif nme.endswith('beta') or nme.endswith('bias'):
      upd[nme] = (torch.randn(shape, generator=gen) * 0.1).numpy()
  store.load_dict(upd)


class _Case:
  """build(store) -> module-ish; hip(ctx, mod, vars, masks) -> Var; ref(net, xs, masks) -> tensor."""

  def __init__(self, name, shapes, mask_shape, build, hip, ref, n_decisions):
    self.name, self.shapes, self.mask_shape = name, shapes, mask_shape
    self.build, self.hip, self.ref, self.n_decisions = build, hip, ref, n_decisions


def _bottleneck_case():
  def build(store):
    ds = layers.PartialSpectralConv(store, 'st/downsample', 1024, 2048, 1, 2, 'SAME', use_bias=False)
    return layers.Bottleneck(store, 'st/block0', 1024, 512, 2, 4, ds, True, layers.PartialSpectralConv)
  def hip(ctx, mod, xs, mask):
    out, um = mod(ctx, xs[0], mask)
    return out, um
  def ref(net, xs, mask):
    out, um = net.bottleneck(xs[0], mask[..., None], 'st/block0', 2, True, 'st/downsample', True, True)
    return out, um[..., 0]
  return _Case('bottleneck_stack3_block0', [(2, 64, 128, 1024)], (2, 64, 128), build, hip, ref, 3)


def _trans_block_case():
  def build(store):
    return layers.TransBasicBlock(store, 'dc/block0', 1024, 1024, circular_pad=True,
                                  conv_fn=layers.SpectralConv)
  def hip(ctx, mod, xs, mask):
    return mod(ctx, xs[0]), None
  def ref(net, xs, mask):
    return net.trans_basic_block(xs[0], 'dc/block0', 1, None, None, True, True), None
  return _Case('trans_basic_block_1024', [(2, 32, 64, 1024)], None, build, hip, ref, 2)


def _upsampling_block_case():
  def build(store):
    up = layers._Upsample(store, 'dc/upsample', 512, 256, 2, layers.SpectralConv)
    return layers.TransBasicBlock(store, 'dc/block3', 512, 256, 2, upsample=up, circular_pad=True,
                                  conv_fn=layers.SpectralConv)
  def hip(ctx, mod, xs, mask):
    return mod(ctx, xs[0]), None
  def ref(net, xs, mask):
    return net.trans_basic_block(xs[0], 'dc/block3', 2, 'convT', 'dc/upsample', True, True), None
  return _Case('upsampling_block_512_256', [(2, 32, 64, 512)], None, build, hip, ref, 2)


class _UpcAgents:
  """The decoder's entry and first skip merge (image_models.py:351-441,455-462), deconv1 left out:
  upc (SN 1x1 -> BN -> LeakyReLU 0.2 -> nearest x2) -> agent4, then + agent3(skip s3)."""

  def __init__(self, store):
    d = 128
    self.upc_conv = layers.SpectralConv(store, 'dec/upc/conv', d * 4, d * 2, 1, 1, 'SAME')
    self.upc_bn = layers.SyncBatchNormalization(store, 'dec/upc/bn', d * 2)
    self.agent4 = layers.PartialSpectralConv(store, 'dec/agent4', d * 2, d * 8, 1, 1, 'SAME', use_bias=False)
    self.agent4_bn = layers.SyncBatchNormalization(store, 'dec/agent4_bn', d * 8)
    self.agent3 = layers.PartialSpectralConv(store, 'dec/agent3', d * 16, d * 8, 1, 1, 'SAME', use_bias=False)
    self.agent3_bn = layers.SyncBatchNormalization(store, 'dec/agent3_bn', d * 8)

  def __call__(self, ctx, x, s3):
    out = self.upc_conv(ctx, x)
    out = self.upc_bn(ctx, out, act=ACT_LRELU, alpha=0.2)
    out = nn.upsample2x(ctx, out)
    y, _ = self.agent4(ctx, out, None)
    out = self.agent4_bn(ctx, y, act=ACT_RELU)
    y, _ = self.agent3(ctx, s3, None)
    return nn.add(ctx, out, self.agent3_bn(ctx, y, act=ACT_RELU))


def _upc_agents_case():
  def hip(ctx, mod, xs, mask):
    return mod(ctx, xs[0], xs[1]), None
  def ref(net, xs, mask):
    out = net.spectral_conv(xs[0], 'dec/upc/conv', 1, 'SAME')
    out = net.act(net.sync_bn(out, 'dec/upc/bn'), 'dec/upc/bn', 0.2)
    out = out.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    def agent(t, nm):
      y, _ = net.partial_conv(t, None, 'dec/' + nm, 1, 'SAME', True)
      return net.act(net.sync_bn(y, 'dec/' + nm + '_bn'), 'dec/' + nm + '_bn')
    return agent(out, 'agent4') + agent(xs[1], 'agent3'), None
  return _Case('upc_agent4_agent3_add', [(2, 16, 32, 512), (2, 32, 64, 2048)], None, _UpcAgents,
               hip, ref, 3)


def _head_case():
  def build(store):
    return image_models._Head(store, 'rgb_conv', 128, 3, True, layers.SpectralConv)
  def hip(ctx, mod, xs, mask):
    return mod(ctx, xs[0]), None
  def ref(net, xs, mask):
    return O.head(net, xs[0], 'rgb_conv', True), None
  return _Case('head_rgb_128', [(2, 128, 256, 128)], None, build, hip, ref, 2)


def _patch_discriminator_case():
  # one scale of the multi-scale PatchGAN discriminator at highres.gin's width and depth
  # (dis_dims 128, 6 layers: 4x4 stride-2 convs 4 -> 128 -> 256 -> 512 -> 512, a stride-1 512, the
  # 1-channel logits; spectral norm, InstanceNorm, LeakyReLU 0.2) on the second scale's input
  def build(store):
    return image_models.SNPatchDiscriminator(store, 'dis0', 4, 4, 128, 6, False)
  def hip(ctx, mod, xs, mask):
    return mod(ctx, xs[0])[-1], None
  def ref(net, xs, mask):
    return O.patch_discriminator(net, xs[0], 'dis0', 6)[-1], None
  return _Case('patch_discriminator_128x6', [(2, 256, 512, 4)], None, build, hip, ref, 6)


CASES = [_bottleneck_case(), _trans_block_case(), _upsampling_block_case(), _upc_agents_case(),
         _head_case(), _patch_discriminator_case()]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[c.name for c in CASES])
def test_block_fwd_bwd_at_production_width(case, dtype):
  gen = torch.Generator().manual_seed(11)
  store = nn.ParamStore()
  mod = case.build(store)
  store.finalize(DEV, torch.Generator().manual_seed(7))
  _randomise(store, gen)
  dev = torch.device(DEV)
  sg = nn.SpectralGroup(image_models._conv_layers_of(mod), dev)
  # activations as they arrive behind a ReLU + residual stream: non-negative, channel-dependent scale
  xs = [_bf(torch.relu(torch.randn(s, generator=gen) * 0.7 + torch.randn(s[-1], generator=gen) * 0.5))
        for s in case.shapes]
  mask = _mask(*case.mask_shape, gen) if case.mask_shape else None
  params = {k: v.detach().cpu().clone() for k, v in store.views.items()}   # before u advances

  # ---- device path
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  ctx.act_taps = {}
  xv = [nn.Var(x.to(DEV).to(dtype)) for x in xs]
  sg.power_iteration(True)
  out, um = case.hip(ctx, mod, xv, mask.to(DEV) if mask is not None else None)
  is_head = case.name.startswith('head')
  if is_head:
    y_h, push = nn.head(ctx, out, 0)
    y_h = y_h.cpu()
  else:
    y_h = out.data.float().cpu()
  gy = _bf(torch.randn(y_h.shape, generator=gen))
  if is_head:
    push(gy.to(DEV))
  else:
    out.grad = gy.to(DEV).to(dtype)
  ctx.backward()
  sg.backward_fixup()
  decisions = {k: (v.float() > 0).cpu() for k, v in ctx.act_taps.items()}
  assert len(decisions) == case.n_decisions, sorted(decisions)

  # ---- oracle, backward under the device path's sign decisions
  for k in store.trainable_names:
    params[k].requires_grad_(True)
  net = O.Net(params, training=True)
  net.decisions, net.pre_acts = decisions, {}
  xo = [x.clone().requires_grad_(True) for x in xs]
  yo, um_o = case.ref(net, xo, mask)
  if is_head:
    yo = (torch.tanh(yo) + 1) / 2
  yo.backward(gy)
  tol = TOL[dtype]
  flips_total = 0
  for tag, dec in decisions.items():
    pre = net.pre_acts[tag]
    flip = dec != (pre > 0)
    flips_total += int(flip.sum())
    assert float(flip.float().mean()) <= FLIP_FRAC[dtype], (tag, int(flip.sum()), flip.numel())
    # a decision may only differ where the pre-activation lies inside the forward error band
    band = tol * float(pre.abs().max())
    assert not bool((flip & (pre.abs() > band)).any()), (tag, 'sign decision outside the error band')

  def err(a, b, floor=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))

  e = {'y': err(y_h.numpy(), yo.detach().numpy())}
  if um is not None:
    assert torch.equal(um.cpu(), um_o.detach()), 'update mask must be exact'
  for i, (v, o) in enumerate(zip(xv, xo)):
    e[f'dx{i}'] = err(v.grad.float().cpu().numpy(), o.grad.numpy())
  # Parameter gradients are measured against max|oracle| of the tensor, floored at a fraction of
  # the largest gradient entry of the tensor's CLASS in this block (kernels / per-channel vectors).
  # The floor matters for per-channel vectors only: the bias of a conv that feeds a batch norm has
  # a mathematically ZERO gradient (the norm removes the mean; with a partial-conv mask or a
  # LeakyReLU in between, nearly zero), i.e. what both sides compute is the rounding residue of a
  # column sum over N*H*W rows whose natural scale is that of the non-cancelling column sums
  # next to it (dbeta, dgamma).  fp32: 1e-2 of that scale -- the cancelling sums are still pinned
  # to 1e-5 of the terms they add up; bf16 (dy rounded to 8 bits before the sum): 5e-2.
  names = store.trainable_names
  is_vec = {k: params[k].grad.dim() <= 1 for k in names}
  scale = {c: max([float(params[k].grad.abs().max()) for k in names if is_vec[k] == c] or [0.0])
           for c in (False, True)}
  floor = {False: 1e-4 * scale[False],
           True: (1e-2 if dtype == torch.float32 else 5e-2) * scale[True]}
  for k in names:
    fl = floor[is_vec[k]]
    if dtype == torch.bfloat16 and is_vec[k] and float(params[k].grad.abs().max()) < 1e-5 * scale[True]:
      # numerically ZERO in the oracle (a conv bias straight in front of a batch norm): what the
      # bf16 path returns is the rounding residue of summing N*H*W bf16-rounded terms; it must stay
      # below 2 % of the block's per-channel gradient scale (fp32 pins the same tensor to 1e-5)
      fl = scale[True]
    e[k] = err(store.grad_views[k].cpu().numpy(), params[k].grad.numpy(), fl)
  for k, v in net.updates.items():   # BN moving statistics, spectral u
    e['upd:' + k] = err(store[k].cpu().numpy(), v.detach().numpy())
  worst = sorted(e.items(), key=lambda kv: -kv[1])[:4]
  print(f'{case.name} {str(dtype)[6:]}: flips={flips_total} y={e["y"]:.2e} ' +
        ' '.join(f'{k}={v:.2e}' for k, v in worst))
  # bf16: per-channel vectors (sums over N*H*W rows of bf16-rounded terms that largely cancel)
  # get 5e-2; everything else -- activations, input gradients, kernels, state -- 2e-2
  vec_tol = tol if dtype == torch.float32 else 5e-2
  bad = {k: v for k, v in e.items() if not v <= (vec_tol if is_vec.get(k, False) else tol)}
  assert not bad, bad


@pytest.mark.parametrize('case', CASES, ids=[c.name for c in CASES])
def test_fused_backward_passes_match_separate_passes(case):
  """Round 3, two fusions of per-row / per-channel work into kernels that stream the data anyway:
  (a) SE3DS_FUSED_ROW_SCALE (default on): the batch norm behind a biased PARTIAL conv stores its dx
      already multiplied by ratio * update_mask and writes that conv's bias gradient
      (se3ds_norm_bwd_apply_rows) -- instead of the conv's own pass over dx
      (se3ds_colsum_row_scale, ~144 launches per step);
  (b) SE3DS_FUSED_BN_BWD (opt-in: measured slower at batch 8): where a batch norm's
      output feeds a stride-1 convolution, that convolution's data gradient takes the norm's
      backward statistics (sum dz, sum dz * xhat) from the gradient it stores
      (se3ds_conv2d_dgrad_bnstats) instead of a second pass over dy and x.
  Same block, same inputs, each switch against both off: (a) is bit-identical except for the bias
  gradient (a re-ordered fp32 sum); (b) agrees to the rounding of re-ordered fp32 sums (parameter
  gradients 2e-3 of their class scale, input gradients one bf16 ulp of the tensor's maximum); and
  the fused paths are really taken."""
  dtype = torch.bfloat16
  res = {}
  for cfg in ('plain', 'rows', 'bn_bwd'):
    gen = torch.Generator().manual_seed(11)
    store = nn.ParamStore()
    mod = case.build(store)
    store.finalize(DEV, torch.Generator().manual_seed(7))
    _randomise(store, gen)
    sg = nn.SpectralGroup(image_models._conv_layers_of(mod), torch.device(DEV))
    xs = [_bf(torch.relu(torch.randn(s, generator=gen) * 0.7 + torch.randn(s[-1], generator=gen) * 0.5))
          for s in case.shapes]
    mask = _mask(*case.mask_shape, gen) if case.mask_shape else None
    old = nn._FUSED_BN_BWD, nn._FUSED_ROW_SCALE, nn._NORM_DEBUG
    nn._FUSED_BN_BWD, nn._FUSED_ROW_SCALE, nn._NORM_DEBUG = cfg == 'bn_bwd', cfg == 'rows', {}
    try:
      ctx = nn.Ctx(DEV, dtype, training=True, record=True)
      xv = [nn.Var(x.to(DEV).to(dtype)) for x in xs]
      sg.power_iteration(True)
      out, _ = case.hip(ctx, mod, xv, mask.to(DEV) if mask is not None else None)
      if case.name.startswith('head'):
        y_h, push = nn.head(ctx, out, 0)
        push(_bf(torch.randn(y_h.shape, generator=gen)).to(DEV))
      else:
        out.grad = _bf(torch.randn(out.data.shape, generator=gen)).to(DEV).to(dtype)
      ctx.backward()
      sg.backward_fixup()
      counts = {t: sum(v for k, v in nn._NORM_DEBUG.items() if k[0] == t)
                for t in ('fused-bwd', 'fused-rows')}
    finally:
      nn._FUSED_BN_BWD, nn._FUSED_ROW_SCALE, nn._NORM_DEBUG = old
    g = {k: store.grad_views[k].float().cpu().clone() for k in store.trainable_names}
    for i, v in enumerate(xv):
      g[f'dx{i}'] = v.grad.float().cpu()
    res[cfg] = (g, counts)
  assert res['plain'][1] == {'fused-bwd': 0, 'fused-rows': 0}, res['plain'][1]
  # by construction: the bottleneck has biased partial convs in front of norms and norms in front
  # of stride-1 convs; the norms of the upsampling block / decoder entry feed transposed or no convs,
  # the discriminator has instance norms only
  assert res['rows'][1]['fused-rows'] >= (1 if case.name.startswith('bottleneck') else 0), res['rows'][1]
  assert res['bn_bwd'][1]['fused-bwd'] >= (0 if case.name.startswith(('upsampling', 'upc', 'patch')) else 1), \
      res['bn_bwd'][1]
  base = res['plain'][0]
  vec_scale = max([float(v.abs().max()) for k, v in base.items() if v.dim() <= 1] or [1.0])
  for cfg in ('rows', 'bn_bwd'):
    worst = []
    for k, a in res[cfg][0].items():
      b = base[k]
      if cfg == 'rows' and not k.endswith('/bias'):
        assert torch.equal(a, b), (cfg, k)   # same values through a different kernel
        continue
      den = max(float(b.abs().max()), 1e-30)
      if b.dim() <= 1:
        den = max(den, 1e-2 * vec_scale)   # (biases in front of a norm: rounding residue, see above)
      e = float((a - b).abs().max() / den)
      worst.append((e, k))
      assert e <= (8e-3 if k.startswith('dx') else 2e-3), (cfg, k, e)
    print(f'{case.name} [{cfg}]: {res[cfg][1]}; worst', sorted(worst)[-2:])


def _run_block_bf16(case, flags):
  """forward + backward of one block with `nn` module flags overridden; returns gradients + norm path counts"""
  dtype = torch.bfloat16
  gen = torch.Generator().manual_seed(11)
  store = nn.ParamStore()
  mod = case.build(store)
  store.finalize(DEV, torch.Generator().manual_seed(7))
  _randomise(store, gen)
  sg = nn.SpectralGroup(image_models._conv_layers_of(mod), torch.device(DEV))
  xs = [_bf(torch.relu(torch.randn(s, generator=gen) * 0.7 + torch.randn(s[-1], generator=gen) * 0.5))
        for s in case.shapes]
  mask = _mask(*case.mask_shape, gen) if case.mask_shape else None
  old = {k: getattr(nn, k) for k in flags}
  old_dbg = nn._NORM_DEBUG
  for k, v in flags.items():
    setattr(nn, k, v)
  nn._NORM_DEBUG = {}
  try:
    ctx = nn.Ctx(DEV, dtype, training=True, record=True)
    xv = [nn.Var(x.to(DEV).to(dtype)) for x in xs]
    sg.power_iteration(True)
    out, _ = case.hip(ctx, mod, xv, mask.to(DEV) if mask is not None else None)
    if case.name.startswith('head'):
      y_h, push = nn.head(ctx, out, 0)
      push(_bf(torch.randn(y_h.shape, generator=gen)).to(DEV))
    else:
      out.grad = _bf(torch.randn(out.data.shape, generator=gen)).to(DEV).to(dtype)
    y = out.data.float().cpu().clone()
    ctx.backward()
    sg.backward_fixup()
    counts = {}
    for k, v in nn._NORM_DEBUG.items():
      counts[k[0]] = counts.get(k[0], 0) + v
  finally:
    for k, v in old.items():
      setattr(nn, k, v)
    nn._NORM_DEBUG = old_dbg
  g = {k: store.grad_views[k].float().cpu().clone() for k in store.trainable_names}
  for i, v in enumerate(xv):
    g[f'dx{i}'] = v.grad.float().cpu()
  g['y'] = y
  return g, counts


@pytest.mark.parametrize('case', CASES, ids=[c.name for c in CASES])
def test_round5_kernel_choices_match_their_predecessors(case, monkeypatch):
  """Round 5 switched three things on by default; each keeps its predecessor behind a switch, and this
  test owns the switches:
  (a) SE3DS_NORM_CG (nn._NORM_CG + the library's se3ds_norm_bwd_cg_supported): batch-norm backward in
      two launches in the channel-group layout instead of statistics + column reduction + apply.  Same
      sums in a different order: parameter gradients 2e-3 of their class scale, input gradients one
      bf16 ulp of the tensor's maximum (8e-3) -- and the path is really taken where C >= 512;
  (b) SE3DS_CONVT_2X2 (nn._CONVT_2X2): 2x2 stride-2 transposed convs as two pitched 1x1 convolutions
      (se3ds_conv_transpose2x2_fwd) instead of the parity-class data-gradient kernel: BIT-identical;
  (c) SE3DS_MASK_CACHE (nn._MASK_CACHE): partial convs that see the same mask tensor share one
      mask-window launch: BIT-identical;
  (d) SE3DS_DROP_1X1_MASK (nn._DROP_1X1_MASK): 1 x 1 partial convs with a {0, 1} mask leave x * mask
      to the epilogue's `* ratio * update_mask` (both 0 exactly where the mask is 0): equal values
      everywhere (a masked output may change the sign of its zero)."""
  base, cnt = _run_block_bf16(case, {})
  wide = case.name.startswith(('bottleneck', 'trans_basic', 'upsampling'))   # batch norms with C >= 512
  assert cnt.get('cg', 0) >= (1 if wide else 0), cnt
  for name in ('_CONVT_2X2', '_MASK_CACHE', '_DROP_1X1_MASK'):
    got, _ = _run_block_bf16(case, {name: False})
    for k, b in base.items():
      assert torch.equal(got[k], b), (name, k)
  monkeypatch.setenv('SE3DS_NORM_CG', '0')
  got, cnt0 = _run_block_bf16(case, {'_NORM_CG': False})
  assert cnt0.get('cg', 0) == 0, cnt0
  vec_scale = max([float(v.abs().max()) for k, v in base.items() if v.dim() <= 1] or [1.0])
  worst = []
  for k, b in base.items():
    a = got[k]
    if k == 'y':
      assert torch.equal(a, b)   # (the forward pass is not touched)
      continue
    den = max(float(b.abs().max()), 1e-30)
    tol = 8e-3 if k.startswith('dx') else 2e-3
    if b.dim() <= 1 and den < 1e-2 * vec_scale:
      # numerically ZERO (a conv bias straight in front of a batch norm): both forms return the
      # rounding residue of N*H*W bf16-rounded terms summed in their own order; the residues must
      # stay below 2e-4 of the block's per-channel gradient scale
      den, tol = 1e-2 * vec_scale, 2e-2
    e = float((a - b).abs().max() / den)
    worst.append((e, k))
    assert e <= tol, (k, e)
  print(f'{case.name}: cg norms {cnt.get("cg", 0)}; worst vs three-launch backward', sorted(worst)[-2:])
