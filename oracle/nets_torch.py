"""CPU ORACLE (test infrastructure, NOT product code) -- PyTorch-CPU fp32 restatement of the
SE3DS generator / discriminator and GAN step, following the reference statement by statement:
  models/layers.py        PadLayer :22-97, PartialConv :100-209, SpectralConv :275-347,
                          Bottleneck :220-272, ResStack :350-397, TransBasicBlock :400-455,
                          ResStackTranspose :458-511
  models/image_models.py  ResNetGenerator :27-193, ResNetEncoder :196-303, ResNetDecoder
                          :306-488, SNPatchDiscriminator :492-561, SNMultiScaleDiscriminator
                          :564-618
  trainers/se3ds_trainer.py  losses :27-71, train_g_d :129-273, train_d :275-338
  trainers/gan_manager.py :175-183 (Adam), utils/ema.py :54-64
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: the reference's own tests pin PadLayer (three golden arrays), PartialConv ==
conv2d without a mask, masked-pixel invariance, and output shapes/ranges
(tests/test_oracle_nets.py).  Activations, gradients, BN/IN statistics, Adam and EMA values are
NOT pinned by any reference test and TensorFlow cannot be run here: for those this file is
"parity unpinned" -- it follows the published Keras / TF / tfa semantics listed in
SURVEY.md section 8c (glorot init, BN eps 1e-3 momentum 0.99 biased variance, IN eps 1e-3,
LeakyReLU alpha 0.3 default, Adam eps 1e-7, TF SAME padding, clip_by_norm).

Parameters are a flat dict name -> tensor using the product's checkpoint naming, kernels in
Keras layout (Conv2D: kh,kw,cin,cout; Conv2DTranspose: kh,kw,cout,cin).  Activations are NHWC.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM, IN_EPS = 1e-3, 0.99, 1e-3
# binary64 runs only: divide a spectral conv's output (True) or its kernel (False) by sigma; an
# A/B switch for timing the yardstick runs (tools/oracle_f64_time.py)
import os as _os
_F64_DIV_OUTPUT = _os.environ.get('SE3DS_ORACLE_F64_DIV', 'output') == 'output'


# ----------------------------------------------------------------------------- primitives
def _nchw(x):
  return x.permute(0, 3, 1, 2)


def _nhwc(x):
  return x.permute(0, 2, 3, 1)


def pad_layer(x, padding, circular_pad=False, training=None, mode='CONSTANT', constant_value=0):
  """layers.py:62-97 on NHWC."""
  p = padding
  if p == 0:
    return x
  n, h, w, c = x.shape
  if mode == 'CONSTANT' and constant_value == 0:
    if circular_pad and not training:
      left, right = x[:, :, -p:, :], x[:, :, :p, :]
    else:
      left = right = torch.zeros((n, h, p, c), dtype=x.dtype)
    t = torch.cat([left, x, right], dim=2)
    z = torch.zeros((n, p, w + 2 * p, c), dtype=x.dtype)
    return torch.cat([z, t, z], dim=1)
  tmode = {'CONSTANT': 'constant', 'REFLECT': 'reflect', 'SYMMETRIC': 'symmetric'}[mode.upper()]
  def tfpad(t, ph, pw):
    t = _nchw(t)
    if tmode == 'constant':
      t = F.pad(t, (pw, pw, ph, ph), mode='constant', value=constant_value)
    elif tmode == 'reflect':
      t = F.pad(t, (pw, pw, ph, ph), mode='reflect')
    else:  # symmetric = reflect including the edge
      idx_h = torch.cat([torch.arange(ph - 1, -1, -1), torch.arange(h),
                         torch.arange(h - 1, h - 1 - ph, -1)]) if ph else torch.arange(h)
      wcur = t.shape[3]
      idx_w = torch.cat([torch.arange(pw - 1, -1, -1), torch.arange(wcur),
                         torch.arange(wcur - 1, wcur - 1 - pw, -1)]) if pw else torch.arange(wcur)
      t = t[:, :, idx_h][:, :, :, idx_w]
    return _nhwc(t)
  if circular_pad and not training:
    t = tfpad(x, p, 0)
    left, right = t[:, :, -p:, :], t[:, :, :p, :]
    return torch.cat([left, t, right], dim=2)
  return tfpad(x, p, p)


def _same_pads(size, k, s):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  return total // 2, total - total // 2


def _conv2d_patches(x, kernel, stride):
  """'VALID' convolution of an (already padded) NHWC tensor as ONE matrix product in the operands'
  natural layouts: patches[n, y, x, (ky, kx, ci)] @ kernel.reshape(kh*kw*ci, co).  Used for the
  binary64 yardstick runs only: PyTorch's fp64 conv2d is a per-image im2col loop behind a strided
  transposition of every kernel (HWIO -> OIHW), which at 1.1 B parameters was most of those runs'
  time (aten::copy_ 30 %, slow_conv2d 14 %); the products and sums are the same up to
  reassociation (1e-16 relative in binary64)."""
  kh, kw, ci, co = kernel.shape
  n, hp, wp, _ = x.shape
  ho, wo = (hp - kh) // stride + 1, (wp - kw) // stride + 1
  cols = [x[:, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride, :]
          for ky in range(kh) for kx in range(kw)]
  patches = torch.cat(cols, dim=-1) if len(cols) > 1 else cols[0]
  y = patches.reshape(-1, kh * kw * ci) @ kernel.reshape(kh * kw * ci, co)
  return y.reshape(n, ho, wo, co)


def tf_conv2d(x, kernel, stride, padding):
  """tf.nn.conv2d: NHWC input, HWIO kernel, 'VALID' | 'SAME'."""
  k = kernel.shape[0]
  if x.dtype == torch.float64:
    if padding.upper() == 'SAME':
      pt, pb = _same_pads(x.shape[1], k, stride)
      pl, pr = _same_pads(x.shape[2], kernel.shape[1], stride)
      x = F.pad(x, (0, 0, pl, pr, pt, pb))
    return _conv2d_patches(x, kernel, stride)
  xn = _nchw(x)
  if padding.upper() == 'SAME':
    pt, pb = _same_pads(x.shape[1], k, stride)
    pl, pr = _same_pads(x.shape[2], kernel.shape[1], stride)
    xn = F.pad(xn, (pl, pr, pt, pb))
  w = kernel.permute(3, 2, 0, 1).contiguous()
  return _nhwc(F.conv2d(xn.contiguous(), w, stride=stride))


def keras_conv2d_transpose(x, kernel, bias, stride):
  """Keras Conv2DTranspose (kernel kh,kw,cout,cin), stride 2, output 2H x 2W: k3 'SAME' with
  output_padding=1 or k2 ('VALID' / 'SAME').  Gradient of the SAME/VALID forward conv whose
  top/left padding is 0 for these shapes."""
  assert stride == 2
  n, h, w, _ = x.shape
  wt = kernel.permute(3, 2, 0, 1).contiguous()  # (cin, cout, kh, kw)
  y = F.conv_transpose2d(_nchw(x).contiguous(), wt, stride=stride)
  y = _nhwc(y)[:, :2 * h, :2 * w, :]
  if bias is not None:
    y = y + bias
  return y


def power_iteration(kernel, u, eps=1e-10):
  """layers.py:176-186 / 318-327: returns sigma (differentiable in W), u_hat."""
  w = kernel.reshape(-1, kernel.shape[-1])
  v = u @ w.t()
  v_hat = v / (torch.norm(v) + eps)
  un = v_hat @ w
  u_hat = un / (torch.norm(un) + eps)
  u_hat, v_hat = u_hat.detach(), v_hat.detach()
  sigma = (v_hat @ w) @ u_hat.t()
  return sigma, u_hat


def leaky_relu(x, alpha):
  return torch.where(x > 0, x, x * alpha)


class CrossReplicaSum(torch.autograd.Function):
  """tf.distribute ReplicaContext.all_reduce(SUM, .) as Keras SyncBatchNormalization uses it for
  its batch statistics: differentiable, and the gradient of an all-reduce is the all-reduce of the
  upstream gradients -- a replica's activations also receive the sensitivity of the OTHER
  replicas' losses to the shared statistics.  Needs an initialised torch.distributed group; every
  replica must run the same graph (forward and backward collectives then pair up in order)."""

  @staticmethod
  def forward(ctx, t):
    import torch.distributed as dist
    out = t.detach().clone()
    dist.all_reduce(out)
    return out

  @staticmethod
  def backward(ctx, g):
    import torch.distributed as dist
    out = g.detach().clone().contiguous()
    dist.all_reduce(out)
    return out


def pooled_stats_hook(world):
  """stats_hook for Net.sync_bn: (sum, sum of squares, count) over all replicas."""
  def pooled(name, s1, s2, cnt):
    t = CrossReplicaSum.apply(torch.stack([s1, s2]))
    return t[0], t[1], cnt * world
  return pooled


class Net:
  """Parameter access + side-effect bookkeeping (BN moving stats, spectral u)."""

  def __init__(self, params, training, stats_hook=None, bn_training=None):
    self.p = params
    self.training = training
    # test hook: decouple BN statistics mode from the padding / spectral-u training flag
    self.bn_training = training if bn_training is None else bn_training
    self.updates = {}          # name -> new value of non-trainable variables
    self.stats_hook = stats_hook
    # test hook: tag -> bool tensor of "output > 0" decisions to use INSTEAD of the sign of the
    # oracle's own pre-activation (block-level parity tests run the oracle's backward under the
    # decisions the device path took; see tests/test_blocks_gpu.py).  Tags are the names of the
    # norm / conv layer whose output is activated.  acts: tag -> pre-activation (recorded).
    self.decisions = None
    self.pre_acts = None
    # test hook (tools/bf16_attribution.py): quant(x, site) -> tensor, called where the bf16 device
    # path STORES a tensor -- 'weight' (the operand copy of a kernel), 'conv' (a convolution's output),
    # 'act' (normalised + activated), 'norm' (normalised, not activated, stored on its own), 'res' (the
    # residual stream: a block's output), 'add' (a skip connection's sum), 'resgrad' (identity on the
    # block input as the residual operand: its GRADIENT is a stored tensor).  None: no-op.
    self.quant = None

  def q(self, x, site):
    return x if self.quant is None else self.quant(x, site)

  def get(self, name):
    return self.p[name]

  def act(self, x, tag, alpha=0.0, site='act'):
    """ReLU (alpha 0) / LeakyReLU(alpha) of the tensor produced by layer `tag`."""
    if self.pre_acts is not None:
      self.pre_acts[tag] = x.detach()
    if self.decisions is not None and tag in self.decisions:
      return self.q(torch.where(self.decisions[tag], x, x * alpha), site)
    return self.q(torch.where(x > 0, x, x * alpha) if alpha else F.relu(x), site)

  def has(self, name):
    return name in self.p

  # -- layers -----------------------------------------------------------------------------
  def sync_bn(self, x, name):
    g, b = self.get(name + '/gamma'), self.get(name + '/beta')
    if self.bn_training:
      cnt = x.shape[0] * x.shape[1] * x.shape[2]
      s1 = x.sum(dim=(0, 1, 2))
      s2 = (x * x).sum(dim=(0, 1, 2))
      if self.stats_hook is not None:
        s1, s2, cnt = self.stats_hook(name, s1, s2, cnt)
      mean = s1 / cnt
      var = s2 / cnt - mean * mean
      mm, mv = self.get(name + '/moving_mean'), self.get(name + '/moving_variance')
      self.updates[name + '/moving_mean'] = (mm - (mm - mean.detach()) * (1 - BN_MOMENTUM))
      self.updates[name + '/moving_variance'] = (mv - (mv - var.detach()) * (1 - BN_MOMENTUM))
    else:
      mean, var = self.get(name + '/moving_mean'), self.get(name + '/moving_variance')
    inv = torch.rsqrt(var + BN_EPS) * g
    return x * inv + (b - mean * inv)

  def instance_norm(self, x, name):
    g, b = self.get(name + '/gamma'), self.get(name + '/beta')
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = torch.rsqrt(var + IN_EPS) * g
    return x * inv + (b - mean * inv)

  def conv2d(self, x, name, stride=1, padding='VALID'):
    y = tf_conv2d(x, self.q(self.get(name + '/kernel'), 'weight'), stride, padding)
    if self.has(name + '/bias'):
      y = y + self.get(name + '/bias')
    return self.q(y, 'conv')

  def spectral_conv(self, x, name, stride=1, padding='VALID'):
    """layers.py:299-347."""
    kernel = self.get(name + '/kernel')
    sigma, u_hat = power_iteration(kernel, self.get(name + '/u'))
    if self.training:
      self.updates[name + '/u'] = u_hat
    if self.quant is not None:
      # (the device path rounds the RAW kernel to its operand dtype and applies 1 / sigma in the epilogue)
      y = tf_conv2d(x, self.q(kernel, 'weight'), stride, padding) / (sigma + 1e-10)
    elif kernel.dtype == torch.float64 and _F64_DIV_OUTPUT:
      # binary64 yardstick runs: conv(x, W / s) = conv(x, W) / s -- the division runs over the
      # activation instead of materialising (and keeping for the backward pass) a normalised copy
      # of every kernel, 9 GB at the real dimensions; same value up to binary64 rounding
      y = tf_conv2d(x, kernel, stride, padding) / (sigma + 1e-10)
    else:
      w_norm = kernel / (sigma + 1e-10)
      y = tf_conv2d(x, w_norm, stride, padding)
    if self.has(name + '/bias'):
      y = y + self.get(name + '/bias')
    return self.q(y, 'conv')

  def conv_fn(self, spectral):
    return self.spectral_conv if spectral else self.conv2d

  def partial_conv(self, x, mask, name, stride=1, padding='VALID', spectral=False):
    """layers.py:132-209."""
    kernel = self.get(name + '/kernel')
    k = kernel.shape[0]
    if mask is None:
      mask = torch.ones((x.shape[0], x.shape[1], x.shape[2], 1), dtype=x.dtype)
    update_mask = tf_conv2d(mask, torch.ones((k, k, 1, 1), dtype=x.dtype), stride, padding)
    mask_ratio = (k * k) / (update_mask + 1e-6)
    update_mask = torch.clamp(update_mask, 0, 1)
    mask_ratio = mask_ratio * update_mask
    mask, update_mask, mask_ratio = mask.detach(), update_mask.detach(), mask_ratio.detach()
    out = x * mask
    if spectral:
      _, u_hat = power_iteration(kernel, self.get(name + '/u'))
      if self.training:
        self.updates[name + '/u'] = u_hat
      # w_norm is computed by the reference but NOT used (layers.py:189-195)
    out = tf_conv2d(out, self.q(kernel, 'weight'), stride, padding)
    if self.has(name + '/bias'):
      bias = self.get(name + '/bias').reshape(1, 1, 1, -1)
      out = (out - bias) * mask_ratio + bias
      out = out * update_mask
    else:
      out = out * mask_ratio
    return self.q(out, 'conv'), update_mask

  def conv_transpose(self, x, name, stride=2):
    return self.q(keras_conv2d_transpose(x, self.q(self.get(name + '/kernel'), 'weight'),
                                         self.get(name + '/bias') if self.has(name + '/bias') else None,
                                         stride), 'conv')

  def pad(self, x, p, circular=True):
    return pad_layer(x, p, circular_pad=circular, training=self.training)

  # -- blocks -----------------------------------------------------------------------------
  def bottleneck(self, x, mask, name, stride, has_ds, ds_name, spectral, circular):
    """layers.py:253-272."""
    residual = self.q(x, 'resgrad')
    out, um = self.partial_conv(x, mask, name + '/conv1', 1, 'SAME', spectral)
    out = self.act(self.sync_bn(out, name + '/bn1'), name + '/bn1')
    out = self.pad(out, 1, circular)
    um = self.pad(um, 1, circular)
    out, um = self.partial_conv(out, um, name + '/conv2', stride, 'VALID', spectral)
    out = self.act(self.sync_bn(out, name + '/bn2'), name + '/bn2')
    out, um = self.partial_conv(out, um, name + '/conv3', 1, 'SAME', spectral)
    out = self.sync_bn(out, name + '/bn3')
    if has_ds:
      residual, _ = self.partial_conv(x, mask, ds_name, stride, 'SAME', spectral)
      residual = self.q(self.sync_bn(residual, name + '/ds_norm'), 'norm')
    return self.act(out + residual, name + '/bn3', site='res'), um

  def res_stack(self, x, mask, name, inplanes, planes, blocks, stride, spectral, circular,
                expansion=4):
    """layers.py:350-397."""
    has_ds = stride != 1 or inplanes != planes * expansion
    out, um = self.bottleneck(x, mask, name + '/block0', stride, has_ds, name + '/downsample',
                              spectral, circular)
    for i in range(1, blocks):
      out, um = self.bottleneck(out, um, name + f'/block{i}', 1, False, None, spectral, circular)
    return out, um

  def trans_basic_block(self, x, name, stride, up_kind, up_name, spectral, circular):
    """layers.py:400-455.  up_kind: None | 'convT' | 'conv1x1'."""
    conv = self.conv_fn(spectral)
    out = conv(self.pad(x, 1, circular), name + '/conv_a', 1, 'VALID')
    out = self.act(self.sync_bn(out, name + '/bn_a'), name + '/bn_a')
    if up_kind is not None and stride != 1:
      out = self.conv_transpose(out, name + '/conv_b', stride)
    else:
      out = conv(self.pad(out, 1, circular), name + '/conv_b', stride, 'VALID')
    out = self.sync_bn(out, name + '/bn_b')
    residual = self.q(x, 'resgrad')
    if up_kind == 'convT':
      residual = self.q(self.sync_bn(self.conv_transpose(x, up_name + '/conv', stride), up_name + '/bn'), 'norm')
    elif up_kind == 'conv1x1':
      residual = self.q(self.sync_bn(conv(x, up_name + '/conv', 1, 'VALID'), up_name + '/bn'), 'norm')
    return self.act(out + residual, name + '/bn_b', site='res')

  def res_stack_transpose(self, x, name, inplanes, planes, blocks, stride, spectral, circular):
    """layers.py:458-511."""
    up_kind = 'convT' if stride != 1 else ('conv1x1' if inplanes != planes else None)
    for i in range(blocks - 1):
      x = self.trans_basic_block(x, name + f'/block{i}', 1, None, None, spectral, circular)
    return self.trans_basic_block(x, name + f'/block{blocks - 1}', stride, up_kind,
                                  name + '/upsample', spectral, circular)


ENC_BLOCKS = {'50': [3, 4, 6, 3], '101': [3, 4, 23, 3], '152': [3, 8, 36, 3]}
DEC_BLOCKS = {'50': [6, 4, 3, 3], '101': [23, 4, 3, 3], '152': [36, 8, 3, 3]}


def max_pool_same(x):
  return _nhwc(F.max_pool2d(_nchw(x), 2, 2, ceil_mode=True))


def avg_pool3s2_same(x):
  """tf.nn.avg_pool(ksize=3, strides=2, 'SAME'): divisor = number of in-bounds taps."""
  pt, pb = _same_pads(x.shape[1], 3, 2)
  pl, pr = _same_pads(x.shape[2], 3, 2)
  xn = F.pad(_nchw(x), (pl, pr, pt, pb))
  ones = F.pad(torch.ones((1, 1, x.shape[1], x.shape[2]), dtype=x.dtype), (pl, pr, pt, pb))
  s = F.avg_pool2d(xn, 3, 2, divisor_override=1)
  c = F.avg_pool2d(ones, 3, 2, divisor_override=1)
  return _nhwc(s / c)


def encoder(net, x, mask, d, version, spectral, name='encoder'):
  """image_models.py:276-303."""
  um = net.pad(mask, 3)
  out = net.pad(x, 3)
  out, um = net.partial_conv(out, um, name + '/conv1', 2, 'VALID')
  out = net.act(net.sync_bn(out, name + '/bn1'), name + '/bn1')
  b1 = out
  out, um = max_pool_same(out), max_pool_same(um)
  blocks = ENC_BLOCKS[version]
  out, um = net.res_stack(out, um, name + '/stack1', d, d, blocks[0], 1, spectral, True)
  s1 = out
  out, um = net.res_stack(out, um, name + '/stack2', d * 4, d * 2, blocks[1], 2, spectral, True)
  s2 = out
  out, um = net.res_stack(out, um, name + '/stack3', d * 8, d * 4, blocks[2], 2, spectral, True)
  s3 = out
  out, um = net.res_stack(out, um, name + '/stack4', d * 16, d * 8, blocks[3], 2, spectral, True)
  out = net.pad(out, 1)
  um = net.pad(um, 1)
  out, um = net.partial_conv(out, um, name + '/final_conv', 1, 'VALID')
  out = net.act(net.sync_bn(out, name + '/final_bn'), name + '/final_bn')
  return out, [b1, s1, s2, s3]


def decoder(net, x, skip, d, version, spectral, name):
  """image_models.py:443-488 (partial_conv=True, masks all None)."""
  conv = net.conv_fn(spectral)
  out = conv(x, name + '/upc/conv', 1, 'SAME')
  out = net.act(net.sync_bn(out, name + '/upc/bn'), name + '/upc/bn', 0.2)
  out = out.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)   # UpSampling2D nearest
  def agent(t, nm):
    y, _ = net.partial_conv(t, None, name + '/' + nm, 1, 'SAME', spectral)
    return net.act(net.sync_bn(y, name + '/' + nm + '_bn'), name + '/' + nm + '_bn')
  blocks = DEC_BLOCKS[version]
  out = agent(out, 'agent4')
  out = net.res_stack_transpose(out, name + '/deconv1', d * 8, d * 4, blocks[0], 1, spectral, True)
  out = net.q(out + agent(skip[3], 'agent3'), 'add')
  out = net.res_stack_transpose(out, name + '/deconv2', d * 4, d * 2, blocks[1], 2, spectral, True)
  out = net.q(out + agent(skip[2], 'agent2'), 'add')
  out = net.res_stack_transpose(out, name + '/deconv3', d * 2, d, blocks[2], 2, spectral, True)
  out = net.q(out + agent(skip[1], 'agent1'), 'add')
  out = net.res_stack_transpose(out, name + '/deconv4', d, d, blocks[3], 2, spectral, True)
  out = net.q(out + agent(skip[0], 'agent0'), 'add')
  out = net.res_stack_transpose(out, name + '/final_conv', d, d, 3, 1, False, True)
  return net.conv_transpose(out, name + '/final_deconv', 2)


def head(net, x, name, spectral):
  """image_models.py:79-104."""
  conv = net.conv_fn(spectral)
  for i in range(3):
    x = net.q(net.sync_bn(x, name + f'/bn{i}'), 'norm')
    x = conv(net.pad(x, 1), name + f'/conv{i}', 1, 'VALID')
    if i < 2:
      x = net.act(x, name + f'/conv{i}', 0.3)
  return x


def generator_forward(params, cond, training, gen_dims, resnet_version='50', context_layer='convs',
                      conv_mode='spectral', use_blurred_mask=True, z_dim=128, stats_hook=None,
                      taps=None, bn_training=None, quant=None):
  """image_models.py:132-193.  Returns (outputs list, net.updates)."""
  net = Net(params, training, stats_hook, bn_training)
  net.quant = quant
  spectral = conv_mode == 'spectral'
  d = gen_dims
  parts = [cond['proj_image'], cond['proj_depth']]
  if use_blurred_mask:
    parts.append(cond['blurred_mask'])
  x = torch.cat(parts, dim=-1)
  hidden, skip = encoder(net, x, cond['proj_mask'], d, resnet_version, spectral)
  if taps is not None:
    taps.update(b1=skip[0], s1=skip[1], s2=skip[2], s3=skip[3], enc=hidden)
  if context_layer == 'convs':
    for i in range(4):
      hidden = net.q(net.sync_bn(hidden, f'context/bn{i}'), 'norm')
      hidden = net.spectral_conv(net.pad(hidden, 1), f'context/conv{i}', 1, 'VALID')
      if i < 3:
        hidden = net.act(hidden, f'context/conv{i}', 0.3)
  n, hh, hw, _ = hidden.shape
  out = decoder(net, hidden, skip, d, resnet_version, spectral, 'decoder')
  depth_out = decoder(net, hidden, skip, d, resnet_version, spectral, 'depth_decoder')
  if taps is not None:
    taps.update(ctx=hidden, dec=out, ddec=depth_out)
  rgb = head(net, out, 'rgb_conv', spectral)
  depth = head(net, depth_out, 'depth_conv', spectral)
  rgb = (torch.tanh(rgb) + 1) / 2
  depth = torch.clamp(depth, 0, 1)
  z = torch.zeros((n, hh, hw, z_dim))
  seg = torch.zeros(tuple(cond['proj_depth'].shape[:-1]) + (42,))
  return [z, z.clone(), z.clone(), depth, seg, seg.clone(), rgb], net.updates


def patch_discriminator(net, x, name, n_layers, kernel_size=4):
  """image_models.py:492-561."""
  k = kernel_size
  results = []
  out = net.conv2d(pad_layer(x, k // 2), name + '/g0/conv', 2, 'VALID')
  out = net.act(out, name + '/g0/conv', 0.2)   # (LeakyReLU 0.2; tags: see Net.decisions)
  results.append(out)
  for i in range(1, n_layers):
    out = net.spectral_conv(pad_layer(out, k // 2), name + f'/g{i}/conv',
                            2 if i != n_layers - 1 else 1, 'VALID')
    out = net.act(net.instance_norm(out, name + f'/g{i}/in'), name + f'/g{i}/in', 0.2)
    results.append(out)
  out = net.conv2d(out, name + '/final', 1, 'SAME')
  results.append(out)
  return results


def discriminator_forward(params, x, training, n_dis=2, n_layers=5, kernel_size=4):
  """image_models.py:599-618."""
  net = Net(params, training)
  result = []
  prev = x
  for i in range(n_dis):
    result.append(patch_discriminator(net, prev, f'dis{i}', n_layers, kernel_size))
    prev = avg_pool3s2_same(prev)
  return result, net.updates


# --------------------------------------------------------------------------------- trainer
def clip_by_norm(g, clip=5.0):
  """tf.clip_by_norm."""
  l2sum = (g * g).sum()
  norm = torch.sqrt(l2sum) if l2sum > 0 else l2sum
  return (g * clip) / torch.maximum(norm, torch.tensor(clip))


def adam_keras(p, g, m, v, lr, b1, b2, step, eps=1e-7):
  """Keras Adam (optimizer_v2/adam.py -> ResourceApplyAdam, training_ops.cc): hyper-parameters
  are tensors of the VARIABLE's dtype, so 1 - beta and beta^t are taken in that dtype
  (1 - 0.999f = 0.00100004673, not 0.001); alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)."""
  t = lambda x: torch.tensor(x, dtype=p.dtype)
  b1t, b2t, lrt = t(b1), t(b2), t(lr)
  alpha = lrt * torch.sqrt(1 - torch.pow(b2t, step)) / (1 - torch.pow(b1t, step))
  m = m + (g - m) * (1 - b1t)
  v = v + (g * g - v) * (1 - b2t)
  return p - (m * alpha) / (torch.sqrt(v) + eps), m, v


def ema_update(ema_var, value, ema_decay):
  """utils/ema.py:54-64: ema_var.assign_sub((ema_var - value) * (1.0 - ema_decay)); the Python
  float `1.0 - ema_decay` is converted to the variable's dtype by the multiplication."""
  one_minus_decay = 1.0 - ema_decay
  return ema_var - (ema_var - value) * one_minus_decay


def ema_step(ema_vars, new_values, global_step, ema_decay, ema_init_step, num_batched_steps):
  """gan_manager.py:642-655 over dicts of ALL generator variables (trainable, BN moving
  statistics, spectral u): hard copy while global_step < ema_init_step + num_batched_steps
  (global_step only advances on the host once per cluster, :421), moving average afterwards;
  untouched before ema_init_step."""
  if global_step < ema_init_step:
    return dict(ema_vars)
  if global_step >= ema_init_step + num_batched_steps:
    return {k: ema_update(ema_vars[k], new_values[k], ema_decay) for k in ema_vars}
  return {k: new_values[k].clone() for k in ema_vars}


def wc_loss(gen, real, mask):
  """se3ds_trainer.py:39-55 (returns the (N,) vector)."""
  l = torch.abs(gen - real)
  l = (l * mask).sum(dim=(1, 2, 3)) / gen.shape[-1]
  return l / torch.clamp(mask.sum(dim=(1, 2, 3)), min=1)


def d_losses(logits):
  fake_real = [(sub[-1][:sub[-1].shape[0] // 2], sub[-1][sub[-1].shape[0] // 2:]) for sub in logits]
  gen = sum((-f).mean() for f, _ in fake_real) / len(fake_real)
  disc = sum((F.relu(1.0 - r) + F.relu(1.0 + f)).mean() for f, r in fake_real) / len(fake_real)
  return gen, disc


def train_g_d(g_params, d_params, inputs, cfg, replicas=1):
  """se3ds_trainer.py:129-257 up to (and excluding) the optimizer: returns clipped gradient
  dicts for G and D, the variable updates (BN moving stats, u) and the metric values."""
  trainable_g = {k: v.clone().requires_grad_(True) for k, v in g_params.items() if cfg['g_train'](k)}
  trainable_d = {k: v.clone().requires_grad_(True) for k, v in d_params.items() if cfg['d_train'](k)}
  gp = dict(g_params); gp.update(trainable_g)
  dp = dict(d_params); dp.update(trainable_d)
  inputs = dict(inputs)
  if not cfg.get('mask_blurred', False):
    inputs['blurred_mask'] = torch.zeros_like(inputs['blurred_mask'])
  blurred = inputs['blurred_mask']
  depth_t = inputs['depth']
  tmask = ((depth_t > 0) & (depth_t < 1)).float()
  npx = torch.clamp(tmask.sum(dim=(1, 2, 3)), min=1)
  outs, g_updates = generator_forward(gp, inputs, True, **cfg['gen'])
  depth_out, generated = outs[3], outs[6]
  depth_loss = (torch.abs(depth_out - depth_t) * tmask).sum(dim=(1, 2, 3)) / npx
  depth_loss = cfg['lambda_depth'] * depth_loss.mean()
  kld = cfg['lambda_kld'] * outs[2].mean()
  wc = cfg['lambda_wc'] * wc_loss(generated, inputs['proj_image'], inputs['proj_mask'] * (1 - blurred))
  fake = torch.cat([generated, depth_out], dim=-1)
  real = torch.cat([inputs['image'], depth_t], dim=-1)
  logits, d_updates = discriminator_forward(dp, torch.cat([fake, real], dim=0), True, **cfg['dis'])
  gen_loss, disc_loss = d_losses(logits)
  gen_loss, disc_loss = cfg['lambda_gan'] * gen_loss, cfg['lambda_gan'] * disc_loss
  combined = gen_loss + kld + wc + depth_loss          # (N,) vector, as in the reference
  g_names, d_names = list(trainable_g), list(trainable_d)
  # gen_tape.gradient of a vector target sums its elements (se3ds_trainer.py:231-237)
  g_grads = torch.autograd.grad((combined / replicas).sum(), [trainable_g[k] for k in g_names],
                                retain_graph=True, allow_unused=True)
  d_grads = torch.autograd.grad(disc_loss / replicas, [trainable_d[k] for k in d_names],
                                allow_unused=True)
  raw_g = {k: g for k, g in zip(g_names, g_grads)}
  raw_d = {k: g for k, g in zip(d_names, d_grads)}
  cg = {k: clip_by_norm(g) for k, g in raw_g.items() if g is not None}
  cd = {k: clip_by_norm(g) for k, g in raw_d.items() if g is not None}
  metrics = {
      'gen/gen_gan_loss': float(gen_loss), 'dis/disc_loss': float(disc_loss),
      'gen/depth_loss': float(depth_loss), 'gen/wc_loss': float(wc.mean()),
      'gen/gen_loss': float(combined.mean()), 'gen/kld_loss': float(kld),
      'gen/grad_norm': float(torch.stack([g.norm() for g in cg.values()]).mean()),
      'dis/grad_norm': float(torch.stack([g.norm() for g in cd.values()]).mean()),
  }
  return dict(g_grads=cg, d_grads=cd, raw_g=raw_g, raw_d=raw_d, g_updates=g_updates,
              d_updates=d_updates, metrics=metrics, generated=generated.detach(),
              depth_out=depth_out.detach(),
              logits=[[t.detach() for t in sub] for sub in logits])


def train_d(g_params, d_params, inputs, cfg, replicas=1):
  """se3ds_trainer.py:275-338 (gradients only)."""
  trainable_d = {k: v.clone().requires_grad_(True) for k, v in d_params.items() if cfg['d_train'](k)}
  dp = dict(d_params); dp.update(trainable_d)
  inputs = dict(inputs)
  if not cfg.get('mask_blurred', False):
    inputs['blurred_mask'] = torch.zeros_like(inputs['blurred_mask'])
  with torch.no_grad():
    outs, g_updates = generator_forward(g_params, inputs, True, **cfg['gen'])
  fake = torch.cat([outs[6], outs[3]], dim=-1)
  real = torch.cat([inputs['image'], inputs['depth']], dim=-1)
  logits, d_updates = discriminator_forward(dp, torch.cat([fake, real], dim=0), True, **cfg['dis'])
  _, disc_loss = d_losses(logits)
  disc_loss = cfg['lambda_gan'] * disc_loss / replicas
  names = list(trainable_d)
  grads = torch.autograd.grad(disc_loss, [trainable_d[k] for k in names], allow_unused=True)
  cd = {k: clip_by_norm(g) for k, g in zip(names, grads) if g is not None}
  return dict(d_grads=cd, g_updates=g_updates, d_updates=d_updates)
