"""Common layers used for modeling -- MI355X implementation of the reference's
models/layers.py.  Same class names and call structure; differences forced by leaving Keras:
input channel counts are explicit constructor arguments, parameters live in a ParamStore,
and every call takes the execution context `ctx` first (compute dtype, training flag, tape).
Padding layers are folded into the consuming convolution's gather (no padded copy is ever
materialised); `PadLayer.__call__` still exists as a standalone op for API parity."""
from typing import Optional, Tuple

import torch

from se3ds_amd.hipops import nn
from se3ds_amd.hipops.nn import ACT_LRELU, ACT_NONE, ACT_RELU, ConvLayer, Ctx, NormLayer, Var


class PadLayer:
  """Circular (W, inference only) / zero / tf.pad-mode padding (reference :22-97)."""

  def __init__(self, padding: int, circular_pad: bool = False, mode='CONSTANT',
               constant_value=0):
    self.padding = padding
    self.circular_pad = circular_pad
    self.mode = mode
    self.constant_value = constant_value

  def wraps(self, training) -> bool:
    # reference :70,83: circular only when `circular_pad and not training`
    return bool(self.circular_pad and not training)

  def __call__(self, inputs: torch.Tensor, training=None) -> torch.Tensor:
    return nn.pad2d(inputs, self.padding, self.wraps(training), self.mode, self.constant_value)


class PartialConv:
  """Partial 2D convolution (reference :100-209): returns (out, update_mask)."""
  spectral_norm = False

  def __init__(self, store, name, in_channels, filters, kernel_size, strides=1, padding='VALID',
               use_bias=True):
    kind = 'partial_spectral' if self.spectral_norm else 'partial'
    self.conv = ConvLayer(store, name, in_channels, filters, kernel_size, strides, padding,
                          use_bias, kind)

  def __call__(self, ctx: Ctx, feature: Var, mask: Optional[torch.Tensor] = None,
               pad: Optional[PadLayer] = None) -> Tuple[Var, torch.Tensor]:
    p = pad.padding if pad is not None else 0
    wrap = pad.wraps(ctx.training) if pad is not None else False
    return nn.conv2d(ctx, feature, self.conv, pad=p, wrap=wrap, mask=mask)


class PartialSpectralConv(PartialConv):
  """Spectral-normalised partial conv (reference :212-217).  As in the reference the
  normalised weight is computed (and `u` advanced when training) but the convolution uses
  the raw kernel (:189-195)."""
  spectral_norm = True


class SpectralConv:
  """Convolution with spectral normalisation applied to the weights (reference :275-347)."""

  def __init__(self, store, name, in_channels, filters, kernel_size, strides=1, padding='VALID',
               use_bias=True):
    self.conv = ConvLayer(store, name, in_channels, filters, kernel_size, strides, padding,
                          use_bias, 'spectral')

  def __call__(self, ctx: Ctx, feature: Var, pad: Optional[PadLayer] = None, act=ACT_NONE,
               alpha=0.0) -> Var:
    p = pad.padding if pad is not None else 0
    wrap = pad.wraps(ctx.training) if pad is not None else False
    return nn.conv2d(ctx, feature, self.conv, pad=p, wrap=wrap, act=act, alpha=alpha)


class Conv2D:
  """tf.keras.layers.Conv2D."""

  def __init__(self, store, name, in_channels, filters, kernel_size, strides=1, padding='VALID',
               use_bias=True):
    self.conv = ConvLayer(store, name, in_channels, filters, kernel_size, strides, padding,
                          use_bias, 'plain')

  def __call__(self, ctx: Ctx, feature: Var, pad: Optional[PadLayer] = None, act=ACT_NONE,
               alpha=0.0) -> Var:
    p = pad.padding if pad is not None else 0
    wrap = pad.wraps(ctx.training) if pad is not None else False
    return nn.conv2d(ctx, feature, self.conv, pad=p, wrap=wrap, act=act, alpha=alpha)


class Conv2DTranspose:
  """tf.keras.layers.Conv2DTranspose, stride 2 (k3 SAME output_padding=1 / k2)."""

  def __init__(self, store, name, in_channels, filters, kernel_size, strides=2, use_bias=True):
    self.conv = ConvLayer(store, name, in_channels, filters, kernel_size, strides, 'SAME',
                          use_bias, 'plain', transpose=True)

  def __call__(self, ctx: Ctx, x: Var) -> Var:
    return nn.conv_transpose2d(ctx, x, self.conv)


class SyncBatchNormalization:
  def __init__(self, store, name, channels):
    self.norm = NormLayer(store, name, channels, 'batch')

  def __call__(self, ctx: Ctx, x: Var, act=ACT_NONE, alpha=0.0, res: Var = None,
               in_act=None) -> Var:
    return nn.norm_act(ctx, x, self.norm, act=act, alpha=alpha, res=res, in_act=in_act)


class InstanceNormalization:
  def __init__(self, store, name, channels):
    self.norm = NormLayer(store, name, channels, 'instance')

  def __call__(self, ctx: Ctx, x: Var, act=ACT_NONE, alpha=0.0) -> Var:
    return nn.norm_act(ctx, x, self.norm, act=act, alpha=alpha)


class Bottleneck:
  """ResNet bottleneck block (reference :220-272)."""

  def __init__(self, store, name, in_channels, filters=128, strides=1, expansion=4,
               downsample=None, circular_pad=False, partial_fn=PartialConv):
    self.conv1 = partial_fn(store, name + '/conv1', in_channels, filters, 1, 1, 'SAME')
    self.bn1 = SyncBatchNormalization(store, name + '/bn1', filters)
    self.pad1 = PadLayer(1, circular_pad=circular_pad)
    self.conv2 = partial_fn(store, name + '/conv2', filters, filters, 3, strides, 'VALID')
    self.bn2 = SyncBatchNormalization(store, name + '/bn2', filters)
    self.conv3 = partial_fn(store, name + '/conv3', filters, expansion * filters, 1, 1, 'SAME')
    self.bn3 = SyncBatchNormalization(store, name + '/bn3', expansion * filters)
    self.downsample = downsample
    if downsample is not None:
      self.ds_norm = SyncBatchNormalization(store, name + '/ds_norm', expansion * filters)

  def __call__(self, ctx: Ctx, x: Var, mask=None):
    out, update_mask = self.conv1(ctx, x, mask)
    out = self.bn1(ctx, out, act=ACT_RELU)
    # pad1 is applied to the activations and to the mask (reference :260-261); both folded
    out, update_mask = self.conv2(ctx, out, update_mask, pad=self.pad1)
    out = self.bn2(ctx, out, act=ACT_RELU)
    out, update_mask = self.conv3(ctx, out, update_mask)
    residual = x
    if self.downsample is not None:
      residual, _ = self.downsample(ctx, x, mask)   # the block-INPUT mask (reference :266)
      residual = self.ds_norm(ctx, residual)
    out = self.bn3(ctx, out, act=ACT_RELU, res=residual)   # relu(bn3(out) + residual)
    return out, update_mask


class ResStack:
  """Single ResNet stack of Bottleneck blocks (reference :350-397)."""

  def __init__(self, store, name, inplanes, planes, blocks, strides=1, expansion=4,
               circular_pad=False, conv_fn=Conv2D):
    partial_fn = PartialSpectralConv if conv_fn is SpectralConv else PartialConv
    downsample = None
    cin = inplanes
    if strides != 1 or inplanes != planes * expansion:
      downsample = partial_fn(store, name + '/downsample', cin, planes * expansion, 1, strides,
                              'SAME', use_bias=False)
    self.blocks = [Bottleneck(store, name + '/block0', cin, planes, strides, expansion, downsample,
                              circular_pad, partial_fn)]
    for i in range(1, blocks):
      self.blocks.append(Bottleneck(store, name + f'/block{i}', planes * expansion, planes,
                                    expansion=expansion, circular_pad=circular_pad,
                                    partial_fn=partial_fn))

  def __call__(self, ctx: Ctx, x: Var, mask=None):
    out, update_mask = self.blocks[0](ctx, x, mask)
    for block in self.blocks[1:]:
      out, update_mask = block(ctx, out, update_mask)
    return out, update_mask


class TransBasicBlock:
  """Basic block with (optionally transposed, upsampling) convolutions (reference :400-455)."""

  def __init__(self, store, name, inplanes, planes, strides=1, upsample=None, circular_pad=False,
               conv_fn=Conv2D):
    self.pad_a = PadLayer(1, circular_pad=circular_pad)
    self.conv_a = conv_fn(store, name + '/conv_a', inplanes, inplanes, 3, 1, 'VALID',
                          use_bias=False)
    self.bn_a = SyncBatchNormalization(store, name + '/bn_a', inplanes)
    self.transposed = upsample is not None and strides != 1
    if self.transposed:
      self.conv_b = Conv2DTranspose(store, name + '/conv_b', inplanes, planes, 3, strides,
                                    use_bias=False)
    else:
      self.pad_b = PadLayer(1, circular_pad=circular_pad)
      self.conv_b = conv_fn(store, name + '/conv_b', inplanes, planes, 3, strides, 'VALID',
                            use_bias=False)
    self.bn_b = SyncBatchNormalization(store, name + '/bn_b', planes)
    self.upsample = upsample

  def __call__(self, ctx: Ctx, x: Var) -> Var:
    out = self.conv_a(ctx, x, pad=self.pad_a)
    out = self.bn_a(ctx, out, act=ACT_RELU)
    if self.transposed:
      out = self.conv_b(ctx, out)
    else:
      out = self.conv_b(ctx, out, pad=self.pad_b)
    residual = x if self.upsample is None else self.upsample(ctx, x)
    return self.bn_b(ctx, out, act=ACT_RELU, res=residual)   # relu(bn(out) + residual)


class _Upsample:
  """(ConvT k2 s2 VALID | conv_fn 1x1) + SyncBN (reference :472-484)."""

  def __init__(self, store, name, inplanes, planes, strides, conv_fn):
    if strides != 1:
      self.conv = Conv2DTranspose(store, name + '/conv', inplanes, planes, 2, strides,
                                  use_bias=False)
      self.t = True
    else:
      self.conv = conv_fn(store, name + '/conv', inplanes, planes, 1, 1, 'VALID', use_bias=False)
      self.t = False
    self.bn = SyncBatchNormalization(store, name + '/bn', planes)

  def __call__(self, ctx, x):
    return self.bn(ctx, self.conv(ctx, x))


class ResStackTranspose:
  """ResNet stack of transposed blocks; upsamples when strides > 1 (reference :458-511)."""

  def __init__(self, store, name, inplanes, planes, blocks, strides=1, circular_pad=False,
               conv_fn=Conv2D):
    upsample = None
    if strides != 1 or inplanes != planes:
      upsample = _Upsample(store, name + '/upsample', inplanes, planes, strides, conv_fn)
    self.block = []
    for i in range(blocks - 1):
      self.block.append(TransBasicBlock(store, name + f'/block{i}', inplanes, inplanes,
                                        circular_pad=circular_pad, conv_fn=conv_fn))
    self.block.append(TransBasicBlock(store, name + f'/block{blocks - 1}', inplanes, planes, strides,
                                      upsample=upsample, circular_pad=circular_pad,
                                      conv_fn=conv_fn))

  def __call__(self, ctx: Ctx, x: Var, mark: Optional[str] = None, group: int = 1) -> Var:
    """mark: segment-name prefix; a gradient-synchronisation marker `mark/blocks{i}` is placed in
    front of every `group`-th block (ResNetGenerator.SEGMENTS), so that the parameter gradients of
    blocks [i, i + group) can be clipped / reduced / applied as soon as the backward pass has left
    them."""
    for i, b in enumerate(self.block):
      if mark is not None and i % group == 0:
        ctx.mark_segment(f'{mark}/blocks{i}')
      x = b(ctx, x)
    return x
