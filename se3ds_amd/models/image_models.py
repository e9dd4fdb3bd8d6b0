"""Image models used in SE3DS -- MI355X implementation of the reference's
models/image_models.py: ResNetGenerator (partial-conv ResNet encoder, global context, two
RedNet-style decoders, RGB / depth heads) and the spectral-normalised multi-scale PatchGAN
discriminator.  Constructor arguments, call signatures, return structure and error behaviour
follow the reference; see the line citations on each class."""
import os
from typing import Dict, List, Optional, Tuple

import torch

from se3ds_amd import _lib
from se3ds_amd import constants
from se3ds_amd import gin_lite as gin
from se3ds_amd.hipops import nn
from se3ds_amd.hipops.nn import ACT_LRELU, ACT_NONE, ACT_RELU, Ctx, ParamStore, Var
from se3ds_amd.models import layers


# independent branches of a model on their own HIP streams (SE3DS_DUAL_STREAM=0: one stream)
_DUAL_STREAM = os.environ.get('SE3DS_DUAL_STREAM', '1') != '0'
_DUAL_PHASES = os.environ.get('SE3DS_DUAL_PHASES', '')   # debugging: 'fwd' or 'bwd' only
# ... also with several replicas (opt-in until it has run over RCCL on two real GPUs)
_DUAL_STREAM_DP = os.environ.get('SE3DS_DUAL_STREAM_DP', '0') == '1'


def _conv_layers_of(obj, out=None, seen=None):
  """All ConvLayer objects reachable from a module tree (for the spectral-norm table)."""
  out = [] if out is None else out
  seen = set() if seen is None else seen
  if id(obj) in seen:
    return out
  seen.add(id(obj))
  if isinstance(obj, nn.ConvLayer):
    out.append(obj)
  elif isinstance(obj, (list, tuple)):
    for o in obj:
      _conv_layers_of(o, out, seen)
  elif hasattr(obj, '__dict__') and not isinstance(obj, (torch.Tensor, ParamStore)):
    for v in vars(obj).values():
      if isinstance(v, (nn.ConvLayer, list, tuple)) or hasattr(v, '__dict__'):
        _conv_layers_of(v, out, seen)
  return out


class _Model:
  """Shared plumbing: parameter store, device placement, spectral-norm group, contexts."""

  def _finish(self, device, seed, dtype):
    # seed < 0: draw the initial values on the device (fast path for the 1.1 B-parameter model)
    if seed is not None and seed < 0:
      gen = torch.Generator(device=device).manual_seed(-seed)
    else:
      gen = torch.Generator().manual_seed(seed) if seed is not None else None
    self.store.finalize(device, gen)
    self.device = torch.device(device)
    self.dtype = dtype
    self.spectral = nn.SpectralGroup(_conv_layers_of(self), self.device)

  @property
  def trainable_variables(self):
    return [self.store[n] for n in self.store.trainable_names]

  @property
  def variables(self):
    return [self.store[n] for n in self.store.trainable_names + self.store.state_names]

  def make_ctx(self, training, record=False, group=None, dtype=None, world=None) -> Ctx:
    """`world` = strategy.num_replicas_in_sync, the number of replicas SyncBatchNormalization
    sums over; `group` = their process group (None with world > 1 = the default group).
    world=None derives the count from `group` (its size; 1 without a group: a one-device model
    never all-reduces, even inside an initialised torch.distributed job).  A world that does not
    match the group's size (e.g. a group passed together with world=1) raises instead of
    silently running unsynchronised batch norm."""
    ctx = Ctx(self.device, dtype or self.dtype, training=bool(training), record=record,
              group=group, world=world)
    if record:
      # The gradient arena is created on first use.  That must happen HERE, on the stream the
      # step runs on: created lazily by the first backward closure it would be zero-filled on a
      # BRANCH stream while the other branch already writes its gradients into it (found as an
      # intermittent 1e-10 difference of the step-0 update, tools/step_compare.py).
      self.store.grad
    if (_DUAL_STREAM and nn.conv_profiler() is None and self.device.type == 'cuda' and
        (ctx.world == 1 or _DUAL_STREAM_DP)):
      # (not while the bench times single convolution launches: overlapped kernels would be
      # charged each other's time.  Several replicas: the two-stream schedule exists -- the lockstep
      # branch threads issue onto their branch's stream, paired SyncBN sums are ordered by events,
      # bit-identical to one stream on two gloo ranks sharing a GPU -- but it has never met RCCL on
      # two real GPUs (gloo blocks the host and hides stream-ordering mistakes around NCCL streams),
      # so it is opt-in, SE3DS_DUAL_STREAM_DP=1, until tests/test_dist_gpu.py's two-GPU tests have
      # passed on hardware; the default multi-replica step is the single-stream schedule)
      if getattr(self, '_branch_streams', None) is None:
        self._branch_streams = {1: nn.make_stream(self.device, 'branch1'), 2: nn.make_stream(self.device, 'branch2')}
      ctx.streams = self._branch_streams
      if _DUAL_PHASES:
        ctx.stream_phases = tuple(_DUAL_PHASES.split(','))
    return ctx


# --------------------------------------------------------------------------------- encoder
@gin.configurable
class ResNetEncoder:
  """Encoder (reference :196-303)."""

  def __init__(self, store, name, image_size: int, in_channels: int, hidden_dims: int = 64,
               resnet_version: str = '50', flatten_output: bool = True, circular_pad: bool = False,
               conv_fn=layers.Conv2D):
    if flatten_output and image_size not in [128, 256]:
      raise ValueError(f'image_size should be one of {[128, 256]}.')
    if flatten_output:
      raise NotImplementedError('flatten_output=True is never used by the SE3DS generator')
    self.pad1 = layers.PadLayer(3, circular_pad=circular_pad)
    self.conv1 = layers.PartialConv(store, name + '/conv1', in_channels, hidden_dims, 7, 2, 'VALID')
    self.bn1 = layers.SyncBatchNormalization(store, name + '/bn1', hidden_dims)
    if resnet_version == '50':
      filters = [3, 4, 6, 3]
    elif resnet_version == '101':
      filters = [3, 4, 23, 3]
    elif resnet_version == '152':
      filters = [3, 8, 36, 3]
    else:
      raise ValueError('resnet_version should be one of ["50", "101", "152"], '
                       f'got {resnet_version} instead.')
    d = hidden_dims
    self.stack1 = layers.ResStack(store, name + '/stack1', d, d, filters[0],
                                  circular_pad=circular_pad, conv_fn=conv_fn)
    self.stack2 = layers.ResStack(store, name + '/stack2', d * 4, d * 2, filters[1], strides=2,
                                  circular_pad=circular_pad, conv_fn=conv_fn)
    self.stack3 = layers.ResStack(store, name + '/stack3', d * 8, d * 4, filters[2], strides=2,
                                  circular_pad=circular_pad, conv_fn=conv_fn)
    self.stack4 = layers.ResStack(store, name + '/stack4', d * 16, d * 8, filters[3], strides=2,
                                  circular_pad=circular_pad, conv_fn=conv_fn)
    self.final_pad = layers.PadLayer(1, circular_pad=circular_pad)
    self.final_conv = layers.PartialConv(store, name + '/final_conv', d * 32, d * 4, 3, 1, 'VALID')
    self.final_bn = layers.SyncBatchNormalization(store, name + '/final_bn', d * 4)

  def __call__(self, ctx: Ctx, x: Var, mask: Optional[torch.Tensor] = None):
    # gradient-synchronisation segments (see ResNetGenerator.SEGMENTS): a marker fires in the
    # backward pass when the ops after it have all run, i.e. when the module's gradients are final
    ctx.mark_segment('encoder_head')
    out, um = self.conv1(ctx, x, mask, pad=self.pad1)
    out = self.bn1(ctx, out, act=ACT_RELU)
    b1 = out
    out = nn.maxpool2x2(ctx, out)
    um = nn.maxpool2x2_mask(ctx, um)
    ctx.mark_segment('encoder/stack1')
    out, um = self.stack1(ctx, out, um)
    s1 = out
    ctx.mark_segment('encoder/stack2')
    out, um = self.stack2(ctx, out, um)
    s2 = out
    ctx.mark_segment('encoder/stack3')
    out, um = self.stack3(ctx, out, um)
    s3 = out
    ctx.mark_segment('encoder/stack4')
    out, um = self.stack4(ctx, out, um)
    ctx.mark_segment('encoder_tail')
    out, um = self.final_conv(ctx, out, um, pad=self.final_pad)
    out = self.final_bn(ctx, out, act=ACT_RELU)
    return out, [b1, s1, s2, s3]


# --------------------------------------------------------------------------------- decoder
@gin.configurable
class ResNetDecoder:
  """Decoder (reference :306-488)."""

  def __init__(self, store, name, output_dim: int, image_size: int, hidden_dims: int = 64,
               resnet_version: str = '50', flatten_output: bool = True, circular_pad: bool = False,
               partial_conv: bool = True, conv_fn=layers.Conv2D):
    if flatten_output and image_size not in [128, 256]:
      raise ValueError(f'image_size should be one of {[128, 256]}.')
    if flatten_output:
      raise NotImplementedError('flatten_output=True is never used by the SE3DS generator')
    self.partial_conv = partial_conv
    if partial_conv:
      agent_fn = layers.PartialSpectralConv if conv_fn is layers.SpectralConv else \
          layers.PartialConv
    else:
      agent_fn = conv_fn
    d = hidden_dims
    self.name = name
    self.upc_conv = conv_fn(store, name + '/upc/conv', d * 4, d * 2, 1, 1, 'SAME')
    self.upc_bn = layers.SyncBatchNormalization(store, name + '/upc/bn', d * 2)
    def agent(nm, cin, cout):
      conv = agent_fn(store, name + '/' + nm, cin, cout, 1, 1, 'SAME', use_bias=False)
      return conv, layers.SyncBatchNormalization(store, name + '/' + nm + '_bn', cout)
    # (agent4 runs between upc and deconv1: registered there, so that the parameters of the
    # modules whose backward ends LAST -- upc, agent4 -- are contiguous in the arena, see segments())
    self.agent4, self.agent4_bn = agent('agent4', d * 2, d * 8)
    if resnet_version == '50':
      filters = [6, 4, 3, 3]
    elif resnet_version == '101':
      filters = [23, 4, 3, 3]
    elif resnet_version == '152':
      filters = [36, 8, 3, 3]
    else:
      raise ValueError('resnet_version should be one of ["50", "101", "152"], '
                       f'got {resnet_version} instead.')
    self.deconv1 = layers.ResStackTranspose(store, name + '/deconv1', d * 8, d * 4, filters[0],
                                            strides=1, circular_pad=circular_pad, conv_fn=conv_fn)
    self.deconv2 = layers.ResStackTranspose(store, name + '/deconv2', d * 4, d * 2, filters[1],
                                            strides=2, circular_pad=circular_pad, conv_fn=conv_fn)
    self.deconv3 = layers.ResStackTranspose(store, name + '/deconv3', d * 2, d, filters[2],
                                            strides=2, circular_pad=circular_pad, conv_fn=conv_fn)
    self.deconv4 = layers.ResStackTranspose(store, name + '/deconv4', d, d, filters[3],
                                            strides=2, circular_pad=circular_pad, conv_fn=conv_fn)
    self.agent0, self.agent0_bn = agent('agent0', d, d)            # skip b1 (d)
    self.agent1, self.agent1_bn = agent('agent1', d * 4, d)        # skip s1 (4d)
    self.agent2, self.agent2_bn = agent('agent2', d * 8, d * 2)    # skip s2 (8d)
    self.agent3, self.agent3_bn = agent('agent3', d * 16, d * 4)   # skip s3 (16d)
    self.final_conv = layers.ResStackTranspose(store, name + '/final_conv', d, d, 3,
                                               circular_pad=circular_pad)   # plain Conv2D
    self.final_deconv = layers.Conv2DTranspose(store, name + '/final_deconv', d, output_dim, 2, 2,
                                               use_bias=True)

  # deconv1 blocks per gradient segment.  One block per segment made the per-segment optimiser
  # kernels tiny: 49 spectral-dot launches of 365 us each (two 9.4 M-parameter layers = 128
  # workgroups, profiles/r03_*), 18 ms of GPU time per step where one batched launch took 1.3 ms.
  # Four blocks (8 layers, 150 MB of gradient) fill the chip and still stream.
  SEGMENT_BLOCKS = 4

  def segments(self):
    """{segment: parameter-name prefixes}, each contiguous in the arena.  deconv1 holds 0.96 of a
    decoder's parameters and its blocks finish one after the other during the last 3/4 of the
    decoder's backward pass: a segment per SEGMENT_BLOCKS blocks lets their gradients (and, on one
    replica, their Adam updates) stream out under the rest of it."""
    n = self.name
    segs = {n + '_head': (n + '/upc', n + '/agent4', n + '/agent4_bn', n + '/deconv1/upsample')}
    nb, g = len(self.deconv1.block), self.SEGMENT_BLOCKS
    for i in range(0, nb, g):
      segs[f'{n}/deconv1/blocks{i}'] = tuple(f'{n}/deconv1/block{j}' for j in range(i, min(i + g, nb)))
    tail = [n + '/deconv2', n + '/deconv3', n + '/deconv4']
    for k in range(4):
      tail += [f'{n}/agent{k}', f'{n}/agent{k}_bn']
    segs[n + '_tail'] = tuple(tail + [n + '/final_conv', n + '/final_deconv'])
    return segs

  def _agent(self, ctx, conv, bn, x, mask):
    if self.partial_conv:
      y, _ = conv(ctx, x, mask)
    else:
      y = conv(ctx, x)
    return bn(ctx, y, act=ACT_RELU)

  def __call__(self, ctx: Ctx, x: Var, skip: List[Var], masks=None) -> Var:
    if masks is None:
      masks = [None] * len(skip)
    # gradient-synchronisation segments in the order their backward passes END (segments()):
    # tail (everything behind deconv1), deconv1's blocks from the last to the first, head
    ctx.mark_segment(self.name + '_head')
    out = self.upc_conv(ctx, x)
    out = self.upc_bn(ctx, out, act=ACT_LRELU, alpha=0.2)
    out = nn.upsample2x(ctx, out)
    out = self._agent(ctx, self.agent4, self.agent4_bn, out, None)
    out = self.deconv1(ctx, out, mark=self.name + '/deconv1', group=self.SEGMENT_BLOCKS)
    ctx.mark_segment(self.name + '_tail')
    out = nn.add(ctx, out, self._agent(ctx, self.agent3, self.agent3_bn, skip[3], masks[3]))
    out = self.deconv2(ctx, out)
    out = nn.add(ctx, out, self._agent(ctx, self.agent2, self.agent2_bn, skip[2], masks[2]))
    out = self.deconv3(ctx, out)
    out = nn.add(ctx, out, self._agent(ctx, self.agent1, self.agent1_bn, skip[1], masks[1]))
    out = self.deconv4(ctx, out)
    out = nn.add(ctx, out, self._agent(ctx, self.agent0, self.agent0_bn, skip[0], masks[0]))
    out = self.final_conv(ctx, out)
    return self.final_deconv(ctx, out)


class _Head:
  """[BN, pad1, conv3x3, LeakyReLU(0.3)] x2 + BN, pad1, conv3x3 -> out_ch (reference :79-104)."""

  def __init__(self, store, name, d, out_ch, circular_pad, conv_fn):
    self.bn = [layers.SyncBatchNormalization(store, name + f'/bn{i}', d) for i in range(3)]
    self.pad = layers.PadLayer(1, circular_pad=circular_pad)
    self.conv = [conv_fn(store, name + '/conv0', d, d, 3, 1, 'VALID'),
                 conv_fn(store, name + '/conv1', d, d, 3, 1, 'VALID'),
                 conv_fn(store, name + '/conv2', d, out_ch, 3, 1, 'VALID')]

  def __call__(self, ctx, x):
    for i in range(3):
      # bn[i] (i > 0) is the only consumer of conv[i-1]'s LeakyReLU output
      x = self.bn[i](ctx, x, in_act=(ACT_LRELU, 0.3) if i > 0 else None)
      x = self.conv[i](ctx, x, pad=self.pad, act=ACT_LRELU if i < 2 else ACT_NONE, alpha=0.3)
    return x


# ------------------------------------------------------------------------------- generator
@gin.configurable
class ResNetGenerator(_Model):
  """ResNet generator model with partial convs (reference :27-193)."""
  # name -> parameter-name prefixes (contiguous in registration order).  The encoder is split per
  # stage so that only its first, tiny stages are reduced after the backward pass has ended.
  _SEGMENTS = {
      'encoder_head': ('encoder/conv1', 'encoder/bn1'),
      'encoder/stack1': ('encoder/stack1',), 'encoder/stack2': ('encoder/stack2',),
      'encoder/stack3': ('encoder/stack3',), 'encoder/stack4': ('encoder/stack4',),
      'encoder_tail': ('encoder/final_conv', 'encoder/final_bn'),
      'rgb_conv': ('rgb_conv',), 'depth_conv': ('depth_conv',), 'context': ('context',)}
  # (+ the decoders' own segments: SEGMENTS is completed per instance in __init__)

  def __init__(self, image_size: int = 256, gen_dims: int = 96, z_dim: int = 128,
               resnet_version: str = '50', context_layer: str = 'convs',
               conv_mode: str = 'spectral', use_blurred_mask: bool = True, device='cuda',
               seed: Optional[int] = 0, dtype=torch.float32):
    self.hidden_dims = gen_dims
    self.resnet_version = resnet_version
    self.z_dim = z_dim
    self.circular_pad = True
    if context_layer not in ['convs', 'none']:
      raise NotImplementedError
    self.context_layer = context_layer
    self.use_blurred_mask = use_blurred_mask
    conv_fn = layers.SpectralConv if conv_mode == 'spectral' else layers.Conv2D
    self.store = ParamStore()
    d, s = gen_dims, self.store
    cin = 5 if use_blurred_mask else 4
    self.encoder = ResNetEncoder(s, 'encoder', image_size=image_size, in_channels=cin,
                                 hidden_dims=d, resnet_version=resnet_version,
                                 flatten_output=False, circular_pad=True, conv_fn=conv_fn)
    self.decoder = ResNetDecoder(s, 'decoder', output_dim=d, image_size=image_size, hidden_dims=d,
                                 resnet_version=resnet_version, flatten_output=False,
                                 circular_pad=True, conv_fn=conv_fn)
    self.depth_decoder = ResNetDecoder(s, 'depth_decoder', output_dim=d, image_size=image_size,
                                       hidden_dims=d, resnet_version=resnet_version,
                                       flatten_output=False, circular_pad=True, conv_fn=conv_fn)
    self.rgb_conv = _Head(s, 'rgb_conv', d, 3, True, conv_fn)
    self.depth_conv = _Head(s, 'depth_conv', d, 1, True, conv_fn)
    if context_layer == 'convs':
      chans = [d * 4, d * 4, d * 8, d * 4, d * 4]
      self.ctx_bn = [layers.SyncBatchNormalization(s, f'context/bn{i}', chans[i]) for i in range(4)]
      self.ctx_pad = layers.PadLayer(1, circular_pad=True)
      self.ctx_conv = [layers.SpectralConv(s, f'context/conv{i}', chans[i], chans[i + 1], 3, 1,
                                           'VALID') for i in range(4)]
    self.SEGMENTS = dict(self._SEGMENTS)
    self.SEGMENTS.update(self.decoder.segments())
    self.SEGMENTS.update(self.depth_decoder.segments())
    self._finish(device, seed, dtype)

  # -- differentiable forward used by the trainer
  def forward(self, ctx: Ctx, cond: Dict[str, torch.Tensor]):
    """Returns (outputs list as the reference, (rgb_push_grad, depth_push_grad))."""
    gi, gd = cond['proj_image'], cond['proj_depth']
    gm, bm = cond['proj_mask'], cond['blurred_mask']
    _lib.require_cuda(gi, gd, gm, bm)
    parts = [gi, gd, bm] if self.use_blurred_mask else [gi, gd]
    x = Var(nn.concat_channels(ctx, [p.to(torch.float32) for p in parts]), requires_grad=False)
    n, h, w, _ = x.shape
    mask = gm.to(torch.float32).reshape(n, h, w).contiguous()
    # The power iteration reads every spectral kernel twice (2.8 ms alone, HBM-bound) and nothing in
    # the encoder needs its result: PartialSpectralConv convolves with the raw kernel
    # (layers.py:189-195), sigma is first applied by the context convs.  With branch streams it runs
    # on the first decoder's stream under the encoder's forward pass; the main stream waits for it in
    # front of the context module (the decoders' streams are ordered behind the main stream there).
    # Same-box A/B: 190.9 / 191.1 vs 191.6 / 191.8 ms per step; bit-identical to the serial order
    # (tools/step_compare.py).
    sn_ev = None
    if ctx.streams is not None and 'fwd' in ctx.stream_phases:
      side = ctx.streams[1]
      side.wait_stream(torch.cuda.current_stream(ctx.device))
      with torch.cuda.stream(side):
        self.spectral.power_iteration(ctx.training)
        sn_ev = torch.cuda.Event()
        sn_ev.record()
    else:
      self.spectral.power_iteration(ctx.training)
    hidden, skip = self.encoder(ctx, x, mask)   # (marks its own segments)
    if sn_ev is not None:
      torch.cuda.current_stream(ctx.device).wait_event(sn_ev)
    taps = getattr(ctx, 'taps', None)
    if taps is not None:
      taps.update(b1=skip[0], s1=skip[1], s2=skip[2], s3=skip[3], enc=hidden)
    if self.context_layer == 'convs':
      ctx.mark_segment('context')
      for i in range(4):
        hidden = self.ctx_bn[i](ctx, hidden, in_act=(ACT_LRELU, 0.3) if i > 0 else None)
        hidden = self.ctx_conv[i](ctx, hidden, pad=self.ctx_pad,
                                  act=ACT_LRELU if i < 3 else ACT_NONE, alpha=0.3)
    hh, hw = hidden.shape[1], hidden.shape[2]
    # The two decoders (+ their heads) are independent branches: each runs on its own HIP stream
    # when the context has them (Ctx.branch; one replica only).  Their common inputs receive
    # gradients from both streams.
    if ctx.streams is not None:
      for v in list(skip) + [hidden]:
        v.shared = True
    def rgb_branch():
      o = self.decoder(ctx, hidden, skip)   # (marks its own segments)
      ctx.mark_segment('rgb_conv')
      pre = self.rgb_conv(ctx, o)
      return (o,) + nn.head(ctx, pre, 0)
    def depth_branch():
      o = self.depth_decoder(ctx, hidden, skip)
      ctx.mark_segment('depth_conv')
      pre = self.depth_conv(ctx, o)
      return (o,) + nn.head(ctx, pre, 1)
    # one replica: each branch on its own HIP stream; several: lockstep, paired SyncBN sums
    res = ctx.run_branches({1: rgb_branch, 2: depth_branch})
    ctx.join()
    out, rgb, push_rgb = res[1]
    depth_out, depth, push_depth = res[2]
    if taps is not None:
      taps.update(ctx=hidden, dec=out, ddec=depth_out)
    # mu / logvar / kld / seg / depth_seg are constant zeros (reference :308-312, :191-193): one
    # allocation per shape instead of 1.4 GB of memset per 512 x 1024 batch-8 step.  Consumers
    # only read them (as the reference's do: tf.zeros are immutable).
    zc = self.__dict__.setdefault('_zeros', {})
    def zeros(*shape):
      key = (shape, str(ctx.device))
      if key not in zc:
        zc[key] = torch.zeros(shape, dtype=torch.float32, device=ctx.device)
      return zc[key]
    mu = zeros(n, hh, hw, self.z_dim)
    seg = zeros(n, h, w, constants.NUM_MP3D_CLASSES)
    outs = [mu, mu, mu, depth, seg, seg, rgb]
    return outs, (push_rgb, push_depth)

  def __call__(self, inputs, sample_noise: bool = False, training=None) -> List[torch.Tensor]:
    """inputs = [cond_dict, noise(ignored)] -> [mu, logvar, kld, depth, seg, depth_seg, rgb]."""
    if sample_noise:
      raise ValueError('This model does not support noise sampling!')
    cond, _ = inputs
    ctx = self.make_ctx(training)
    outs, _ = self.forward(ctx, cond)
    return outs


# --------------------------------------------------------------------------- discriminator
class SNPatchDiscriminator:
  """Spectral-normalised PatchGAN discriminator (reference :492-561)."""

  def __init__(self, store, name, in_channels=4, kernel_size: int = 4, dis_dims: int = 64,
               n_layers: int = 4, circular_pad: bool = False):
    k = kernel_size
    self.pad = layers.PadLayer(k // 2, circular_pad=circular_pad)
    self.conv0 = layers.Conv2D(store, name + '/g0/conv', in_channels, dis_dims, k, 2, 'VALID')
    self.groups = []
    prev = dis_dims
    for i in range(1, n_layers):
      cur = min(prev * 2, 512)
      conv = layers.SpectralConv(store, name + f'/g{i}/conv', prev, cur, k,
                                 2 if i != n_layers - 1 else 1, 'VALID')
      inorm = layers.InstanceNormalization(store, name + f'/g{i}/in', cur)
      self.groups.append((conv, inorm))
      prev = cur
    self.final = layers.Conv2D(store, name + '/final', prev, 1, k, 1, 'SAME')

  def __call__(self, ctx: Ctx, x: Var) -> List[Var]:
    results = []
    out = self.conv0(ctx, x, pad=self.pad, act=ACT_LRELU, alpha=0.2)
    results.append(out)
    for conv, inorm in self.groups:
      out = conv(ctx, out, pad=self.pad)
      out = inorm(ctx, out, act=ACT_LRELU, alpha=0.2)
      results.append(out)
    out = self.final(ctx, out)
    results.append(out)
    return results


@gin.configurable
class SNMultiScaleDiscriminator(_Model):
  """Spectral-normalised multi-scale PatchGAN discriminator (reference :564-618)."""

  def __init__(self, image_size: int = 256, n_dis: int = 2, kernel_size: int = 4,
               dis_dims: int = 96, n_layers: int = 5, circular_pad: bool = False,
               in_channels: int = 4, device='cuda', seed: Optional[int] = 1,
               dtype=torch.float32):
    del image_size  # fully convolutional
    self.store = ParamStore()
    self.discriminators = [
        SNPatchDiscriminator(self.store, f'dis{i}', in_channels, kernel_size, dis_dims, n_layers,
                             circular_pad) for i in range(n_dis)]
    self._finish(device, seed, dtype)

  def forward(self, ctx: Ctx, x: Var) -> List[List[Var]]:
    self.spectral.power_iteration(ctx.training)
    result = []
    prev = x
    self._scale_inputs = []
    for i, model in enumerate(self.discriminators):
      result.append(model(ctx, prev))
      if i + 1 < len(self.discriminators):
        prev = nn.avgpool3s2(ctx, prev)
        self._scale_inputs.append(prev)
    return result

  def __call__(self, inputs: torch.Tensor, training=None) -> List[List[torch.Tensor]]:
    """inputs (N,H,W,C) fp32 -> list (per scale) of lists of feature maps (fp32)."""
    _lib.require_cuda(inputs)
    ctx = self.make_ctx(training)
    x = nn.to_var(ctx, inputs.to(torch.float32))
    res = self.forward(ctx, x)
    return [[nn.slice_channels(v.data, 0, v.data.shape[-1], torch.float32) for v in sub]
            for sub in res]
