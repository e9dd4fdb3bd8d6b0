"""Trainer for the SE3DS model -- MI355X implementation of the reference's
trainers/se3ds_trainer.py: the hinge-GAN step (`train_g_d` :129-273, `train_d` :275-338) with
its losses (:27-71).  Same class name, constructor arguments, attribute names and metric keys;
all arithmetic runs in libse3ds_hip.so.

Reference semantics kept on purpose:
  * the generator loss is an (N,)-vector (the world-consistency term is not reduced) and its
    gradient is the sum of the elements: G gradients carry a factor N for the scalar terms;
  * losses are pre-divided by the replica count, gradients are clipped per tensor and per
    replica BEFORE the cross-replica sum;
  * hinge terms use only the last feature map of each sub-discriminator;
  * `train_d` runs the generator in training mode outside any tape (BN moving statistics and
    spectral `u` advance on D-only steps too).
"""
import os
from typing import Dict

import torch

from se3ds_amd import _lib
from se3ds_amd import gin_lite as gin
from se3ds_amd.hipops import nn
from se3ds_amd.hipops.nn import Var
from se3ds_amd.models import image_models  # noqa: F401  (gin references)
from se3ds_amd.trainers import dist_utils
from se3ds_amd.trainers import gan_manager
from se3ds_amd.trainers.gan_manager import Mean

# the generator's spectral gradient fix-up rides on the clip pass
FUSED_SN_CLIP = True
GRAD_CLIP_NORM = 5.0   # _clip_grad default, reference :27
# one replica: per-module clip + Adam on a side stream under the backward pass (0: after it)
SEGMENT_OPTIMIZER = os.environ.get('SE3DS_SEGMENT_OPTIMIZER', '1') != '0'
# one replica: the discriminator's parameter-gradient pass on a side stream under the generator's
# backward pass (train_g_d).  Bit-identical; measured 213.8 / 214.1 ms against 214.1 / 214.6 ms per
# step on one box (-0.2 %, inside the noise) for 0.7 GB more memory: off, kept for A/B runs.
D_OVERLAP = os.environ.get('SE3DS_D_OVERLAP', '0') != '0'
# one replica: weight gradients on their own stream (Ctx.on_wgrad_stream).  Bit-identical, but
# measured SLOWER (221.0 vs 213.5 ms per step, same box): two MFMA-bound kernels co-running cost
# more than the dependency chain gains.  Off; kept for A/B runs.
WGRAD_STREAM = os.environ.get('SE3DS_WGRAD_STREAM', '0') != '0'
# one replica: per-tensor clip (+ spectral fix-up) inside the Adam pass -- the gradient arena is read
# once and never rewritten (AdamState.clip_apply).  Bit-identical; 0 = the two separate passes.
FUSED_CLIP_ADAM = os.environ.get('SE3DS_FUSED_CLIP_ADAM', '1') != '0'
# the split reductions of a module's weight gradients in ONE launch when the backward pass has left
# the module (nn.flush_wgrad_reduces; one replica: on the optimiser's side stream) instead of one
# launch behind every weight-gradient kernel.  Bit-identical; 0 = reduce right away.
DEFER_WGRAD_REDUCE = os.environ.get('SE3DS_DEFER_WGRAD_REDUCE', '1') != '0'


def _L():
  return _lib.lib()


@gin.configurable(denylist=['strategy', 'model_dir'])
class GAN(gan_manager.GANManager):
  """One stage GAN (reference :74-93)."""

  def __init__(self, lambda_gan, lambda_kld, lambda_wc, lambda_depth, *args,
               dis_use_pred_depth=True, mask_blurred=False, **kwargs):
    super().__init__(*args, **kwargs)
    self.lambda_gan = lambda_gan
    self.lambda_kld = lambda_kld
    self.lambda_wc = lambda_wc
    self.lambda_depth = lambda_depth
    self.dis_use_pred_depth = dis_use_pred_depth
    self.mask_blurred = mask_blurred

  def _create_metrics(self):
    """Metric names of the reference (:107-127)."""
    names = ['gen/gen_loss', 'dis/disc_loss', 'dis/grad_norm', 'gen/gen_feat_loss',
             'gen/gen_gan_loss', 'gen/depth_loss', 'gen/seg_loss', 'gen/depth_seg_loss',
             'gen/depth_seg_consistency', 'gen/kld_loss', 'gen/kld_nan', 'gen/wc_loss',
             'gen/grad_norm']
    self.metrics = {n: Mean(n.split('/')[-1]) for n in names}

  # --------------------------------------------------------------------------- D plumbing
  def _prep_inputs(self, inputs):
    inputs = dict(inputs)
    for k in ('image', 'proj_image', 'proj_depth', 'proj_mask', 'depth', 'blurred_mask'):
      t = inputs[k]
      _lib.require_cuda(t)
      inputs[k] = t.to(torch.float32).contiguous()
    if not self.mask_blurred:
      bm = torch.empty_like(inputs['blurred_mask'])
      _lib.check(_L().se3ds_fill(bm.data_ptr(), _lib.F32, bm.numel(), 0.0, _lib.stream()),
                 'se3ds_fill')
      inputs['blurred_mask'] = bm
    return inputs

  def _disc_input(self, ctx, generated, depth_out, image, depth):
    """concat([fake(rgb|depth), real(image|depth)], axis=0) -> (2N,H,W,4) (reference :181-192)."""
    n = image.shape[0]
    fake_d = depth_out if self.dis_use_pred_depth else depth
    fake = nn.concat_channels(ctx, [generated, fake_d])
    real = nn.concat_channels(ctx, [image, depth])
    x = ctx.empty((2 * n,) + tuple(fake.shape[1:]))
    x[:n].copy_(fake)
    x[n:].copy_(real)
    return Var(x, requires_grad=False)

  def _run_discriminator(self, ctx, x_all):
    res = self.discriminator.forward(ctx, x_all)
    return res

  def _hinge(self, ctx, logits, coef_d, coef_g, want_g):
    """Per sub-discriminator hinge sums + gradient seeds on the LAST feature map only."""
    sums, seeds_d, seeds_g = [], [], []
    for sub in logits:
      last = sub[-1].data
      half = last.numel() // 2
      s = torch.empty(2, dtype=torch.float32, device=ctx.device)
      gd = torch.empty_like(last)
      gg = torch.empty_like(last) if want_g else None
      cnt = float(half)
      _lib.check(_L().se3ds_hinge(last.data_ptr(), ctx.code, half, coef_d / cnt, coef_g / cnt,
                                  s.data_ptr(), gd.data_ptr(), _lib.ptr(gg), _lib.stream()),
                 'se3ds_hinge')
      sums.append((s, cnt))
      seeds_d.append(gd)
      seeds_g.append(gg)
    return sums, seeds_d, seeds_g

  def _segments(self):
    """{module: (first tensor, end tensor, first element, end element)} of the generator's arena."""
    if getattr(self, '_g_segments', None) is None:
      G = self.generator
      self._g_segments = G.store.segments(G.SEGMENTS)
      covered = sum(t1 - t0 for t0, t1, _, _ in self._g_segments.values())
      assert covered == len(G.store.trainable_names), 'generator segments miss some tensors'
    return self._g_segments

  def _all_conv_layers(self, model):
    cache = self.__dict__.setdefault('_conv_layer_cache', {})
    if id(model) not in cache:
      from se3ds_amd.models import image_models
      cache[id(model)] = image_models._conv_layers_of(model)
    return cache[id(model)]

  def _segment_layers(self, G):
    """{module: [its ConvLayers]} by parameter-name prefix."""
    if getattr(self, '_seg_layers', None) is None:
      layers = self._all_conv_layers(G)
      self._seg_layers = {
          name: [l for l in layers if any(l.name.startswith(pre + '/') or l.name == pre
                                          for pre in prefixes)]
          for name, prefixes in G.SEGMENTS.items()}
      assert sum(len(v) for v in self._seg_layers.values()) == len(layers), 'segments miss conv layers'
    return self._seg_layers

  def _operand_group(self, model, segment, dtype):
    """nn.OperandGroup of one generator segment (or of a whole model: segment None)."""
    cache = self.__dict__.setdefault('_operand_groups', {})
    key = (id(model), segment, dtype)
    if key not in cache:
      layers = self._all_conv_layers(model) if segment is None else self._segment_layers(model)[segment]
      cache[key] = nn.OperandGroup(layers, dtype, model.store.theta.device)
    return cache[key]

  def _d_stream(self, dev):
    if getattr(self, '_dis_stream', None) is None:
      self._dis_stream = torch.cuda.Stream(dev)
    return self._dis_stream

  def _wgrad_stream(self, dev):
    if getattr(self, '_wg_stream', None) is None:
      self._wg_stream = torch.cuda.Stream(dev)
    return self._wg_stream

  def _optimizer_stream(self, dev):
    if getattr(self, '_opt_stream', None) is None:
      self._opt_stream = nn.make_stream(dev, 'optimizer')
    return self._opt_stream

  def _grad_sync(self):
    """GradSync when gradients cross replicas (or SE3DS_FORCE_GRAD_SYNC=1, which exercises the
    segment-wise clip / side-stream path on one GPU), else None."""
    R = self.strategy.num_replicas_in_sync
    if R == 1 and os.environ.get('SE3DS_FORCE_GRAD_SYNC') != '1':
      return None
    if getattr(self, '_sync', None) is None:
      G = self.generator
      group = self.strategy.group
      own = R > 1 and os.environ.get('SE3DS_GRAD_SYNC_OWN_COMM') == '1'
      if own:
        # opt-in: a second communicator keeps the 4.5 GB of gradient traffic from queueing ahead
        # of the small SyncBN statistics all-reduces the backward pass waits on.  Two RCCL
        # communicators in flight on one device are only safe when every rank issues them in the
        # same order (true here: the program is deterministic) AND the device can co-schedule
        # both kernels; until that is validated on an 8-GPU node the default is the shared
        # communicator, where collectives simply serialise in issue order.
        group = dist_utils.clone_group(group)
        # build the communicator now, at a point every rank reaches together, instead of
        # lazily at the first bucket in the middle of the backward pass
        warm = torch.zeros(1, dtype=torch.float32, device=G.store.theta.device)
        torch.distributed.all_reduce(warm, group=group)
        torch.cuda.synchronize(G.store.theta.device)
      # shared communicator: buckets are drip-fed behind the SyncBN collectives (GradSync)
      self._sync = dist_utils.GradSync(G.store.theta.device, group, drip=not own)
      self._segments()
    return self._sync

  def _sync_discriminator(self, sync):
    """Discriminator gradients are final after pass 1: clip, then all-reduce on the side
    stream while the main stream runs pass 2 and the generator backward."""
    D = self.discriminator
    norm = self.d_optimizer.clip_gradients(GRAD_CLIP_NORM).clone()
    sync.reduce_range(D.store.grad, 0, D.store.grad.numel())
    return norm

  def _backward_tape(self, ctx, tape, seeds, logits):
    for sub, g in zip(logits, seeds):
      sub[-1].grad = g
    for entry in reversed(tape):   # (fn, stream branch, sync flag): no branches in the discriminator
      entry[0]()

  # ------------------------------------------------------------------------------ train_g_d
  def train_g_d(self, inputs: Dict[str, torch.Tensor]) -> None:
    """One generator + discriminator update (reference :129-273)."""
    inputs = self._prep_inputs(inputs)
    image, depth_t = inputs['image'], inputs['depth']
    n, h, w, _ = image.shape
    p = h * w
    R = self.strategy.num_replicas_in_sync
    group = self.strategy.group
    G, D = self.generator, self.discriminator
    L = _L()
    dev = image.device
    f32 = dict(dtype=torch.float32, device=dev)
    sync = self._grad_sync()   # (first call builds the gradient communicator: do it up front)

    # ---- generator forward (both "tapes" of the reference share this forward)
    ctx_g = G.make_ctx(training=True, record=True, group=group, world=R)
    # PartialConv fast paths assume {0, 1} masks (what the reference's input pipeline produces,
    # indoor_datasets.py:281-304,577-585); `gan.binary_masks = False` takes the exact kernels for
    # fractional proj_mask values
    ctx_g.binary_masks = bool(getattr(self, 'binary_masks', True))
    outs, (push_rgb, push_depth) = G.forward(ctx_g, inputs)
    depth_out, generated = outs[3], outs[6]

    # ---- per-sample normalisers and loss sums
    ssws = torch.empty(n * 256, **f32)  # scratch of the two-stage per-sample reductions
    npx = torch.empty(n, **f32)        # valid-depth pixel counts (:148-152)
    _lib.check(L.se3ds_sample_sum(depth_t.data_ptr(), None, None, n, p, 1, 2, npx.data_ptr(),
                                  ssws.data_ptr(), _lib.stream()), 'se3ds_sample_sum')
    wcm = torch.empty(n, **f32)        # sum of proj_mask * (1 - blurred_mask) (:176-178)
    _lib.check(L.se3ds_sample_sum(inputs['proj_mask'].data_ptr(), inputs['blurred_mask'].data_ptr(),
                                  None, n, p, 1, 3, wcm.data_ptr(), ssws.data_ptr(),
                                  _lib.stream()),
               'se3ds_sample_sum')
    # gradient coefficients of the SUM over the (N,)-vector loss, already / replicas
    coef_depth = torch.empty(n, **f32)
    coef_wc = torch.empty(n, **f32)
    depth_scale = (self.lambda_depth / R) if self.predict_depth else 0.0
    _lib.check(L.se3ds_recip_clamp(npx.data_ptr(), n, depth_scale, coef_depth.data_ptr(),
                                   _lib.stream()), 'se3ds_recip_clamp')
    _lib.check(L.se3ds_recip_clamp(wcm.data_ptr(), n, self.lambda_wc / (3.0 * R),
                                   coef_wc.data_ptr(), _lib.stream()), 'se3ds_recip_clamp')
    d_depth = torch.empty_like(depth_out)
    _lib.check(L.se3ds_l1_grad(depth_out.data_ptr(), depth_t.data_ptr(), None, None,
                               coef_depth.data_ptr(), n, p, 1, 0, d_depth.data_ptr(),
                               _lib.stream()), 'se3ds_l1_grad')
    d_rgb = torch.empty_like(generated)
    _lib.check(L.se3ds_l1_grad(generated.data_ptr(), inputs['proj_image'].data_ptr(),
                               inputs['proj_mask'].data_ptr(), inputs['blurred_mask'].data_ptr(),
                               coef_wc.data_ptr(), n, p, 3, 1, d_rgb.data_ptr(), _lib.stream()),
               'se3ds_l1_grad')
    # metric sums (read lazily)
    depth_l1 = torch.empty(n, **f32)
    tmask = torch.empty((n, p), **f32)   # 1[0 < depth < 1] as an explicit mask for mode 1
    _lib.check(L.se3ds_l1_grad(depth_t.data_ptr(), depth_t.data_ptr(), None, None, None, n, p, 1, 2,
                               tmask.data_ptr(), _lib.stream()), 'se3ds_l1_grad')
    _lib.check(L.se3ds_sample_sum(depth_out.data_ptr(), depth_t.data_ptr(), tmask.data_ptr(), n, p,
                                  1, 1, depth_l1.data_ptr(), ssws.data_ptr(), _lib.stream()),
               'se3ds_sample_sum')
    wc_l1 = torch.empty(n, **f32)
    wmask = torch.empty((n, p), **f32)
    _lib.check(L.se3ds_l1_grad(inputs['proj_mask'].data_ptr(), inputs['proj_mask'].data_ptr(),
                               inputs['proj_mask'].data_ptr(), inputs['blurred_mask'].data_ptr(),
                               None, n, p, 1, 3, wmask.data_ptr(), _lib.stream()), 'se3ds_l1_grad')
    _lib.check(L.se3ds_sample_sum(generated.data_ptr(), inputs['proj_image'].data_ptr(),
                                  wmask.data_ptr(), n, p, 3, 1, wc_l1.data_ptr(), ssws.data_ptr(),
                                  _lib.stream()),
               'se3ds_sample_sum')

    # ---- discriminator forward on [fake; real]
    ctx_d = D.make_ctx(training=True, record=True, group=group, world=R)
    x_all = self._disc_input(ctx_d, generated, depth_out, image, depth_t)
    logits = self._run_discriminator(ctx_d, x_all)
    n_dis = len(logits)
    # d(disc_loss / R)/dlogit and d(sum_i gen_loss / R)/dlogit = N * d(gen_loss / R)/dlogit
    coef_d = self.lambda_gan / (n_dis * R)
    coef_g = self.lambda_gan * n / (n_dis * R)
    hinge_sums, seeds_d, seeds_g = self._hinge(ctx_d, logits, coef_d, coef_g, want_g=True)
    tape_d = ctx_d.tape
    ctx_d.tape = None

    def pass1():
      # discriminator parameter gradients (disc_tape, :244)
      ctx_d.param_grads = True
      ctx_d.batch_limit = None
      self._set_input_grad(x_all, False)
      self._backward_tape(ctx_d, tape_d, seeds_d, logits)
      D.spectral.backward_fixup()

    def pass2():
      # gradient of the generator loss w.r.t. the fake images (gen_tape, :236).  Only the fake
      # half of [fake; real] carries generator loss, and nothing in the discriminator couples
      # samples (instance norm, no batch statistics): the pass runs on the first n samples of
      # every saved activation.
      ctx_d.param_grads = False
      ctx_d.batch_limit = n
      self._set_input_grad(x_all, True)
      self._backward_tape(ctx_d, tape_d, [g[:n] for g in seeds_g], logits)
      ctx_d.batch_limit = None
      g = x_all.grad
      x_all.grad = None
      return g

    side_opt = sync is None and SEGMENT_OPTIMIZER and nn.conv_profiler() is None
    d_done = None
    if side_opt and D_OVERLAP:
      # One replica: the generator's backward pass only needs pass 2.  Pass 1 (the discriminator's
      # own parameter gradients over [fake; real]) is issued behind it on a side stream and runs
      # under the generator's backward pass; the discriminator's activations stay alive until the
      # streams have joined (tape_d is released at the end of the step).
      gx = pass2()
      dstream = self._d_stream(dev)
      dstream.wait_stream(torch.cuda.current_stream(dev))
      with torch.cuda.stream(dstream):
        pass1()
        d_done = torch.cuda.Event()
        d_done.record()
    else:
      pass1()
      if sync is not None:
        d_norm = self._sync_discriminator(sync)
      gx = pass2()
      del tape_d
    g_rgb = nn.slice_channels(gx[:n], 0, 3, torch.float32)
    _lib.check(L.se3ds_add(d_rgb.data_ptr(), g_rgb.data_ptr(), _lib.F32, d_rgb.numel(),
                           d_rgb.data_ptr(), _lib.stream()), 'se3ds_add')
    if self.dis_use_pred_depth:
      g_dep = nn.slice_channels(gx[:n], 3, 1, torch.float32)
      _lib.check(L.se3ds_add(d_depth.data_ptr(), g_dep.data_ptr(), _lib.F32, d_depth.numel(),
                             d_depth.data_ptr(), _lib.stream()), 'se3ds_add')
    # ---- generator backward
    ctx_g.param_grads = True
    if ctx_g.streams is not None and WGRAD_STREAM and sync is None:
      # ONE replica only: the conv layers' weight gradients leave the dgrad -> norm -> dgrad chain.
      # (With several replicas the segment hand-over below orders the optimiser's stream behind the
      # module's backward stream only: a wgrad stream would let clip / all-reduce read slabs and
      # gradients whose kernels are still running.)
      ctx_g.wgrad_stream = self._wgrad_stream(dev)
    push_rgb(d_rgb)
    push_depth(d_depth)
    # ---- clip per tensor (per replica), aggregate, apply (:238-257)
    ema_theta, ema_omd = self.ema_fused_args()   # EMA of the trainable variables rides on Adam
    if side_opt:
      # (not while the bench times single convolution launches: see _Model.make_ctx)
      # One replica: a module's spectral fix-up, per-tensor clip and Adam (+ EMA) update run on a
      # SIDE STREAM as soon as the backward pass has left the module, under the rest of the
      # backward pass (HBM-bound optimiser passes under MFMA-bound convolutions; the two decoders
      # hold 0.9 of the parameters and finish first).  Per-tensor clipping needs no global norm,
      # so the update is bit-identical to the serial order below.
      segs = self._segments()
      opt = self._optimizer_stream(dev)
      main = torch.cuda.current_stream(dev)
      ev = d_done
      if ev is None:
        ev = torch.cuda.Event()
        ev.record()
      with torch.cuda.stream(opt):   # discriminator: its gradients are final behind pass 1
        opt.wait_event(ev)
        self._update_all(self.d_optimizer, False, None, 0.0)
        d_norm = self.d_optimizer.mean_clipped_norm(GRAD_CLIP_NORM).clone()
      self.g_optimizer.begin_step()
      def segment_done(name):
        if name not in segs:
          return
        t0, t1, e0, e1 = segs[name]
        done = torch.cuda.Event()
        done.record()   # on the stream that ran the module's backward (main or a decoder branch)
        wdone = ctx_g.wgrad_event()   # ... and behind its weight gradients
        with torch.cuda.stream(opt):
          opt.wait_event(done)
          if wdone is not None:
            opt.wait_event(wdone)
          nn.flush_wgrad_reduces(ctx_g)   # (the module's deferred split reductions, one launch)
          G.spectral.backward_fixup(prefix=G.SEGMENTS[name], dots_only=FUSED_SN_CLIP)
          if not (FUSED_CLIP_ADAM and self.g_optimizer.clip_apply(
              t0, t1, GRAD_CLIP_NORM, FUSED_SN_CLIP, ema_theta, ema_omd)):
            self.g_optimizer.clip_segment(t0, t1, GRAD_CLIP_NORM, fused_sn=FUSED_SN_CLIP)
            self.g_optimizer.apply_segment(e0, e1, ema_theta, ema_omd)
          # ... and the module's compute-dtype operand copies for the NEXT step (its backward is
          # over, nothing reads the old copies any more): 315 launches leave the critical path
          self._operand_group(G, name, ctx_g.dtype).prep(G.store.version + 1)
      ctx_g.on_segment = segment_done
      if DEFER_WGRAD_REDUCE and ctx_g.wgrad_stream is None:
        ctx_g.wgrad_defer = []
      ctx_g.backward()
      ctx_g.on_segment = None
      ctx_g.wgrad_defer = None
      with torch.cuda.stream(opt):   # (the discriminator's pass 2 read its old copies until now)
        opt.wait_stream(main)
        self._operand_group(D, None, ctx_d.dtype).prep(D.store.version)
      main.wait_stream(opt)
      self.g_optimizer.end_step()
      g_norm = self.g_optimizer.mean_clipped_norm(GRAD_CLIP_NORM).clone()
    elif sync is None:
      # serial order (SE3DS_SEGMENT_OPTIMIZER=0, and the bench's instrumented step): the split
      # reductions of ALL weight gradients go out as one table-driven launch behind the backward
      # pass, the operand copies of a model as one launch behind its update (round 5: 295 + 290
      # launches per step less on this path too)
      if DEFER_WGRAD_REDUCE and ctx_g.wgrad_stream is None:
        ctx_g.wgrad_defer = []
      ctx_g.backward()
      ctx_g.wgrad_defer = None
      G.spectral.backward_fixup(dots_only=FUSED_SN_CLIP)
      self._update_all(self.g_optimizer, FUSED_SN_CLIP, ema_theta, ema_omd)
      g_norm = self.g_optimizer.mean_clipped_norm(GRAD_CLIP_NORM).clone()
      self._update_all(self.d_optimizer, False, None, 0.0)
      d_norm = self.d_optimizer.mean_clipped_norm(GRAD_CLIP_NORM).clone()
      self._operand_group(G, None, ctx_g.dtype).prep(G.store.version)
      self._operand_group(D, None, ctx_d.dtype).prep(D.store.version)
    else:
      # Several replicas (round 4: the SAME stream schedule as the benchmarked one-replica step):
      # when the backward pass leaves a module -- on the main stream or on a decoder's branch
      # stream -- its deferred split reductions, spectral fix-up and per-tensor clip run on the
      # optimiser's side stream behind an event (they share the optimiser's scratch: one stream
      # for all of them), and the clipped slice is handed to the gradient all-reduce from there;
      # the all-reduce overlaps the rest of the backward pass
      opt = self._optimizer_stream(dev)
      main = torch.cuda.current_stream(dev)
      opt.wait_stream(main)
      def segment_done(name):
        if name not in self._g_segments:
          return
        t0, t1, e0, e1 = self._g_segments[name]
        done = torch.cuda.Event()
        done.record()   # on the stream that ran the module's backward
        with torch.cuda.stream(opt):
          opt.wait_event(done)
          nn.flush_wgrad_reduces(ctx_g)
          G.spectral.backward_fixup(prefix=G.SEGMENTS[name], dots_only=FUSED_SN_CLIP)
          self.g_optimizer.clip_segment(t0, t1, GRAD_CLIP_NORM, fused_sn=FUSED_SN_CLIP)
          sync.reduce_range(G.store.grad, e0, e1)   # (its `ready` event lands on this stream)
      ctx_g.on_segment = segment_done
      ctx_g.after_collective = sync.pump
      assert ctx_g.wgrad_stream is None   # (see above: one replica only)
      if DEFER_WGRAD_REDUCE:
        ctx_g.wgrad_defer = []
      ctx_g.backward()
      ctx_g.on_segment = None
      ctx_g.after_collective = None
      ctx_g.wgrad_defer = None
      main.wait_stream(opt)
      g_norm = self.g_optimizer.mean_clipped_norm(GRAD_CLIP_NORM).clone()
      sync.finish()
      self.g_optimizer.apply_gradients(group, 1, ema_theta, ema_omd)
      self.d_optimizer.apply_gradients(group, 1)
    self.last_collectives = ctx_g.collectives   # SyncBN all-reduces of this step (tests)
    if self.global_step == 0:
      # builds the EMA model's variables in the reference (:258-259): a throw-away forward
      self.ema_generator.forward(self.ema_generator.make_ctx(training=True, group=group, world=R), inputs)
    self.update_ema_model(theta_done=ema_theta is not None)

    # ---- metrics (:261-273); values are resolved when read
    lam_d = self.lambda_depth if self.predict_depth else 0.0
    def gen_gan():
      return self.lambda_gan * sum(float(s[0]) / c for s, c in hinge_sums) / n_dis
    def disc():
      return self.lambda_gan * sum(float(s[1]) / c for s, c in hinge_sums) / n_dis
    def depth_loss():
      return lam_d * float((depth_l1 / torch.clamp(npx, min=1)).mean())
    def wc_loss():
      return self.lambda_wc * float((wc_l1 / 3.0 / torch.clamp(wcm, min=1)).mean())
    self.metrics['dis/disc_loss'].update_state(disc)
    self.metrics['dis/grad_norm'].update_state(d_norm)
    self.metrics['gen/gen_gan_loss'].update_state(gen_gan)
    self.metrics['gen/gen_loss'].update_state(lambda: gen_gan() + depth_loss() + wc_loss())
    self.metrics['gen/depth_loss'].update_state(depth_loss)
    self.metrics['gen/seg_loss'].update_state(0.0)
    self.metrics['gen/depth_seg_loss'].update_state(0.0)
    self.metrics['gen/depth_seg_consistency'].update_state(0.0)
    self.metrics['gen/kld_loss'].update_state(0.0)
    self.metrics['gen/kld_nan'].update_state(0.0)
    self.metrics['gen/wc_loss'].update_state(wc_loss)
    self.metrics['gen/grad_norm'].update_state(g_norm)

  def _update_all(self, opt, fused_sn, ema_theta, ema_omd):
    """One replica: per-tensor clip + Adam (+ EMA) over a model's whole arena -- in one pass
    (AdamState.clip_apply) unless switched off or a test hook wants the clipped arena."""
    nt = len(opt.model.store.trainable_names)
    opt.begin_step()
    if not (FUSED_CLIP_ADAM and opt.clip_apply(0, nt, GRAD_CLIP_NORM, fused_sn, ema_theta, ema_omd)):
      opt.iterations -= 1   # (apply_gradients advances the counter itself)
      opt.clip_gradients(GRAD_CLIP_NORM, fused_sn=fused_sn)
      opt.apply_gradients(None, 1, ema_theta, ema_omd)
    else:
      opt.end_step()

  def _set_input_grad(self, x_all, flag):
    x_all.requires_grad = flag
    for v in getattr(self.discriminator, '_scale_inputs', []):
      v.requires_grad = flag

  # -------------------------------------------------------------------------------- train_d
  def train_d(self, inputs: Dict[str, torch.Tensor]) -> None:
    """One discriminator-only update (reference :275-338)."""
    inputs = self._prep_inputs(inputs)
    image, depth_t = inputs['image'], inputs['depth']
    R = self.strategy.num_replicas_in_sync
    group = self.strategy.group
    G, D = self.generator, self.discriminator
    # generator forward in training mode, outside any tape (:292-293)
    ctx_g = G.make_ctx(training=True, record=False, group=group, world=R)
    ctx_g.binary_masks = bool(getattr(self, 'binary_masks', True))
    outs, _ = G.forward(ctx_g, inputs)
    depth_out, generated = outs[3], outs[6]
    ctx_d = D.make_ctx(training=True, record=True, group=group, world=R)
    x_all = self._disc_input(ctx_d, generated, depth_out, image, depth_t)
    logits = self._run_discriminator(ctx_d, x_all)
    coef_d = self.lambda_gan / (len(logits) * R)
    _, seeds_d, _ = self._hinge(ctx_d, logits, coef_d, 0.0, want_g=False)
    tape_d, ctx_d.tape = ctx_d.tape, None
    ctx_d.param_grads = True
    self._set_input_grad(x_all, False)
    self._backward_tape(ctx_d, tape_d, seeds_d, logits)
    D.spectral.backward_fixup()
    sync = self._grad_sync()
    if sync is None:
      self._update_all(self.d_optimizer, False, None, 0.0)
      self._operand_group(D, None, ctx_d.dtype).prep(D.store.version)
    else:
      self._sync_discriminator(sync)
      sync.finish()
      self.d_optimizer.apply_gradients(group, 1)
