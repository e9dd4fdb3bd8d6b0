"""GAN training manager -- MI355X implementation of the hot-path half of the reference's
trainers/gan_manager.py: model / optimizer construction (:169-183), the cluster step
(:351-385), the EMA hooks (:642-655) and checkpoint save / restore as one .npz.  Dataset
pipelines, TensorFlow checkpoint bundles, TensorBoard
logging and the FID evaluation loop of the reference are out of scope;
`train()` runs the same host loop on a synthetic (or user supplied) batch iterator."""
import abc
import os
import random
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from se3ds_amd import _lib
from se3ds_amd import gin_lite as gin
from se3ds_amd import hipops  # noqa: F401
from se3ds_amd.trainers import dist_utils
from se3ds_amd.utils import ema


class OneDeviceStrategy:
  """Stands in for tf.distribute.OneDeviceStrategy (reference main.py:55-60)."""
  num_replicas_in_sync = 1
  group = None

  def __init__(self, device='cuda:0'):
    self.device = torch.device(device)


class DataParallelStrategy:
  """One process per GPU over RCCL (torch.distributed 'nccl'); stands in for
  tf.distribute.MirroredStrategy (reference main.py:63)."""

  def __init__(self, device, group=None):
    self.device = torch.device(device)
    self.group = group
    self.num_replicas_in_sync = dist.get_world_size(group)


class AdamState:
  """Keras Adam slots over a model's flat trainable arena (reference :175-183)."""

  def __init__(self, model, lr, beta_1, beta_2, epsilon=1e-7):
    self.model = model
    self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
    self.m = torch.zeros_like(model.store.theta)
    self.v = torch.zeros_like(model.store.theta)
    self.iterations = 0
    self.chunks, self.tensor_chunk_start = model.store.chunk_tables()
    self._tcs_host = self.tensor_chunk_start.cpu().tolist()
    dev = model.store.theta.device
    self.partial = torch.empty(self.chunks.shape[0], dtype=torch.float32, device=dev)
    self.sqnorm = torch.empty(len(model.store.trainable_names), dtype=torch.float32, device=dev)
    self.mean_norm = torch.zeros(1, dtype=torch.float32, device=dev)
    self.on_update = None   # tests: f(e0, e1), called right before Adam consumes grad[e0:e1]

  def _sn_ptr(self, fused_sn):
    """fused_sn: the model's spectral layers only ran SpectralGroup.backward_fixup(dots_only=True);
    their fix-up is folded into this clip pass."""
    if not fused_sn:
      return None
    sn = self.model.spectral.tensor_sn(self.model.store)
    return sn.data_ptr() if sn is not None else None

  def clip_gradients(self, clip_norm=5.0, fused_sn=False):
    """Per-tensor tf.clip_by_norm, in place on the gradient arena (se3ds_trainer.py:27-32)."""
    st = self.model.store
    L = _lib.lib()
    nt = len(st.trainable_names)
    sn = self._sn_ptr(fused_sn)
    _lib.check(L.se3ds_multi_sqnorm_sn(st.grad.data_ptr(), self.chunks.data_ptr(),
                                       self.chunks.shape[0], self.tensor_chunk_start.data_ptr(), nt,
                                       self.partial.data_ptr(), self.sqnorm.data_ptr(), sn, 0,
                                       _lib.stream()), 'se3ds_multi_sqnorm_sn')
    _lib.check(L.se3ds_multi_clip_by_norm_sn(st.grad.data_ptr(), self.chunks.data_ptr(),
                                             self.chunks.shape[0], self.sqnorm.data_ptr(), nt,
                                             float(clip_norm), self.mean_norm.data_ptr(), sn,
                                             _lib.stream()), 'se3ds_multi_clip_by_norm_sn')
    return self.mean_norm

  def clip_segment(self, t0, t1, clip_norm=5.0, fused_sn=False):
    """clip_gradients restricted to tensors [t0, t1) (per-segment gradient synchronisation)."""
    st = self.model.store
    L = _lib.lib()
    sn = self._sn_ptr(fused_sn)
    c0, c1 = self._tcs_host[t0], self._tcs_host[t1]
    if c1 <= c0:
      return
    cache = self.__dict__.setdefault('_seg_tcs', {})
    if (t0, t1) not in cache:   # chunk prefix of the segment, relative to its first chunk
      cache[(t0, t1)] = (self.tensor_chunk_start[t0:t1 + 1] - c0).contiguous()
    tcs = cache[(t0, t1)]
    _lib.check(L.se3ds_multi_sqnorm_sn(st.grad.data_ptr(), self.chunks.data_ptr() + 24 * c0, c1 - c0,
                                       tcs.data_ptr(), t1 - t0, self.partial.data_ptr(),
                                       self.sqnorm.data_ptr() + 4 * t0, sn, t0, _lib.stream()),
               'se3ds_multi_sqnorm_sn')
    # chunk rows carry absolute tensor ids: sqnorm is passed from its base
    _lib.check(L.se3ds_multi_clip_by_norm_sn(st.grad.data_ptr(), self.chunks.data_ptr() + 24 * c0,
                                             c1 - c0, self.sqnorm.data_ptr(), t1 - t0,
                                             float(clip_norm), None, sn, _lib.stream()),
               'se3ds_multi_clip_by_norm_sn')

  def clip_apply(self, t0, t1, clip_norm=5.0, fused_sn=False, ema_theta=None, one_minus_decay=0.0):
    """One replica: clip_segment(t0, t1) + apply_segment over the same tensors in ONE pass over the
    gradient arena (se3ds_multi_clip_adam_keras_ema): the clipped gradient only ever exists in
    registers.  Bit-identical to the two calls.  Returns False (nothing done) when a test hook
    wants to see the clipped arena (on_update) -- the caller then takes the separate passes.
    Call begin_step() first, end_step() last, as for apply_segment."""
    if self.on_update is not None:
      return False
    st = self.model.store
    L = _lib.lib()
    sn = self._sn_ptr(fused_sn)
    c0, c1 = self._tcs_host[t0], self._tcs_host[t1]
    if c1 <= c0:
      return True
    cache = self.__dict__.setdefault('_seg_tcs', {})
    if (t0, t1) not in cache:
      cache[(t0, t1)] = (self.tensor_chunk_start[t0:t1 + 1] - c0).contiguous()
    tcs = cache[(t0, t1)]
    _lib.check(L.se3ds_multi_sqnorm_sn(st.grad.data_ptr(), self.chunks.data_ptr() + 24 * c0, c1 - c0,
                                       tcs.data_ptr(), t1 - t0, self.partial.data_ptr(),
                                       self.sqnorm.data_ptr() + 4 * t0, sn, t0, _lib.stream()),
               'se3ds_multi_sqnorm_sn')
    _lib.check(L.se3ds_multi_clip_adam_keras_ema(
        st.theta.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
        self.chunks.data_ptr() + 24 * c0, c1 - c0, self.sqnorm.data_ptr(), float(clip_norm), sn,
        self.lr, self.beta_1, self.beta_2, self.epsilon, self.iterations, _lib.ptr(ema_theta),
        float(one_minus_decay), _lib.stream()), 'se3ds_multi_clip_adam_keras_ema')
    return True

  def mean_clipped_norm(self, clip_norm=5.0):
    _lib.check(_lib.lib().se3ds_mean_clipped_norm(
        self.sqnorm.data_ptr(), len(self.model.store.trainable_names), float(clip_norm),
        self.mean_norm.data_ptr(), _lib.stream()), 'se3ds_mean_clipped_norm')
    return self.mean_norm

  def begin_step(self):
    """Per-segment updates (apply_segment): the Adam step counter advances once per step."""
    self.iterations += 1

  def apply_segment(self, e0, e1, ema_theta=None, one_minus_decay=0.0):
    """The Keras Adam update (+ the EMA of the same variables) on elements [e0, e1) of the arena:
    a module is updated as soon as the backward pass has left it (se3ds_trainer.train_g_d), on a
    side stream, while the rest of the backward pass still runs.  Element-wise, so the result is
    bit-identical to one pass over the whole arena.  Call begin_step() first, end_step() last."""
    st = self.model.store
    if e1 <= e0:
      return
    if self.on_update is not None:
      self.on_update(e0, e1)
    _lib.check(_lib.lib().se3ds_multi_adam_keras_ema(
        st.theta.data_ptr() + 4 * e0, st.grad.data_ptr() + 4 * e0, self.m.data_ptr() + 4 * e0,
        self.v.data_ptr() + 4 * e0, e1 - e0, self.lr, self.beta_1, self.beta_2, self.epsilon,
        self.iterations, (ema_theta.data_ptr() + 4 * e0) if ema_theta is not None else None,
        float(one_minus_decay), _lib.stream()), 'se3ds_multi_adam_keras_ema')

  def end_step(self):
    self.model.store.version += 1

  def apply_gradients(self, group=None, world=1, ema_theta=None, one_minus_decay=0.0):
    """Cross-replica SUM of the (already clipped, already 1/R-scaled) gradients, then the
    Keras Adam update (reference se3ds_trainer.py:253-257).  With `ema_theta` (the EMA model's
    trainable arena) the moving average of these variables is advanced in the same pass."""
    st = self.model.store
    if world > 1:
      dist_utils.allreduce_arena_sum(st.grad, group)
    if self.on_update is not None:
      self.on_update(0, st.theta.numel())
    self.iterations += 1
    _lib.check(_lib.lib().se3ds_multi_adam_keras_ema(
        st.theta.data_ptr(), st.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
        st.theta.numel(), self.lr, self.beta_1, self.beta_2, self.epsilon, self.iterations,
        _lib.ptr(ema_theta), float(one_minus_decay), _lib.stream()), 'se3ds_multi_adam_keras_ema')
    st.version += 1


class Mean:
  """tf.keras.metrics.Mean over device scalars; values are read lazily (no sync per step)."""

  def __init__(self, name):
    self.name = name
    self._vals = []

  def update_state(self, v):
    self._vals.append(v)

  def reset_states(self):
    self._vals = []

  def result(self):
    if not self._vals:
      return np.float32(0.0)
    tot = 0.0
    for v in self._vals:
      if callable(v):
        v = v()
      if isinstance(v, torch.Tensor):
        v = float(v.detach().float().mean().cpu())
      tot += float(v)
    return np.float32(tot / len(self._vals))


@gin.configurable(denylist=['strategy', 'model_dir'])
class GANManager(abc.ABC):
  """Constructor surface of the reference (:98-167)."""

  def __init__(self, strategy, model_dir: str = '', image_size: int = 128, seed: int = 1,
               optimizer_type: str = 'adam', beta1: float = 0.0, beta2: float = 0.999,
               g_lr: float = 0.0002, d_lr: float = 0.0002, train_batch_size: int = 128,
               test_batch_size: int = 128, parallel_calls: int = -1, log_every_steps: int = 1000,
               save_every_steps: int = 2000, eval_every_steps: int = 2000, num_epochs: int = 100,
               d_step_per_g_step: int = 1, num_batched_steps: int = 5, show_num: int = 16,
               shuffle_buffer_size: int = 1000, ema_decay: float = 0.999, ema_init_step: int = 0,
               generator_fn=None, discriminator_fn=None, train_dataset_glob: Optional[str] = None,
               test_dataset_glob: Optional[str] = None, eval_size: Optional[int] = 10000,
               test_split: str = 'val_seen', eval_seq_len: int = 4, predict_depth: bool = False,
               compute_dtype=torch.float32):
    self.strategy = strategy
    self.model_dir = model_dir
    self.image_size = image_size
    self.seed = seed
    self.optimizer_type = optimizer_type
    self.beta1, self.beta2 = beta1, beta2
    self.g_lr, self.d_lr = g_lr, d_lr
    self.global_batch_size = train_batch_size
    self.train_batch_size = train_batch_size
    self.test_batch_size = test_batch_size
    self.parallel_calls = parallel_calls
    self.log_every_steps = log_every_steps
    self.save_every_steps = save_every_steps
    self.eval_every_steps = eval_every_steps
    self.num_epochs = num_epochs
    self.d_step_per_g_step = d_step_per_g_step
    self.num_batched_steps = num_batched_steps
    self.show_num = show_num
    self.shuffle_buffer_size = shuffle_buffer_size
    self.ema_decay = ema_decay
    self.ema_init_step = ema_init_step
    self.generator_fn = generator_fn
    self.discriminator_fn = discriminator_fn
    self.train_dataset_glob = train_dataset_glob
    self.test_dataset_glob = test_dataset_glob
    self.eval_size = eval_size
    self.test_split = test_split
    self.eval_seq_len = eval_seq_len
    self.predict_depth = predict_depth
    self.compute_dtype = compute_dtype
    self.global_step = 0
    self.train_ds = None
    if seed > 0:   # reference :164-167 (the shipped configs set seed = 0: nothing is seeded)
      random.seed(seed)
      np.random.seed(seed)
      torch.manual_seed(seed)

  # ----------------------------------------------------------------------------- building
  def _build_model(self):
    """Creates Generator, Discriminator and EMA Generator (reference :169-173)."""
    dev = self.strategy.device
    kw = dict(device=dev, dtype=self.compute_dtype)
    sgn = -1 if getattr(self, 'device_init', False) else 1   # negative seed: on-device init
    self.generator = self.generator_fn(image_size=self.image_size,
                                       seed=sgn * (1000 + max(self.seed, 0)), **kw)
    self.discriminator = self.discriminator_fn(image_size=self.image_size,
                                               seed=sgn * (2000 + max(self.seed, 0)), **kw)
    self.ema_generator = self.generator_fn(image_size=self.image_size,
                                           seed=sgn * (3000 + max(self.seed, 0)), **kw)

  def _build_optimizer(self):
    """Creates optimizers for both Generator and Discriminator (reference :175-183)."""
    if self.optimizer_type == 'adam':
      self.g_optimizer = AdamState(self.generator, self.g_lr, self.beta1, self.beta2)
      self.d_optimizer = AdamState(self.discriminator, self.d_lr, self.beta1, self.beta2)
    else:
      raise NotImplementedError

  def _create_metrics(self):
    self.metrics = {'gen_loss': Mean('gen_loss'), 'disc_loss': Mean('disc_loss')}

  def _create_obj(self):
    """reference :333-349 without the checkpoint objects."""
    self.global_step = 0
    self._build_model()
    self._build_optimizer()
    self._create_metrics()

  def _split_input_dict(self, input_dict, splits):
    """Splits a batch dict along axis 0 into `splits` dicts (reference :351-364)."""
    output = [dict() for _ in range(splits)]
    for key, item in input_dict.items():
      n = item.shape[0]
      if n % splits != 0:
        raise ValueError(f'batch {n} of {key} is not divisible by {splits}')
      for i, part in enumerate(torch.split(item, n // splits)):
        output[i][key] = part
    return output

  @abc.abstractmethod
  def train_d(self, inputs):
    """Learns the Discriminator updates."""

  @abc.abstractmethod
  def train_g_d(self, inputs):
    """Learns both the Generator and Discriminator updates."""

  def train_cluster(self, steps=1):
    """`steps` cluster steps: d_step_per_g_step-1 train_d calls then one train_g_d, each on
    its own chunk of the cluster batch (reference :376-385)."""
    for _ in range(int(steps)):
      inputs = next(self.train_ds)
      input_list = self._split_input_dict(inputs, self.d_step_per_g_step)
      for i in range(self.d_step_per_g_step - 1):
        self.train_d(input_list[i])
      self.train_g_d(input_list[-1])

  def train(self, train_ds=None, num_train_steps=None):
    """Host loop of the reference (:387-423) minus logging / checkpoint files."""
    self.global_batch_size = self.train_batch_size
    if not hasattr(self, 'generator'):
      self._create_obj()
    if train_ds is not None:
      self.train_ds = iter(train_ds)
    if self.train_ds is None:
      raise ValueError('no training data: pass train_ds (an iterator of batch dicts)')
    if num_train_steps is None:
      num_train_steps = 1 if self.num_epochs == -1 else self.num_batched_steps
    for step in range(self.global_step, num_train_steps, self.num_batched_steps):
      self.train_cluster(self.num_batched_steps)
      if step % self.log_every_steps < self.num_batched_steps:
        self._save_metrics_to_dict()
        self._reset_metrics()
      self.global_step += self.num_batched_steps

  def _reset_metrics(self):
    for key in self.metrics:
      self.metrics[key].reset_states()

  def _save_metrics_to_dict(self):
    output_dict = {}
    for key, value in self.metrics.items():
      r = value.result()
      if np.any(np.isnan(r)):
        raise ValueError(f'NaN losses recorded for {key}.')
      output_dict[key] = r
    return output_dict

  # ----------------------------------------------------------------------- evaluation loop
  def _get_image_grid(self, inputs, modes=('normal', 'ema')):
    """Video branch of the reference's `_get_image_grid` (:458-541): for the training and the EMA
    generator, roll a trajectory out autoregressively -- project the memory, generate, feed the
    frame back (utils/eval_metric.generated_rollout; unprojection with void_class 0 as :536-539
    does).  inputs: image (N,T,H,W,3), depth (N,T,H,W,1), position (N,T,3), depth_scale (N,).
    Returns {mode: RolloutOutput}; composing / writing the TensorBoard grid stays out of scope."""
    from se3ds_amd.utils import eval_metric
    res = {}
    for mode in modes:
      gen = self.ema_generator if mode == 'ema' else self.generator
      res[mode] = eval_metric.generated_rollout(gen, inputs, self.eval_seq_len,
                                                predict_depth=self.predict_depth,
                                                unproject_void_class=0)
    return res

  # -------------------------------------------------------------------------- checkpoint
  # The reference keeps tf.train.Checkpoint(generator, discriminator, ema_generator, g_optimizer,
  # d_optimizer) (:333-349).  TensorFlow's bundle format cannot be read or written here; the same
  # state goes to one .npz whose keys are '<object>/<variable path>' with the variable paths of the
  # ParamStore, which mirror the reference's attribute paths (INTEGRATION.md section 5 has the
  # TF-side exporter that produces a compatible file from a real checkpoint).
  def state_dict(self):
    """Flat {key: numpy array} of everything a resumed run needs."""
    out = {'global_step': np.asarray(self.global_step, np.int64)}
    for prefix, model in (('generator', self.generator), ('discriminator', self.discriminator),
                          ('ema_generator', self.ema_generator)):
      for k, v in model.store.to_dict().items():
        out[f'{prefix}/{k}'] = v
    for prefix, opt in (('g_optimizer', self.g_optimizer), ('d_optimizer', self.d_optimizer)):
      out[f'{prefix}/iter'] = np.asarray(opt.iterations, np.int64)
      st = opt.model.store
      m, v = opt.m.detach().cpu().numpy(), opt.v.detach().cpu().numpy()
      for name in st.trainable_names:
        o, n, shape = st._off_tr[name]
        out[f'{prefix}/{name}/m'] = m[o:o + n].reshape(shape).copy()
        out[f'{prefix}/{name}/v'] = v[o:o + n].reshape(shape).copy()
    return out

  def load_state_dict(self, d, strict=True):
    if not hasattr(self, 'generator'):
      self._create_obj()
    seen = set()
    for prefix, model in (('generator', self.generator), ('discriminator', self.discriminator),
                          ('ema_generator', self.ema_generator)):
      sub = {k[len(prefix) + 1:]: d[k] for k in d if k.startswith(prefix + '/')}
      known = {k: v for k, v in sub.items() if k in model.store.views}
      if strict and set(known) != set(model.store.views):
        missing = sorted(set(model.store.views) - set(known))[:5]
        raise KeyError(f'checkpoint lacks {prefix} variables, e.g. {missing}')
      model.store.load_dict(known)
      seen.update(f'{prefix}/{k}' for k in known)
    for prefix, opt in (('g_optimizer', self.g_optimizer), ('d_optimizer', self.d_optimizer)):
      if f'{prefix}/iter' not in d:
        if strict:
          raise KeyError(f'checkpoint lacks {prefix}/iter')
        continue
      opt.iterations = int(d[f'{prefix}/iter'])
      st = opt.model.store
      for name in st.trainable_names:
        o, n, shape = st._off_tr[name]
        for slot, arena in (('m', opt.m), ('v', opt.v)):
          key = f'{prefix}/{name}/{slot}'
          if key in d:
            arena[o:o + n].copy_(torch.as_tensor(np.asarray(d[key]), dtype=torch.float32)
                                 .reshape(-1).to(arena.device))
            seen.add(key)
          elif strict:
            raise KeyError(f'checkpoint lacks {key}')
    if 'global_step' in d:
      self.global_step = int(d['global_step'])
    return sorted(set(d) - seen - {'global_step', 'g_optimizer/iter', 'd_optimizer/iter'})

  def save_checkpoint(self, path):
    """One uncompressed .npz (reference: checkpoint_manager.save, :425-428)."""
    np.savez(path, **self.state_dict())

  def restore_checkpoint(self, path, strict=True):
    """Returns the keys of the file that matched nothing (reference: checkpoint.restore)."""
    with np.load(path) as f:
      return self.load_state_dict({k: f[k] for k in f.files}, strict=strict)

  # ---------------------------------------------------------------------------------- EMA
  def ema_fused_args(self):
    """(ema arena, 1 - decay) for AdamState.apply_gradients when this step's update_ema_model
    will be a moving-average step (not the copy phase), else (None, 0)."""
    if (os.environ.get('SE3DS_UNFUSED_EMA') != '1' and
        self.global_step >= self.ema_init_step + self.num_batched_steps):
      return self.ema_generator.store.theta, 1.0 - self.ema_decay
    return None, 0.0

  def update_ema_model(self, theta_done=False):
    """reference :642-651.  theta_done: the trainable variables were already averaged by the
    optimiser pass (ema_fused_args); only the non-trainable ones are left."""
    if self.global_step >= self.ema_init_step:
      if self.global_step >= self.ema_init_step + self.num_batched_steps:
        ema.update_ema_variables(self.ema_generator, self.generator, self.ema_decay,
                                 skip_trainable=theta_done)
      else:
        assert not theta_done
        self.assign_ema_model_first_time()

  def assign_ema_model_first_time(self):
    ema.assign_ema_vars_from_initial_values(self.ema_generator, self.generator)
