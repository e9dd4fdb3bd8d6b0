"""Interface for using SE3DS models for prediction -- MI355X implementation of the reference's
models/models.py: `SE3DSModel` keeps a point-cloud memory (coords / semantic feats / rgb coords /
rgb), adds equirectangular observations to it (`add_to_memory`, reference :180-245) and renders +
inpaints a panorama at a new position (`__call__`, reference :247-366).  The whole per-call
pipeline (two projections over the memory, mask, generator forward with circular padding,
quantisation, optional feedback into the memory) stays on the device."""
from typing import List, NamedTuple, Optional

import numpy as np
import torch

from se3ds_amd import _lib
from se3ds_amd import constants
from se3ds_amd.models import image_models
from se3ds_amd.utils import pano_utils
from se3ds_amd.utils import point_cloud_utils


def _quantize(x: torch.Tensor, out_dtype, mul=1.0, div=1.0, lo=0.0, hi=0.0, pre=None):
  """libse3ds_hip.so `se3ds_quantize`: Q(clamp?(x) * mul / div), clamped to [lo, hi] (NaN
  propagates like tf.clip_by_value); integer outputs truncate toward zero like tf.cast
  (reference :198,:289-291,:325-331,:353); lo > hi = pure cast without a clamp (int32 -> uint8
  wraps modulo 256 as tf.cast does)."""
  from se3ds_amd import hipops  # noqa: F401  (registers the signature)
  _lib.require_cuda(x)
  x = x.contiguous()
  out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
  _lib.check(_lib.lib().se3ds_quantize(
      x.data_ptr(), _lib.dtype_code(x), x.numel(), 0 if pre is None else 1,
      0.0 if pre is None else float(pre[0]), 0.0 if pre is None else float(pre[1]), float(mul),
      float(div), float(lo), float(hi), out.data_ptr(), _lib.dtype_code(out), _lib.stream()),
      'se3ds_quantize')
  if out_dtype == torch.int32 and lo <= hi:
    # the clamp bounds hold by construction: the 8-byte splat's byte-range promise for this tensor
    # needs no read-back (point_cloud_utils.byte_range)
    from se3ds_amd.utils import point_cloud_utils
    point_cloud_utils.set_int_range(out, int(lo), int(hi))
  return out


class PanoData(NamedTuple):
  position: torch.Tensor
  rgb: torch.Tensor
  semantic: torch.Tensor
  depth: torch.Tensor


class OutputData(NamedTuple):
  """Output tuple (reference :41-74)."""
  proj_semantic: torch.Tensor
  pred_semantic: torch.Tensor
  proj_rgb: torch.Tensor
  pred_rgb: torch.Tensor
  proj_depth: torch.Tensor
  pred_depth: torch.Tensor
  mu: torch.Tensor
  logvar: torch.Tensor
  proj_mask: Optional[torch.Tensor] = None
  heading_enc: Optional[np.ndarray] = None
  pitch_enc: Optional[np.ndarray] = None
  features_enc: Optional[np.ndarray] = None
  predicted_bucket_probs: Optional[torch.Tensor] = None
  predicted_node_xyz: Optional[List[torch.Tensor]] = None


class MemoryState(NamedTuple):
  """coords (N,4,M) fp32; feats (N,M,1) uint8; rgb_coords (N,4,M'); rgb (N,M',3) int32."""
  coords: torch.Tensor
  feats: torch.Tensor
  rgb_coords: torch.Tensor
  rgb: torch.Tensor


class SE3DSModel(object):
  """Interface to use an SE3DS model for predictions (reference :90-366)."""

  def __init__(self, config, device='cuda:0', dtype=torch.float32):
    self.config = config
    if config.batch_size != 1:
      raise ValueError('Several methods do not support batch_size > 1.')
    self.device = torch.device(device)
    self.model = image_models.ResNetGenerator(
        resnet_version=config.resnet_version, gen_dims=config.gen_dims,
        use_blurred_mask=config.use_blurred_mask, device=self.device, dtype=dtype)
    if config.ckpt_path is not None:
      # reference :100-104 restores tf.train.Checkpoint(ema_generator=model): a TensorFlow tensor
      # bundle prefix ('<dir>/ckpt-N', files .index + .data-*) is read by utils/tf_bundle.py
      # through the object-graph key table; the same variables are also accepted as an .npz with
      # keys 'ema_generator/<variable path>' (GANManager.save_checkpoint).
      if not str(config.ckpt_path).endswith('.npz'):
        from se3ds_amd.utils import tf_bundle
        tf_bundle.load_generator(self.model, str(config.ckpt_path), root='ema_generator')
        print(f'Restored SE3DS generator from {config.ckpt_path}.')
        self._restored = True
    if config.ckpt_path is not None and not getattr(self, '_restored', False):
      import numpy as _np
      with _np.load(config.ckpt_path) as f:
        for prefix in ('ema_generator/', 'generator/'):
          sub = {k[len(prefix):]: f[k] for k in f.files if k.startswith(prefix)}
          if sub:
            break
      missing = sorted(set(self.model.store.views) - set(sub))
      if not sub or missing:
        raise KeyError(f'{config.ckpt_path}: generator variables missing, e.g. {missing[:5]}')
      self.model.store.load_dict({k: v for k, v in sub.items() if k in self.model.store.views})
      print(f'Restored SE3DS generator from {config.ckpt_path}.')
      self._restored = True
    if not getattr(self, '_restored', False):
      print('Initializing SE3DS model from scratch.')
    self.prev_rgb_frame = None
    self.batch_size = config.batch_size
    self.height = config.image_height
    self.width = config.image_height * 2
    self.depth_scale = config.depth_scale
    self.reset_memory()

  def _check_batch_size(self, input_batch_size):
    if input_batch_size != self.batch_size:
      raise ValueError('Input batch size is not suitable. Expected '
                       f'{self.batch_size}, got {input_batch_size} instead.')

  def reset_memory(self):
    """Resets memory to empty (reference :127-134)."""
    d = self.device
    self._memory = MemoryState(
        coords=torch.zeros((self.batch_size, 4, 0), device=d),
        feats=torch.zeros((self.batch_size, 0, 1), dtype=torch.uint8, device=d),
        rgb_coords=torch.zeros((self.batch_size, 4, 0), device=d),
        rgb=torch.zeros((self.batch_size, 0, 3), dtype=torch.int32, device=d))

  @staticmethod
  def _clone_state(state):
    out = []
    for t in state:
      c = t.clone()
      point_cloud_utils.propagate_int_range(c, t)   # (a copy keeps values and their known bounds)
      out.append(c)
    return MemoryState(*out)

  def get_memory_state(self) -> MemoryState:
    return self._clone_state(self._memory)

  def set_memory_state(self, state: MemoryState):
    self._memory = self._clone_state(state)

  def write_memory_as_pointcloud(self, filename):
    """Writes memory at batch position 0 to an ASCII .ply file (reference :154-178)."""
    state = self.get_memory_state()
    point_cloud_utils.check_promise(state.rgb_coords.device)   # (a host synchronisation anyway)
    xyz = state.rgb_coords[0, 0:3].cpu().numpy().T
    rgb = state.rgb[0].cpu().numpy()
    with open(filename, 'w') as fp:
      fp.write('ply\nformat ascii 1.0 \n')
      fp.write('element vertex %d\n' % xyz.shape[0])
      fp.write('property float x\nproperty float y\nproperty float z\n')
      fp.write('property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n')
      for i in range(xyz.shape[0]):
        fp.write('{} {} {} {} {} {} \n'.format(xyz[i, 0], xyz[i, 1], xyz[i, 2], rgb[i, 0],
                                               rgb[i, 1], rgb[i, 2]))

  def add_to_memory(self, pano_rgb, pano_semantic, pano_depth, position, mask_blurred=True):
    """Adds an equirectangular observation to the memory (reference :180-245)."""
    self._check_batch_size(pano_semantic.shape[0])
    assert pano_rgb.dtype in [torch.uint8, torch.int32]
    assert pano_semantic.dtype in [torch.uint8, torch.int32]
    _lib.require_cuda(pano_rgb, pano_semantic, pano_depth, position)
    if pano_rgb.dtype != torch.int32:
      pano_rgb = _quantize(pano_rgb, torch.int32, lo=0, hi=255)          # tf.cast(uint8 -> int32)
    if pano_semantic.dtype != torch.uint8:
      pano_semantic = _quantize(pano_semantic, torch.uint8, lo=1, hi=0)  # tf.cast(int32 -> uint8): wraps
    self.prev_rgb_frame = _quantize(pano_rgb, torch.float32, div=255.0, lo=-1.0, hi=1.0)
    if mask_blurred:
      pano_rgb = pano_utils.mask_pano(pano_rgb, masked_region_value=constants.INVALID_RGB_VALUE)
    position = position.to(torch.float32)
    xyz1, feats = pano_utils.equirectangular_to_pointcloud(
        pano_semantic, pano_depth, constants.INVALID_SEM_VALUE, self.depth_scale,
        interpolation_method='nearest', position=position)
    rgb_xyz1, rgb_feats = pano_utils.equirectangular_to_pointcloud(
        pano_rgb, pano_depth, constants.INVALID_RGB_VALUE, self.depth_scale,
        interpolation_method='bilinear', position=position)
    # Filter coords if they are not valid (stream compaction, reference :229-236)
    f_xyz1, f_feats = pano_utils.compact_valid_points(xyz1, feats, constants.INVALID_SEM_VALUE)
    f_rgb_xyz1, f_rgb = pano_utils.compact_valid_points(rgb_xyz1, rgb_feats,
                                                        constants.INVALID_RGB_VALUE)
    f_rgb = f_rgb.to(self._memory.rgb.dtype)
    # (the 'bilinear' unprojection returns fp32 like tf.image.resize; its values are still pano_rgb's
    # integers or the void class, so pano_rgb's known bounds hold for the int32 copy)
    point_cloud_utils.propagate_int_range(f_rgb, pano_rgb, extra=(constants.INVALID_RGB_VALUE,))
    rgb_mem = torch.cat([self._memory.rgb, f_rgb], dim=1)
    # (a concatenation keeps values: the known [-1, 255] bounds of both parts carry over, so the
    # packed splat's byte-range promise for the grown memory needs no device read-back)
    point_cloud_utils.propagate_int_range(rgb_mem, self._memory.rgb, f_rgb)
    self.set_memory_state(MemoryState(
        coords=torch.cat([self._memory.coords, f_xyz1], dim=2),
        feats=torch.cat([self._memory.feats, f_feats], dim=1),
        rgb_coords=torch.cat([self._memory.rgb_coords, f_rgb_xyz1], dim=2),
        rgb=rgb_mem))

  def __call__(self, position, add_preds_to_memory: bool = False, sample_noise: bool = False,
               use_projected_rgb: bool = False, z: Optional[torch.Tensor] = None) -> OutputData:
    """Predicts the frame at `position` (reference :247-366)."""
    batch_size = position.shape[0]
    self._check_batch_size(batch_size)
    _lib.require_cuda(position)
    position = position.to(torch.float32)
    h, w = self.height, self.width
    _, proj_semantic = pano_utils.project_feats_to_equirectangular(
        self._memory.feats, self._memory.coords, h, w, constants.INVALID_SEM_VALUE,
        self.depth_scale, offset=position)
    proj_depth, proj_rgb, proj_mask = pano_utils.project_feats_to_equirectangular(
        self._memory.rgb, self._memory.rgb_coords, h, w, constants.INVALID_RGB_VALUE,
        self.depth_scale, offset=position, with_mask=True,
        mask_void=constants.INVALID_RGB_VALUE)
    proj_mask = proj_mask[..., None]
    proj_semantic = _quantize(proj_semantic[..., 0], torch.uint8, lo=0, hi=255)
    proj_rgb = _quantize(proj_rgb, torch.float32, div=255.0, lo=0.0, hi=1.0)   # clip(x / 255, 0, 1)
    assert self.prev_rgb_frame is not None
    inputs = {
        'prev_image': self.prev_rgb_frame, 'proj_image': proj_rgb,
        'proj_depth': proj_depth[..., None], 'proj_mask': proj_mask,
        'blurred_mask': torch.zeros_like(proj_mask),
        'dataset_type': torch.zeros((batch_size,), dtype=torch.int32, device=self.device),
    }
    (mu, logvar, _, pred_depth, pred_semantic, _, generated_pred_rgb) = self.model(
        inputs=[inputs, None], sample_noise=sample_noise, training=False)
    pred_depth = _quantize(pred_depth[..., 0], torch.float32, lo=0.0, hi=1.0)
    # int32(g * 255) clipped to [-1, 255] (memory RGB, :327-329); int32(clip(g, 0, 1) * 255) (:330-331)
    pc_rgb_tensor = _quantize(generated_pred_rgb, torch.int32, mul=255.0,
                              lo=constants.INVALID_RGB_VALUE, hi=255)
    pred_rgb = _quantize(generated_pred_rgb, torch.int32, mul=255.0, lo=0, hi=255, pre=(0.0, 1.0))
    # tf.argmax over the all-zero segmentation logits (image_models.py:191-193) is class 0
    pred_semantic = torch.zeros(pred_semantic.shape[:-1], dtype=torch.uint8, device=self.device)
    if add_preds_to_memory:
      pred_rgb_mem, pred_semantic_mem, pred_depth_mem = pc_rgb_tensor, pred_semantic, pred_depth
      if use_projected_rgb:
        # reference :339-344 adds a float32 and an int32 tensor, which TensorFlow itself rejects
        # (no implicit promotion): the branch cannot run there, so it is not offered here either
        raise TypeError('use_projected_rgb with add_preds_to_memory adds float32 proj_rgb to int32 '
                        'predictions (models.py:340): TensorFlow raises on the dtype mismatch')
      self.prev_rgb_frame = generated_pred_rgb
      self.add_to_memory(pred_rgb_mem, pred_semantic_mem[..., None], pred_depth_mem, position)
    pred_rgb = _quantize(pred_rgb, torch.uint8, lo=0, hi=255)   # in [0, 255] already: a cast
    # (byte-range promise of the packed splats above: poll without waiting)
    point_cloud_utils.check_promise(pred_rgb.device, wait=False)
    return OutputData(proj_semantic=proj_semantic, pred_semantic=pred_semantic,
                      proj_rgb=_quantize(proj_rgb, torch.uint8, mul=255.0, lo=0, hi=255),
                      pred_rgb=pred_rgb,
                      proj_depth=proj_depth, pred_depth=pred_depth, mu=mu, logvar=logvar,
                      proj_mask=proj_mask)
