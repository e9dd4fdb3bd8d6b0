"""Device-side half of the reference's datasets/indoor_datasets.py: `augment`
(:34-61) and `R2RImageDataset`'s per-example transform (:263-375) + batch transform (:553-597) as
ONE gather kernel over decoded frames resident in HBM (`se3ds_input_transform`).  TFRecord / PNG
decoding, sharding, shuffling and prefetching stay outside: the caller hands over
uint8 / uint16 frames (what `tf.image.decode_png` yields, :185-228) as CUDA tensors.

Same constructor arguments and gin selectors as the reference (`R2RImageDataset.image_size`, ...).
The random draws follow the statement order of `_transform_fn` and are made on the host with a
NumPy generator (TensorFlow's RNG streams cannot be reproduced without TensorFlow); everything
downstream of the draws is bit-exact against oracle/input_np.py."""
from typing import Dict, List, Optional

import numpy as np
import torch

from se3ds_amd import _lib
from se3ds_amd import constants
from se3ds_amd import gin_lite as gin

F32 = np.float32
RAW_DTYPES = dict(image=torch.uint8, proj_image=torch.uint8, depth=torch.int16, proj_depth=torch.int16,
                  proj_mask=torch.uint8, blurred_mask=torch.uint8, segmentation=torch.uint8)


def draw_augment(rng: np.random.Generator, width: int, random_roll_range: Optional[int] = None,
                 random_flip: bool = True):
  """The draws of `augment` (:54-60): roll amount in [-range, range), flip with p = 0.5."""
  random_roll_range = random_roll_range or (width // 2)
  roll = int(rng.integers(-random_roll_range, random_roll_range))
  flip = bool(rng.uniform() < 0.5) if random_flip else False
  return roll, flip


@gin.configurable
class R2RImageDataset:
  """Preprocessing of R2R / Matterport panoramas for training (reference :64-120)."""

  def __init__(self, image_size: int = 256, preprocessed_image_height: int = 512, z_dim: int = 64,
               num_classes: int = constants.NUM_MP3D_CLASSES, data_dir: str = 'data/train/',
               return_filename: bool = False, horizontal_mask_ratio: float = 0.5,
               vertical_mask_ratio: float = 0.5, random_roll_and_flip: bool = True,
               random_crop: bool = True, random_resize_max: float = 2.0, pad_minval: float = -0.05,
               pad_maxval: float = 0.1, re_10k_crop: bool = False, **kwargs):
    del kwargs
    self.image_size = image_size
    self.preprocessed_image_height = preprocessed_image_height
    self.z_dim = z_dim
    self.num_classes = num_classes
    self.data_dir = data_dir
    self.return_filename = return_filename
    self.horizontal_mask_ratio = horizontal_mask_ratio
    self.vertical_mask_ratio = vertical_mask_ratio
    self.random_roll_and_flip = random_roll_and_flip
    self.random_crop = random_crop
    self.random_resize_max = random_resize_max
    self.pad_minval = pad_minval
    self.pad_maxval = pad_maxval
    self.re_10k_crop = re_10k_crop

  # ------------------------------------------------------------------------------ the draws
  def draw_params(self, rng: np.random.Generator, height: int, width: int) -> dict:
    """One example's random draws, in the statement order of `_transform_fn` (:272-330)."""
    s = self.image_size
    prm = dict(resize=(s, 2 * s), hmask=None, vmask=None, roll=0, flip=False, crop=(0, 0))
    if self.random_crop:
      mult = F32(rng.uniform(1.0, self.random_resize_max))
      prm['resize'] = (int(F32(s) * mult), int(F32(2 * s) * mult))
    if self.horizontal_mask_ratio > 0:
      mask_ratio = F32(rng.uniform(0, self.horizontal_mask_ratio))
      keep_ratio = F32(1) - mask_ratio
      start = F32(rng.uniform(0, width))
      end = F32(np.mod(start + F32(width) * keep_ratio, F32(width)))
      prm['hmask'] = (2 if start > end else 1, float(start), float(end))
    if self.vertical_mask_ratio > 0:
      mask_ratio = F32(rng.uniform(0, self.vertical_mask_ratio))
      image_height = F32(height) * (F32(1) - mask_ratio)
      start = F32(rng.uniform(0, max(float(F32(height) - image_height), 1e-30)))
      prm['vmask'] = (float(start), float(start + image_height))
    if self.random_roll_and_flip:
      prm['roll'], prm['flip'] = draw_augment(rng, prm['resize'][1],
                                              int(float(s) * 2 * self.random_resize_max))
    if self.random_crop:
      rh, rw = prm['resize']
      prm['crop'] = (int(rng.integers(0, rh - s + 1)), int(rng.integers(0, rw - 2 * s + 1)))
    return prm

  # ------------------------------------------------------------------------- device transform
  def device_transform(self, raw: Dict[str, torch.Tensor], params: List[dict]
                       ) -> Dict[str, torch.Tensor]:
    """raw: decoded frames on the GPU -- image / proj_image uint8 (N,H0,W0,3); depth / proj_depth
    uint16 bit patterns held as int16 (N,H0,W0); proj_mask / blurred_mask / segmentation uint8
    (N,H0,W0).  Returns the step's batch dict (image, proj_image, proj_mask, proj_depth, depth,
    blurred_mask fp32 (N,h,w,C); segmentation int32 (N,h,w,1))."""
    for k, dt in RAW_DTYPES.items():
      _lib.require_cuda(raw[k])
      if raw[k].dtype != dt:
        raise ValueError(f'{k}: expected {dt}, got {raw[k].dtype}')
    n, h0, w0 = raw['proj_mask'].shape
    if len(params) != n:
      raise ValueError(f'{len(params)} parameter rows for a batch of {n}')
    s = self.image_size
    ip = np.zeros((n, 8), np.int32)
    fp = np.zeros((n, 4), F32)
    for i, prm in enumerate(params):
      rh, rw = prm['resize']
      oy, ox = prm.get('crop', (0, 0))
      if oy < 0 or ox < 0 or oy + s > rh or ox + 2 * s > rw:
        raise ValueError(f'crop {oy, ox} of a {rh}x{rw} frame does not hold a {s}x{2 * s} panorama')
      ip[i, :6] = (rh, rw, prm.get('roll', 0), int(bool(prm.get('flip', False))), oy, ox)
      if prm.get('hmask') is not None:
        ip[i, 6], fp[i, 0], fp[i, 1] = prm['hmask']
      if prm.get('vmask') is not None:
        ip[i, 7] = 1
        fp[i, 2], fp[i, 3] = prm['vmask']
    dev = raw['proj_mask'].device
    ipd, fpd = torch.from_numpy(ip).to(dev), torch.from_numpy(fp).to(dev)
    f = lambda c: torch.empty((n, s, 2 * s, c), dtype=torch.float32, device=dev)
    out = dict(image=f(3), proj_image=f(3), proj_mask=f(1), proj_depth=f(1), depth=f(1),
               blurred_mask=f(1),
               segmentation=torch.empty((n, s, 2 * s, 1), dtype=torch.int32, device=dev))
    r = {k: raw[k].contiguous() for k in RAW_DTYPES}
    rc = _lib.lib().se3ds_input_transform(
        _lib.ptr(r['image']), _lib.ptr(r['proj_image']), _lib.ptr(r['depth']),
        _lib.ptr(r['proj_depth']), _lib.ptr(r['proj_mask']), _lib.ptr(r['blurred_mask']),
        _lib.ptr(r['segmentation']), _lib.ptr(ipd), _lib.ptr(fpd), n, h0, w0, s, 2 * s,
        _lib.ptr(out['image']), _lib.ptr(out['proj_image']), _lib.ptr(out['proj_mask']),
        _lib.ptr(out['proj_depth']), _lib.ptr(out['depth']), _lib.ptr(out['blurred_mask']),
        _lib.ptr(out['segmentation']), _lib.stream())
    _lib.check(rc, 'se3ds_input_transform')
    return out

  def transform(self, raw: Dict[str, torch.Tensor], rng: np.random.Generator):
    """Draws + device transform of one batch."""
    n, h0, w0 = raw['proj_mask'].shape
    return self.device_transform(raw, [self.draw_params(rng, h0, w0) for _ in range(n)])
