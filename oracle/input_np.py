"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the reference's
per-example input transform and batch transform: datasets/indoor_datasets.py `augment` (:34-61),
the int -> float conversions of `_parse` (:185-228), `_transform_fn` (:263-375) and
`_train_batch_transform_fn` (:553-597), with the random draws passed in explicitly (the draws are
TF RNG streams, unpinnable here; everything downstream of them is deterministic).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: "parity unpinned" -- the reference has no test of this transform; tf.image.resize
follows the published kernels (half-pixel centres; bilinear = two lerps in fp32; nearest =
floor((i + 0.5) * scale)), restated in oracle/warp_np.py.
"""
import numpy as np

from oracle import warp_np

F32 = np.float32


def convert_frames(raw):
  """:185-228: tf.image.convert_image_dtype (x * (1 / dtype.max) in fp32), mask clipping."""
  s8, s16 = F32(1.0 / 255.0), F32(1.0 / 65535.0)
  return dict(
      image=(raw['image'].astype(F32) * s8).astype(F32),
      proj_image=(raw['proj_image'].astype(F32) * s8).astype(F32),
      depth=(raw['depth'].astype(F32) * s16).astype(F32),
      proj_depth=(raw['proj_depth'].astype(F32) * s16).astype(F32),
      proj_mask=np.clip(raw['proj_mask'], 0, 1).astype(F32),
      blurred_mask=np.clip(raw['blurred_mask'], 0, 1).astype(F32),
      segmentation=raw['segmentation'].astype(np.int32))


def transform_example(f, prm, image_size):
  """`_transform_fn` on one converted example `f` (H0,W0[,C]) with explicit draws `prm`:
  resize (rh, rw), hmask (mode, start, end) | None, vmask (start, end) | None, roll, flip,
  crop (oy, ox)."""
  h0, w0 = f['proj_mask'].shape
  proj_mask = f['proj_mask'][..., None]
  if prm.get('hmask') is not None:
    mode, start, end = prm['hmask']
    r = np.arange(w0, dtype=F32)
    m = ((r > F32(start)) | (r < F32(end))) if mode == 2 else ((r > F32(start)) & (r < F32(end)))
    proj_mask = proj_mask * m[None, :, None].astype(F32)
  if prm.get('vmask') is not None:
    start, end = prm['vmask']
    r = np.arange(h0, dtype=F32)
    m = (r > F32(start)) & (r < F32(end))
    proj_mask = proj_mask * m[:, None, None].astype(F32)
  rh, rw = prm['resize']
  semantics = np.concatenate([f['segmentation'][..., None].astype(F32), f['depth'][..., None],
                              f['proj_depth'][..., None], proj_mask, f['blurred_mask'][..., None],
                              f['proj_image']], axis=-1)
  images = np.clip(warp_np._resize_bilinear(f['image'][None], rh, rw)[0], 0.0, 1.0).astype(F32)
  semantics = warp_np._resize_nearest(semantics[None], rh, rw)[0]
  aug = np.concatenate([images, semantics], axis=-1)
  aug = np.roll(aug, prm.get('roll', 0), axis=1)       # augment(): tf.roll(x, amount, axis=2)
  if prm.get('flip', False):
    aug = aug[:, ::-1]
  oy, ox = prm.get('crop', (0, 0))
  aug = aug[oy:oy + image_size, ox:ox + 2 * image_size]
  assert aug.shape[:2] == (image_size, 2 * image_size)
  return dict(image=aug[..., 0:3], segmentation=aug[..., 3:4].astype(np.int32), depth=aug[..., 4:5],
              proj_depth=aug[..., 5:6], proj_mask=aug[..., 6:7], blurred_mask=aug[..., 7:8],
              proj_image=aug[..., 8:11])


def transform_batch(raw, params, image_size):
  """Per-example transform, batching, then `_train_batch_transform_fn` (:577-585)."""
  f = convert_frames(raw)
  outs = [transform_example({k: v[i] for k, v in f.items()}, params[i], image_size)
          for i in range(len(params))]
  b = {k: np.stack([o[k] for o in outs]) for k in outs[0]}
  b['proj_image'] = (b['proj_image'] * b['proj_mask']).astype(F32)
  b['proj_depth'] = (b['proj_depth'] * b['proj_mask']).astype(F32)
  return b
