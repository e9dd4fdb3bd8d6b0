"""Input pipeline, device-side half: oracle sanity on CPU, HIP kernel bit-exact vs the
oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import input_np
from se3ds_amd.datasets import indoor_datasets

F32 = np.float32


def _raw(rng, n, h0):
  w0 = 2 * h0
  return dict(image=rng.integers(0, 256, (n, h0, w0, 3)).astype(np.uint8),
              proj_image=rng.integers(0, 256, (n, h0, w0, 3)).astype(np.uint8),
              depth=rng.integers(0, 65536, (n, h0, w0)).astype(np.uint16),
              proj_depth=rng.integers(0, 65536, (n, h0, w0)).astype(np.uint16),
              proj_mask=(rng.uniform(size=(n, h0, w0)) < 0.6).astype(np.uint8) * 255,
              blurred_mask=(rng.uniform(size=(n, h0, w0)) < 0.2).astype(np.uint8),
              segmentation=rng.integers(0, 42, (n, h0, w0)).astype(np.uint8))


def test_oracle_identity_and_draw_ranges():
  rng = np.random.default_rng(0)
  raw = _raw(rng, 2, 16)
  ds = indoor_datasets.R2RImageDataset(image_size=16, horizontal_mask_ratio=0, vertical_mask_ratio=0,
                                       random_roll_and_flip=False, random_crop=False)
  prm = [ds.draw_params(rng, 16, 32) for _ in range(2)]
  assert prm[0] == dict(resize=(16, 32), hmask=None, vmask=None, roll=0, flip=False, crop=(0, 0))
  out = input_np.transform_batch(raw, prm, 16)
  f = input_np.convert_frames(raw)
  np.testing.assert_array_equal(out['image'], f['image'])
  np.testing.assert_array_equal(out['proj_mask'][..., 0], f['proj_mask'])
  np.testing.assert_array_equal(out['proj_depth'][..., 0], f['proj_depth'] * f['proj_mask'])
  np.testing.assert_array_equal(out['segmentation'][..., 0], f['segmentation'])
  assert out['image'].dtype == F32 and out['segmentation'].dtype == np.int32
  ds = indoor_datasets.R2RImageDataset(image_size=16)   # reference defaults: everything on
  for _ in range(200):
    p = ds.draw_params(rng, 64, 128)
    rh, rw = p['resize']
    assert 16 <= rh <= 32 and 32 <= rw <= 64 and 0 <= p['crop'][0] <= rh - 16 and 0 <= p['crop'][1] <= rw - 32
    assert -64 <= p['roll'] < 64 and p['hmask'][0] in (1, 2) and 0 <= p['vmask'][0] <= p['vmask'][1] <= 64 + 1e-3
  # a roll by the full width is the identity; a flip twice too
  a = input_np.transform_batch(raw, [dict(resize=(16, 32), roll=32)] * 2, 16)
  np.testing.assert_array_equal(a['image'], out['image'])


@pytest.mark.gpu
@pytest.mark.parametrize('h0,size', [(64, 32), (128, 128), (96, 64)])
def test_device_transform_bit_exact(h0, size):
  rng = np.random.default_rng(10 + h0)
  n = 3
  raw = _raw(rng, n, h0)
  ds = indoor_datasets.R2RImageDataset(image_size=size, preprocessed_image_height=h0)
  params = [ds.draw_params(rng, h0, 2 * h0) for _ in range(n)]
  params[0]['flip'], params[1]['flip'] = True, False       # both branches
  params[2] = dict(resize=(size, 2 * size), hmask=(2, 2 * h0 * 0.7, 2 * h0 * 0.2), vmask=None,
                   roll=-5, flip=True, crop=(0, 0))         # no resize jitter, wrapped band
  ref = input_np.transform_batch(raw, params, size)
  dev = 'cuda:0'
  t = {k: torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).to(dev) for k, v in raw.items()}
  out = ds.device_transform(t, params)
  for k in ('image', 'proj_image', 'proj_mask', 'proj_depth', 'depth', 'blurred_mask', 'segmentation'):
    np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=k)
  assert out['image'].dtype == torch.float32 and out['segmentation'].dtype == torch.int32
  # the step consumes it as is
  assert float(out['proj_image'].max()) <= 1.0 and set(np.unique(out['proj_mask'].cpu().numpy())) <= {0.0, 1.0}
  with pytest.raises(ValueError):
    ds.device_transform(t, params[:1])
