"""CPU: pins oracle/model_np.py (the SE3DSModel restatement) against the reference's own tests of
models/models.py -- models_test.py:38-79 (round trip, shapes, ranges) and :81-137 (plane at 1 m,
run per sample: the constructor refuses batch_size != 1, models.py:95-96)."""
import numpy as np
import pytest
import torch

from oracle import model_np
from se3ds_amd import gin_lite
from se3ds_amd.models import image_models


def _params(size, gen_dims=4, version='50'):
  gin_lite.clear_config()
  G = image_models.ResNetGenerator(image_size=size, gen_dims=gen_dims, z_dim=4,
                                   resnet_version=version, device='cpu', seed=3)
  return {k: v.detach().clone() for k, v in G.store.views.items()}


def test_oracle_model_roundtrip_shapes_ranges():
  # models_test.py:38-79
  size = 64
  rng = np.random.default_rng(0)
  rgb = rng.integers(0, 255, (1, size, 2 * size, 3)).astype(np.uint8)
  seg = rng.integers(0, 42, (1, size, 2 * size, 1)).astype(np.uint8)
  depth = rng.uniform(0, 1, (1, size, 2 * size)).astype(np.float32)
  pos = rng.standard_normal((1, 3)).astype(np.float32)
  m = model_np.SE3DSModelOracle(_params(size), size, 4, '50', z_dim=4)
  m.add_to_memory(rgb, seg, depth, pos, mask_blurred=False)
  out = m(pos)
  assert np.mean(np.all(out['proj_rgb'] == rgb, axis=-1)) >= 0.95
  assert out['proj_semantic'].shape == (1, size, 2 * size)
  assert out['pred_semantic'].shape == (1, size, 2 * size)
  assert out['proj_rgb'].shape == rgb.shape and out['pred_rgb'].shape == rgb.shape
  assert out['pred_rgb'].dtype == np.uint8
  assert out['pred_depth'].shape == depth.shape
  assert out['pred_depth'].min() >= 0 and out['pred_depth'].max() <= 1
  # feedback path (models.py:334-346): predictions join the memory, masked rows seed -1
  m0 = m.rgb.shape[1]
  m(pos + np.float32(0.1), add_preds_to_memory=True)
  assert m.rgb.shape[1] > m0 and m.rgb.dtype == np.int32 and m.feats.dtype == np.uint8
  with pytest.raises(ValueError):
    model_np.SE3DSModelOracle({}, size, 4, batch_size=2)
  with pytest.raises(ValueError):
    m(np.zeros((2, 3), np.float32))


def test_oracle_model_plane_known_answer():
  # models_test.py:81-137, one sample at a time
  image_size = 4
  offset = 0.5 * np.pi / image_size
  heading = np.linspace(-np.pi + offset, np.pi - offset, image_size * 2).astype(np.float32)
  pitch = np.linspace(0.5 * np.pi - offset, -0.5 * np.pi + offset, image_size).astype(np.float32)
  with np.errstate(divide='ignore'):
    depth = (1.0 / np.cos(heading))[None, :] / np.cos(pitch)[:, None]
  depth = np.where(depth > 0, depth, 0).astype(np.float32)
  depth1 = np.roll(depth, image_size // 2, -1)
  rng = np.random.default_rng(1)
  for d, start, (axis, value) in ((depth, [0., 0, 0], (1, 1)), (depth1, [1., 0, 0], (0, 2))):
    rgb = rng.integers(0, 255, (1, image_size, 2 * image_size, 3)).astype(np.uint8)
    seg = rng.integers(1, 42, (1, image_size, 2 * image_size, 1)).astype(np.uint8)
    m = model_np.SE3DSModelOracle({}, image_size, 4)
    m.add_to_memory(rgb, seg, (d / 20.0)[None].astype(np.float32),
                    np.asarray([start], np.float32), mask_blurred=False)
    # half of the 32 pixels see the plane (depth > 0): the compacted memory holds those
    assert m.rgb_coords.shape == (1, 4, image_size ** 2)
    np.testing.assert_allclose(m.rgb_coords[0, axis], image_size ** 2 * [value], rtol=1e-5, atol=1e-5)
