"""CPU ORACLE (test infrastructure, NOT product code) -- restatement of the reference's
inference wrapper models/models.py `SE3DSModel` (:90-366): point-cloud memory, `add_to_memory`
(:180-245) and `__call__` (:247-366), on top of the NumPy / C warp oracle (oracle/warp_np.py,
oracle/warp_c.py) and the PyTorch-CPU generator oracle (oracle/nets_torch.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: pinned by the reference's own tests as far as they go -- models_test.py:38-79
(round trip >= 95 %, shapes, ranges) and :96-137 (plane at 1 m) are asserted on this class in
tests/test_oracle_model.py.  The quantisation steps follow the statement order of :325-345;
TensorFlow cannot run here, so cast semantics are the published ones (float -> int32 truncates
toward zero; int / int true-divides in float64 and is then cast).
"""
import numpy as np
import torch

from oracle import nets_torch as O
from oracle import warp_c
from oracle import warp_np

F32 = np.float32
INVALID_SEM_VALUE, INVALID_RGB_VALUE = 0, -1   # constants.py:21-22


class SE3DSModelOracle:
  """`params`: generator variables (name -> torch tensor) in the product's naming."""

  def __init__(self, params, image_height, gen_dims, resnet_version='101', depth_scale=20.0,
               use_blurred_mask=True, z_dim=128, batch_size=1):
    if batch_size != 1:
      raise ValueError('Several methods do not support batch_size > 1.')   # :95-96
    self.p = params
    self.height, self.width = image_height, 2 * image_height
    self.depth_scale = depth_scale
    self.gen = dict(gen_dims=gen_dims, resnet_version=resnet_version, context_layer='convs',
                    z_dim=z_dim, use_blurred_mask=use_blurred_mask)
    self.batch_size = batch_size
    self.prev_rgb_frame = None
    self.reset_memory()

  def reset_memory(self):   # :127-134
    n = self.batch_size
    self.coords = np.zeros((n, 4, 0), F32)
    self.feats = np.zeros((n, 0, 1), np.uint8)
    self.rgb_coords = np.zeros((n, 4, 0), F32)
    self.rgb = np.zeros((n, 0, 3), np.int32)

  def _check_batch_size(self, b):
    if b != self.batch_size:
      raise ValueError(f'Input batch size is not suitable. Expected {self.batch_size}, got {b} instead.')

  def add_to_memory(self, pano_rgb, pano_semantic, pano_depth, position, mask_blurred=True):
    """:180-245."""
    self._check_batch_size(pano_semantic.shape[0])
    assert pano_rgb.dtype in (np.uint8, np.int32) and pano_semantic.dtype in (np.uint8, np.int32)
    pano_rgb = pano_rgb.astype(np.int32)
    pano_semantic = pano_semantic.astype(np.uint8)
    self.prev_rgb_frame = (pano_rgb / 255).astype(F32)   # int / int: float64 true division, cast
    if mask_blurred:
      pano_rgb = warp_np.mask_pano(pano_rgb, masked_region_value=INVALID_RGB_VALUE)
    position = np.asarray(position, F32)
    xyz1, feats = warp_np.equirectangular_to_pointcloud(pano_semantic, pano_depth, INVALID_SEM_VALUE,
                                                        self.depth_scale)
    rgb_xyz1, rgb_feats = warp_np.equirectangular_to_pointcloud(
        pano_rgb, pano_depth, INVALID_RGB_VALUE, self.depth_scale, interpolation_method='bilinear')
    t = np.concatenate([position, np.zeros((self.batch_size, 1), F32)], axis=1)   # (x, y, z, 0)
    xyz1 = (xyz1 + t[:, :, None]).astype(F32)
    rgb_xyz1 = (rgb_xyz1 + t[:, :, None]).astype(F32)
    f_xyz1, f_feats = warp_np.compact_valid(xyz1, feats, INVALID_SEM_VALUE)
    f_rgb_xyz1, f_rgb = warp_np.compact_valid(rgb_xyz1, rgb_feats, INVALID_RGB_VALUE)
    self.coords = np.concatenate([self.coords, f_xyz1], axis=2)
    self.feats = np.concatenate([self.feats, f_feats.astype(np.uint8)], axis=1)
    self.rgb_coords = np.concatenate([self.rgb_coords, f_rgb_xyz1], axis=2)
    self.rgb = np.concatenate([self.rgb, f_rgb.astype(np.int32)], axis=1)

  def __call__(self, position, add_preds_to_memory=False, sample_noise=False,
               use_projected_rgb=False):
    """:247-366.  Returns a dict with the OutputData fields."""
    self._check_batch_size(position.shape[0])
    position = np.asarray(position, F32)
    h, w = self.height, self.width
    # memory - (x, y, z, 0): the C oracle subtracts `offset` from rows 0..2 (one fp32 rounding)
    _, proj_semantic = warp_c.project_feats_to_equirectangular(
        self.feats, self.coords, h, w, INVALID_SEM_VALUE, self.depth_scale, offset=position)
    proj_depth, proj_rgb = warp_c.project_feats_to_equirectangular(
        self.rgb, self.rgb_coords, h, w, INVALID_RGB_VALUE, self.depth_scale, offset=position)
    proj_mask = warp_np.proj_mask(proj_depth, proj_rgb, INVALID_RGB_VALUE)
    proj_semantic = proj_semantic[..., 0].astype(np.uint8)
    proj_rgb = np.clip((proj_rgb / F32(255)).astype(F32), 0, 1)
    assert self.prev_rgb_frame is not None
    cond = {'proj_image': torch.from_numpy(proj_rgb),
            'proj_depth': torch.from_numpy(proj_depth[..., None].copy()),
            'proj_mask': torch.from_numpy(proj_mask),
            'blurred_mask': torch.zeros(proj_mask.shape)}
    if sample_noise:
      raise ValueError('This model does not support noise sampling!')
    with torch.no_grad():
      outs, _ = O.generator_forward(self.p, cond, False, **self.gen)
    mu, logvar, generated = outs[0].numpy(), outs[1].numpy(), outs[6].numpy()
    pred_depth = np.clip(outs[3].numpy()[..., 0], 0, 1)
    pc_rgb = np.clip(np.trunc(generated * F32(255)).astype(np.int32), INVALID_RGB_VALUE, 255)
    pred_rgb = np.trunc(np.clip(generated, 0, 1) * F32(255)).astype(np.int32)
    pred_semantic = np.argmax(outs[4].numpy(), axis=-1).astype(np.uint8)
    if add_preds_to_memory:
      pred_rgb_mem, pred_semantic_mem, pred_depth_mem = pc_rgb, pred_semantic, pred_depth
      if use_projected_rgb:
        # :339-344: `proj_rgb + pred_rgb_mem` adds a float32 and an int32 tensor; TensorFlow has no
        # implicit promotion and raises InvalidArgumentError -- the branch cannot run in the reference
        raise TypeError('models.py:340 adds float32 proj_rgb to int32 predictions')
      self.prev_rgb_frame = generated
      self.add_to_memory(pred_rgb_mem, pred_semantic_mem[..., None], pred_depth_mem, position)
    pred_rgb_u8 = pred_rgb.astype(np.uint8)   # tf.cast(., uint8); values are in [0, 255]
    return dict(proj_semantic=proj_semantic, pred_semantic=pred_semantic,
                proj_rgb=np.trunc(proj_rgb * F32(255)).astype(np.uint8),
                pred_rgb=pred_rgb_u8,
                proj_depth=proj_depth, pred_depth=pred_depth, mu=mu, logvar=logvar,
                proj_mask=proj_mask, generated=generated, pc_rgb=pc_rgb)


def generated_rollout(params, gen_cfg, inputs, eval_seq_len, predict_depth=True,
                      unproject_void_class=INVALID_RGB_VALUE, feedback=None):
  """utils/eval_metric.py `_get_generated_pool.step_fn` :144-239 (without the Inception half) /
  trainers/gan_manager.py `_get_image_grid` :458-541, statement by statement on NumPy arrays.
  inputs: image (N,T,H,W,3), depth (N,T,H,W,1), position (N,T,3), depth_scale (N,).
  `feedback[k] = (generated, depth_out)`: when given, THESE arrays (another implementation's
  generator outputs) are fed back into the memory instead of this oracle's own, so that the warp /
  quantisation half of every later frame can be compared bit for bit."""
  image, depth, position = (np.asarray(inputs[k], F32) for k in ('image', 'depth', 'position'))
  n, _, h, w, _ = image.shape
  depth_scale = float(np.asarray(inputs['depth_scale'])[0])
  memory_coords = np.zeros((n, 4, 0), F32)
  memory_feats = np.zeros((n, 0, 3), np.int32)
  out = dict(generated=[], pred_depth=[], projected=[], proj_mask=[], proj_depth=[], depth_rmse=[],
             depth_out=[])
  for k in range(eval_seq_len):
    target_depth = depth[:, k]
    rgb_tensor = image[:, k]
    depth_tensor = depth[:, k]
    rel = position[:, k]
    pred_depth, pred_rgb = warp_c.project_feats_to_equirectangular(
        memory_feats, memory_coords, h, w, INVALID_RGB_VALUE, depth_scale, offset=rel)
    pred_mask = warp_np.proj_mask(pred_depth, pred_rgb, INVALID_RGB_VALUE)
    pred_depth = pred_depth[..., None]
    pred_rgb = np.clip((pred_rgb / F32(255)).astype(F32), 0, 1)
    cond = {'proj_image': torch.from_numpy(pred_rgb), 'proj_mask': torch.from_numpy(pred_mask),
            'proj_depth': torch.from_numpy(pred_depth.copy()),
            'blurred_mask': torch.zeros(pred_depth.shape)}
    with torch.no_grad():
      outs, _ = O.generator_forward(params, cond, False, **gen_cfg)
    depth_out, generated = outs[3].numpy(), outs[6].numpy()
    out['depth_out'].append(depth_out)
    own_generated = generated
    if feedback is not None:
      generated, depth_out = feedback[k]
    if k == 0:
      rgb_tensor = warp_np.mask_pano(rgb_tensor, masked_region_value=INVALID_RGB_VALUE)
    else:
      rgb_tensor = generated
      if predict_depth and depth_out is not None:
        depth_tensor = depth_out
    m = ((target_depth > 0) & (target_depth < 1)).astype(F32)
    diff = ((depth_tensor - target_depth) ** 2 * m).sum(axis=(1, 2, 3)) / np.maximum(
        m.sum(axis=(1, 2, 3)), 1)
    out['depth_rmse'].append(np.sqrt(diff))
    pc_rgb = np.clip(np.trunc(rgb_tensor * F32(255)).astype(np.int32), INVALID_RGB_VALUE, 255)
    xyz1, feats = warp_np.equirectangular_to_pointcloud(pc_rgb, depth_tensor[..., 0],
                                                        unproject_void_class, depth_scale)
    xyz1 = (xyz1 + np.concatenate([rel, np.zeros((n, 1), F32)], 1)[:, :, None]).astype(F32)
    memory_coords = np.concatenate([memory_coords, xyz1], axis=2)
    memory_feats = np.concatenate([memory_feats, feats.astype(np.int32)], axis=1)
    out['generated'].append(own_generated)
    out['pred_depth'].append(depth_tensor)
    out['projected'].append(pred_rgb)
    out['proj_mask'].append(pred_mask)
    out['proj_depth'].append(pred_depth)
  out['memory_coords'], out['memory_feats'] = memory_coords, memory_feats
  return out
