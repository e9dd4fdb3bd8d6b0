"""The autoregressive evaluation roll-out -- the loop body the reference runs in
utils/eval_metric.py `_get_generated_pool.step_fn` (:144-239) and in trainers/gan_manager.py
`_get_image_grid` (:458-541): project the point-cloud memory to the next position, mask, run the
generator in inference mode, quantise, unproject the (ground-truth first, then generated) frame and
append it to the memory -- as ONE on-device pipeline.  Every arithmetic step runs in
libse3ds_hip.so; the memory lives in preallocated HBM buffers (PointCloudMemory), so no frame ever
copies it.  The Inception / FID half of the reference's evaluator is out of scope."""
from typing import Callable, Dict, List, NamedTuple, Optional

import torch

from se3ds_amd import _lib
from se3ds_amd import constants
from se3ds_amd.models.models import _quantize
from se3ds_amd.utils import pano_utils
from se3ds_amd.utils import point_cloud_utils
from se3ds_amd.utils.point_cloud_utils import PointCloudMemory


class RolloutOutput(NamedTuple):
  generated: List[torch.Tensor]     # per frame (N,H,W,3) fp32 in [0,1]
  pred_depth: List[torch.Tensor]    # per frame (N,H,W,1): the depth that entered the memory
  projected: List[torch.Tensor]     # per frame proj_image (N,H,W,3) fp32
  proj_mask: List[torch.Tensor]     # per frame (N,H,W,1)
  proj_depth: List[torch.Tensor]    # per frame (N,H,W,1)
  depth_rmse: List[torch.Tensor]    # per frame (N,) (eval_metric.py:225-234)
  memory: PointCloudMemory


def depth_rmse(depth: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
  """sqrt(sum((d - t)^2 * 1[0 < t < 1]) / max(count, 1)) per sample (eval_metric.py:225-234)."""
  n = depth.shape[0]
  p = depth.numel() // n
  dev = depth.device
  L = _lib.lib()
  depth, target = depth.contiguous(), target.contiguous()
  ws = torch.empty(n * 256, dtype=torch.float32, device=dev)
  num = torch.empty(n, dtype=torch.float32, device=dev)
  cnt = torch.empty(n, dtype=torch.float32, device=dev)
  _lib.check(L.se3ds_sample_sum(depth.data_ptr(), target.data_ptr(), None, n, p, 1, 4,
                                num.data_ptr(), ws.data_ptr(), _lib.stream()), 'se3ds_sample_sum')
  _lib.check(L.se3ds_sample_sum(target.data_ptr(), None, None, n, p, 1, 2, cnt.data_ptr(),
                                ws.data_ptr(), _lib.stream()), 'se3ds_sample_sum')
  # N scalars: finished on the host when read (the loop itself never synchronises)
  return torch.sqrt(num / torch.clamp(cnt, min=1))


def generated_rollout(generator_fn: Callable, inputs: Dict[str, torch.Tensor], eval_seq_len: int,
                      predict_depth: bool = True,
                      unproject_void_class: float = constants.INVALID_RGB_VALUE) -> RolloutOutput:
  """inputs: image (N,T,H,W,3) fp32 [0,1], depth (N,T,H,W,1), position (N,T,3), depth_scale (N,).
  generator_fn(inputs=[cond, None], training=False) -> [mu, logvar, kld, depth, seg, depth_seg,
  rgb].  `unproject_void_class`: eval_metric.py:236-238 passes INVALID_RGB_VALUE, gan_manager.py:
  536-539 passes 0 -- both are kept.  predict_depth=False feeds the ground-truth depth of every
  frame (gan_manager.py:468-469)."""
  image, depth, position = inputs['image'], inputs['depth'], inputs['position']
  _lib.require_cuda(image, depth, position)
  n, t, h, w, _ = image.shape
  if eval_seq_len > t:
    raise ValueError(f'eval_seq_len {eval_seq_len} exceeds the {t} frames of the batch')
  depth_scale = float(inputs['depth_scale'][0])   # all depth_scale within a batch are the same
  dev = image.device
  memory = PointCloudMemory(n, 3, torch.int32, dev, capacity=eval_seq_len * h * w)
  out = RolloutOutput([], [], [], [], [], [], memory)
  prev_rgb = None
  for k in range(eval_seq_len):
    target_depth = depth[:, k].contiguous()
    rgb = image[:, k].contiguous()
    depth_tensor = target_depth
    pos = position[:, k].contiguous()
    # memory - position, projection, splat and the mask (eval_metric.py:160-172) in one call
    pred_depth, pred_rgb, pred_mask = memory.project(
        h, w, constants.INVALID_RGB_VALUE, depth_scale, position=pos, with_mask=True,
        mask_void=constants.INVALID_RGB_VALUE)
    pred_rgb = _quantize(pred_rgb, torch.float32, div=255.0, lo=0.0, hi=1.0)
    if prev_rgb is None:
      prev_rgb = torch.zeros_like(rgb)
    first = 1.0 if k == 0 else 0.0
    cond = {
        'prev_image': prev_rgb, 'proj_image': pred_rgb, 'proj_mask': pred_mask[..., None],
        'proj_depth': pred_depth[..., None], 'blurred_mask': torch.zeros_like(pred_depth)[..., None],
        'first_frame': torch.full((n,), first, device=dev),
        'dataset_type': inputs.get('dataset_type'), 'depth': depth_tensor,
    }
    outs = generator_fn(inputs=[cond, None], training=False)
    depth_out, generated = outs[3], outs[6]
    if k == 0:
      prev_rgb = rgb
      # ground truth: the blurred top / bottom rows never enter the memory (:213-217)
      rgb_mem = pano_utils.mask_pano(rgb, masked_region_value=constants.INVALID_RGB_VALUE)
    else:
      rgb_mem = generated
      prev_rgb = generated
      if predict_depth and depth_out is not None:
        depth_tensor = depth_out
    out.depth_rmse.append(depth_rmse(depth_tensor, target_depth))
    # int32(rgb * 255) clipped to [-1, 255] (:234-236), unprojected at the frame's position
    pc_rgb = _quantize(rgb_mem, torch.int32, mul=255.0, lo=constants.INVALID_RGB_VALUE, hi=255)
    memory.append_equirect(pc_rgb, depth_tensor[..., 0], unproject_void_class, depth_scale,
                           position=pos)
    out.generated.append(generated)
    out.pred_depth.append(depth_tensor)
    out.projected.append(pred_rgb)
    out.proj_mask.append(pred_mask[..., None])
    out.proj_depth.append(pred_depth[..., None])
  point_cloud_utils.check_promise(memory.device)   # (the roll-out's results are read next)
  return out
