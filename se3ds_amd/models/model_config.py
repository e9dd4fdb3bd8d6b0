"""Config object for SE3DS models (reference models/model_config.py:22-61)."""
from typing import Optional

from se3ds_amd import constants


class SE3DSConfig:
  """Parameters used to configure SE3DS models."""
  batch_size: int = 1
  ckpt_path: Optional[str] = constants.CKPT_UNSEEN
  hidden_dims: int = 128
  random_noise: bool = True
  z_dim: int = 32
  circular_pad: bool = True
  depth_scale: float = constants.DEPTH_SCALE
  gen_dims: int = 128
  image_height: int = 512
  h_fov: float = 0.17
  resnet_version: str = '101'
  use_blurred_mask: bool = True


def get_config() -> SE3DSConfig:
  """Returns the Val-Unseen config for SE3DS."""
  config = SE3DSConfig()
  config.ckpt_path = constants.CKPT_UNSEEN
  config.resnet_version = '101'
  return config


def get_re10k_config() -> SE3DSConfig:
  """Returns the RealEstate10K config for SE3DS."""
  config = SE3DSConfig()
  config.ckpt_path = constants.CKPT_RE10K
  config.resnet_version = '101'
  config.use_blurred_mask = False
  return config


def get_test_config() -> SE3DSConfig:
  """Returns config used for unit tests."""
  config = SE3DSConfig()
  config.ckpt_path = None
  config.hidden_dims = 4
  config.z_dim = 4
  config.gen_dims = 4
  return config
