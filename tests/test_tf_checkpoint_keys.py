"""CPU: the TF object-graph key table (se3ds_amd/utils/tf_checkpoint_keys.py) --
every ParamStore variable maps to exactly one distinct key whose path follows the reference's
attribute structure, and the shapes behind the keys add up to the 1.114 B-parameter generator."""
import numpy as np
import pytest

from se3ds_amd import gin_lite
from se3ds_amd.hipops import nn
from se3ds_amd.models import image_models
from se3ds_amd.utils import tf_checkpoint_keys as K


def test_generator_table_is_a_bijection_with_reference_paths():
  gin_lite.clear_config()
  G = image_models.ResNetGenerator(image_size=128, gen_dims=4, z_dim=4, resnet_version='101',
                                   device='cpu', seed=None)
  tab = K.generator_table(G)
  names = G.store.trainable_names + G.store.state_names
  assert set(tab) == set(names) and len(set(tab.values())) == len(names)
  assert all(v.startswith('ema_generator/') and v.endswith(K.SUFFIX) for v in tab.values())
  sfx = K.SUFFIX
  # spot checks against the constructors (image_models.py:55-128,220-274,351-441; layers.py:233-251,
  # 373-388,417-444,472-508)
  expect = {
      'encoder/conv1/kernel': 'ema_generator/encoder/conv1/kernel',
      'encoder/bn1/moving_mean': 'ema_generator/encoder/act1/layer_with_weights-0/moving_mean',
      'encoder/stack2/downsample/u': 'ema_generator/encoder/stack2/blocks/0/downsample/u',
      'encoder/stack3/block22/conv2/bias': 'ema_generator/encoder/stack3/blocks/22/conv2/bias',
      'encoder/stack3/block22/bn2/gamma': 'ema_generator/encoder/stack3/blocks/22/act2/layer_with_weights-0/gamma',
      'encoder/stack1/block0/bn3/beta': 'ema_generator/encoder/stack1/blocks/0/act3/beta',
      'encoder/stack1/block0/ds_norm/gamma': 'ema_generator/encoder/stack1/blocks/0/ds_norm/gamma',
      'encoder/final_bn/gamma': 'ema_generator/encoder/final_act/layer_with_weights-0/gamma',
      'decoder/upc/conv/u': 'ema_generator/decoder/upc/layer_with_weights-0/u',
      'decoder/upc/bn/beta': 'ema_generator/decoder/upc/layer_with_weights-1/beta',
      'decoder/agent3/kernel': 'ema_generator/decoder/agent3/kernel',
      'decoder/agent3_bn/gamma': 'ema_generator/decoder/agent3_act/layer_with_weights-0/gamma',
      # deconv1: 23 blocks, stride 1, 1024 -> 512 channels: 1x1 conv upsample on the last block
      'decoder/deconv1/block5/conv_a/kernel':
          'ema_generator/decoder/deconv1/block/layer_with_weights-5/main/layer_with_weights-0/kernel',
      'decoder/deconv1/block5/conv_b/kernel':
          'ema_generator/decoder/deconv1/block/layer_with_weights-5/main/layer_with_weights-2/layer_with_weights-0/kernel',
      'decoder/deconv1/block22/bn_b/moving_variance':
          'ema_generator/decoder/deconv1/block/layer_with_weights-22/main/layer_with_weights-3/moving_variance',
      'decoder/deconv1/upsample/conv/kernel':
          'ema_generator/decoder/deconv1/block/layer_with_weights-22/upsample/layer_with_weights-0/kernel',
      # deconv2: last block upsamples with Conv2DTranspose k3 (main) and k2 (upsample)
      'depth_decoder/deconv2/block3/conv_b/kernel':
          'ema_generator/depth_decoder/deconv2/block/layer_with_weights-3/main/layer_with_weights-2/kernel',
      'depth_decoder/deconv2/upsample/bn/gamma':
          'ema_generator/depth_decoder/deconv2/block/layer_with_weights-3/upsample/layer_with_weights-1/gamma',
      'depth_decoder/final_conv/block2/conv_b/kernel':
          'ema_generator/depth_decoder/final_conv/block/layer_with_weights-2/main/layer_with_weights-2/layer_with_weights-0/kernel',
      'depth_decoder/final_deconv/bias': 'ema_generator/depth_decoder/final_deconv/bias',
      'rgb_conv/bn0/gamma': 'ema_generator/rgb_conv/layer_with_weights-0/gamma',
      'rgb_conv/conv2/kernel': 'ema_generator/rgb_conv/layer_with_weights-5/kernel',
      'depth_conv/conv1/u': 'ema_generator/depth_conv/layer_with_weights-3/u',
      'context/bn3/moving_mean': 'ema_generator/global_context_layer/layer_with_weights-6/moving_mean',
      'context/conv3/bias': 'ema_generator/global_context_layer/layer_with_weights-7/bias',
  }
  for k, v in expect.items():
    assert tab[k] == v + sfx, (k, tab[k])


def test_discriminator_table():
  D = image_models.SNMultiScaleDiscriminator(n_dis=2, dis_dims=4, n_layers=6, device='cpu', seed=None)
  tab = K.discriminator_table(D)
  names = D.store.trainable_names + D.store.state_names
  assert set(tab) == set(names) and len(set(tab.values())) == len(names)
  sfx = K.SUFFIX
  assert tab['dis0/g0/conv/kernel'] == 'discriminator/discriminators/0/discriminator_groups/0/layer_with_weights-0/kernel' + sfx
  assert tab['dis1/g3/conv/u'] == 'discriminator/discriminators/1/discriminator_groups/3/layer_with_weights-0/u' + sfx
  assert tab['dis1/g3/in/gamma'] == 'discriminator/discriminators/1/discriminator_groups/3/layer_with_weights-1/gamma' + sfx
  assert tab['dis0/final/bias'] == 'discriminator/discriminators/0/discriminator_groups/6/bias' + sfx


def test_shapes_add_up_to_the_shipped_generator(monkeypatch):
  """gen_dims 128, ResNet-101 (highres.gin / lowres.gin): 1 113.7 M convolution weights + biases /
  affine parameters, SURVEY 8d -- counted from the variable shapes without allocating them."""
  def light_finalize(self, device, generator=None):
    self.theta = self.state = None
    self.trainable_names = [s[0] for s in self._specs if s[3]]
    self.state_names = [s[0] for s in self._specs if not s[3]]
    return self
  monkeypatch.setattr(nn.ParamStore, 'finalize', light_finalize)
  monkeypatch.setattr(image_models._Model, '_finish',
                      lambda self, device, seed, dtype: self.store.finalize(device))
  gin_lite.clear_config()
  G = image_models.ResNetGenerator(image_size=512, gen_dims=128, resnet_version='101', device='cpu')
  tab = K.generator_table(G)
  shapes = {s[0]: s[1] for s in G.store._specs}
  assert set(tab) == set(shapes)
  trainable = sum(int(np.prod(shapes[n])) for n in G.store.trainable_names)
  kernels = sum(int(np.prod(shapes[n])) for n in G.store.trainable_names if n.endswith('/kernel'))
  assert abs(kernels / 1e6 - 1113.7) < 1.0, kernels
  assert 1.113e9 < trainable < 1.116e9, trainable
  # Keras layouts behind the keys: Conv2D HWIO, Conv2DTranspose (kh, kw, out, in), u (1, out)
  assert shapes['decoder/deconv1/block0/conv_a/kernel'] == (3, 3, 1024, 1024)
  assert shapes['decoder/deconv2/block3/conv_b/kernel'] == (3, 3, 256, 512)
  assert shapes['decoder/deconv2/upsample/conv/kernel'] == (2, 2, 256, 512)
  assert shapes['encoder/conv1/kernel'] == (7, 7, 5, 128)
  assert shapes['rgb_conv/conv2/u'] == (1, 3)
