"""Variable-path table between this package's ParamStore names and the keys of the reference's
TensorFlow object-graph checkpoints.

The reference saves `tf.train.Checkpoint(generator=..., discriminator=..., ema_generator=...,
g_optimizer=..., d_optimizer=...)` (trainers/gan_manager.py:340-349) and the inference wrapper
restores `tf.train.Checkpoint(ema_generator=model)` (models/models.py:100-104).  An object-graph
checkpoint names a variable by the path of tracked-attribute edges from the root object, followed
by `/.ATTRIBUTES/VARIABLE_VALUE`.  The edges follow from the reference's constructors:

  * attributes assigned in `__init__` of a subclassed `tf.keras.Model` / layer are edges by their
    attribute name (`encoder`, `conv1`, `kernel`, `u`, ...): image_models.py:55-128,220-274,
    351-441; layers.py:233-251,373-388;
  * a Python list attribute is a ListWrapper whose edges are the indices (`blocks/0`):
    layers.py:373-388;
  * a `tf.keras.Sequential` reaches its layers through `layer_with_weights-<k>` (k counts only
    the layers that own variables; `layer-<i>` edges exist too but are listed later, so the
    breadth-first naming uses the former): image_models.py:79-128,224-227,271-274,316-322,
    367-379; layers.py:235-244,417-444,472-486,508;
  * Keras variables are edges by their attribute name on the owning layer: Conv2D / Conv2DTranspose
    `kernel`, `bias`; SpectralConv / PartialSpectralConv additionally `u` (layers.py:122-128,
    287-293); SyncBatchNormalization `gamma`, `beta`, `moving_mean`, `moving_variance`.

TensorFlow cannot be run here, so the table is DERIVED from those rules, not read from a real
checkpoint ("unpinned" in the sense of DESIGN.md section 4); tests/test_tf_checkpoint_keys.py pins
what can be pinned without TF: the map is a bijection onto distinct keys, covers every variable,
and the shapes add up to the 1.114 B-parameter generator of the shipped configurations.
INTEGRATION.md section 5 shows the TF-side exporter that uses this table.
"""
import re
from typing import Dict

SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'

_BN = {'gamma', 'beta', 'moving_mean', 'moving_variance'}


def _lww(k):
  return f'layer_with_weights-{k}'


def _trans_stack(rest, blocks_in_stack):
  """ResStackTranspose (layers.py:458-511): block<i>/{conv_a,bn_a,conv_b,bn_b}, upsample/{conv,bn}."""
  m = re.match(r'block(\d+)/(conv_a|bn_a|conv_b|bn_b)/(\w+)$', rest)
  if m:
    i, part, var = int(m.group(1)), m.group(2), m.group(3)
    base = f'block/{_lww(i)}/main/'
    if part == 'conv_a':
      return base + f'{_lww(0)}/{var}'
    if part == 'bn_a':
      return base + f'{_lww(1)}/{var}'
    if part == 'bn_b':
      return base + f'{_lww(3)}/{var}'
    # conv_b: Conv2DTranspose directly (upsampling last block), else Sequential([PadLayer, conv])
    return None, base + f'{_lww(2)}/', var
  m = re.match(r'upsample/(conv|bn)/(\w+)$', rest)
  if m:
    last = blocks_in_stack - 1
    return f'block/{_lww(last)}/upsample/{_lww(0 if m.group(1) == "conv" else 1)}/{m.group(2)}'
  raise KeyError(rest)


def generator_key(name: str, transposed_conv_b, blocks) -> str:
  """TF attribute path (without root and suffix) of the generator variable `name`.
  transposed_conv_b(stack_path, i) -> bool: block i of that ResStackTranspose upsamples with a
  Conv2DTranspose; blocks(stack_path) -> number of blocks."""
  top, rest = name.split('/', 1)
  if top in ('rgb_conv', 'depth_conv', 'context'):
    tf_top = 'global_context_layer' if top == 'context' else top
    m = re.match(r'(bn|conv)(\d+)/(\w+)$', rest)
    k = 2 * int(m.group(2)) + (1 if m.group(1) == 'conv' else 0)
    return f'{tf_top}/{_lww(k)}/{m.group(3)}'
  if top == 'encoder':
    m = re.match(r'(conv1|final_conv)/(\w+)$', rest)
    if m:
      return f'encoder/{m.group(1)}/{m.group(2)}'
    m = re.match(r'(bn1|final_bn)/(\w+)$', rest)
    if m:
      return f'encoder/{"act1" if m.group(1) == "bn1" else "final_act"}/{_lww(0)}/{m.group(2)}'
    m = re.match(r'(stack\d)/downsample/(\w+)$', rest)
    if m:   # created in ResStack.__init__, owned by blocks[0].downsample (layers.py:373-380)
      return f'encoder/{m.group(1)}/blocks/0/downsample/{m.group(2)}'
    m = re.match(r'(stack\d)/block(\d+)/(conv[123]|bn[123]|ds_norm)/(\w+)$', rest)
    if m:
      st, i, part, var = m.groups()
      base = f'encoder/{st}/blocks/{i}/'
      if part.startswith('conv') or part == 'ds_norm':
        return base + f'{part}/{var}'
      j = part[-1]
      # act1 / act2 are Sequential([SyncBN, ReLU]); act3 is the SyncBN itself (layers.py:235-247)
      return base + (f'act{j}/{_lww(0)}/{var}' if j in '12' else f'act3/{var}')
    raise KeyError(name)
  if top in ('decoder', 'depth_decoder'):
    m = re.match(r'upc/(conv|bn)/(\w+)$', rest)
    if m:
      return f'{top}/upc/{_lww(0 if m.group(1) == "conv" else 1)}/{m.group(2)}'
    m = re.match(r'(agent\d)/(\w+)$', rest)
    if m:
      return f'{top}/{m.group(1)}/{m.group(2)}'
    m = re.match(r'(agent\d)_bn/(\w+)$', rest)
    if m:
      return f'{top}/{m.group(1)}_act/{_lww(0)}/{m.group(2)}'
    m = re.match(r'final_deconv/(\w+)$', rest)
    if m:
      return f'{top}/final_deconv/{m.group(1)}'
    m = re.match(r'(deconv\d|final_conv)/(.+)$', rest)
    if m:
      stack, sub = m.group(1), m.group(2)
      path = f'{top}/{stack}'
      r = _trans_stack(sub, blocks(path))
      if isinstance(r, tuple):
        _, base, var = r
        i = int(re.match(r'block(\d+)/', sub).group(1))
        inner = '' if transposed_conv_b(path, i) else f'{_lww(0)}/'
        return f'{path}/{base}{inner}{var}'
      return f'{path}/{r}'
  raise KeyError(name)


def generator_table(generator, root: str = 'ema_generator') -> Dict[str, str]:
  """{ParamStore name: full TF checkpoint key} for a ResNetGenerator of this package."""
  from se3ds_amd.models import layers
  stacks = {}
  for dec_name in ('decoder', 'depth_decoder'):
    dec = getattr(generator, dec_name)
    for st in ('deconv1', 'deconv2', 'deconv3', 'deconv4', 'final_conv'):
      stacks[f'{dec_name}/{st}'] = getattr(dec, st)
  def transposed(path, i):
    return bool(stacks[path].block[i].transposed)
  def nblocks(path):
    return len(stacks[path].block)
  del layers
  names = generator.store.trainable_names + generator.store.state_names
  return {n: f'{root}/{generator_key(n, transposed, nblocks)}{SUFFIX}' for n in names}


def discriminator_key(name: str, n_layers: int) -> str:
  """SNMultiScaleDiscriminator (image_models.py:564-618): `discriminators` is a list of
  SNPatchDiscriminator, each with the list `discriminator_groups` (:510-541): n_layers Sequential
  groups -- group 0 = [PadLayer, Conv2D, LeakyReLU], groups >= 1 = [PadLayer, SpectralConv,
  InstanceNormalization, LeakyReLU] -- followed by the final Conv2D itself."""
  m = re.match(r'dis(\d+)/g(\d+)/(conv|in)/(\w+)$', name)
  if m:
    d, g, part, var = m.groups()
    return f'discriminators/{d}/discriminator_groups/{g}/{_lww(0 if part == "conv" else 1)}/{var}'
  m = re.match(r'dis(\d+)/final/(\w+)$', name)
  if m:
    return f'discriminators/{m.group(1)}/discriminator_groups/{n_layers}/{m.group(2)}'
  raise KeyError(name)


def discriminator_table(discriminator, root: str = 'discriminator') -> Dict[str, str]:
  n_layers = len(discriminator.discriminators[0].groups) + 1
  names = discriminator.store.trainable_names + discriminator.store.state_names
  return {n: f'{root}/{discriminator_key(n, n_layers)}{SUFFIX}' for n in names}
