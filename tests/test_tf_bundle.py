"""TensorFlow checkpoint bundle reader / writer (se3ds_amd/utils/tf_bundle.py): CRC-32C known
answers, table round trips across block boundaries, dtype coverage, corruption detection, and a
toy generator through the object-graph key table (models/models.py:100-104's restore path)."""
import os
import struct

import numpy as np
import pytest

from se3ds_amd.utils import tf_bundle as B


def test_crc32c_known_answers():
  # RFC 3720 appendix B.4 + the check value of the CRC catalogue
  assert B.crc32c(b'123456789') == 0xE3069283
  assert B.crc32c(bytes(32)) == 0x8A9136AA
  assert B.crc32c(bytes([0xff] * 32)) == 0x62A8AB43
  assert B.crc32c(bytes(range(32))) == 0x46DD794E
  assert B.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
  a = np.arange(32, dtype=np.uint8)
  assert B.crc32c_array(a) == 0x46DD794E
  # incremental form and the mask are inverses
  assert B.crc32c(b'6789', B.crc32c(b'12345')) == 0xE3069283
  for c in (0, 1, 0xE3069283, 0xffffffff):
    assert B.unmask_crc(B.mask_crc(c)) == c
  assert B.mask_crc(0xE3069283) != 0xE3069283


def test_varint_and_proto_entry_round_trip():
  for v in (0, 1, 127, 128, 300, 2 ** 32 - 1, 2 ** 40 + 5):
    enc = B._put_varint(v)
    assert B._get_varint(enc, 0) == (v, len(enc))
  e = B._encode_entry(1, (3, 3, 128, 256), 123456789012, 4 * 3 * 3 * 128 * 256, 0xdeadbeef)
  assert B._decode_entry(e) == (1, (3, 3, 128, 256), 0, 123456789012, 1179648, 0xdeadbeef)
  # scalar: no dims, offset 0 omitted
  e = B._encode_entry(9, (), 0, 8, 7)
  assert B._decode_entry(e) == (9, (), 0, 0, 8, 7)


def test_block_prefix_compression_and_restarts():
  items = [(f'ema_generator/decoder/deconv{i // 40}/blocks/{i:03d}/kernel'.encode(), bytes([i % 251]) * (i % 7))
           for i in range(100)]
  blk = B._build_block(items)
  assert list(B._parse_block(blk)) == items
  n_restarts = struct.unpack_from('<I', blk, len(blk) - 4)[0]
  assert n_restarts == 7    # ceil(100 / 16)
  # compression did happen: far smaller than the plain concatenation of keys
  assert len(blk) < sum(len(k) + len(v) for k, v in items) // 2
  assert list(B._parse_block(B._build_block([]))) == []


def _tensors(rng, n):
  out = {}
  dts = [np.float32, np.float64, np.int32, np.uint8, np.int16, np.int8, np.int64, np.bool_,
         np.uint16, np.float16, np.uint32, np.uint64]
  for i in range(n):
    dt = dts[i % len(dts)]
    shape = tuple(int(x) for x in rng.integers(1, 5, size=i % 4))
    a = (rng.standard_normal(shape) * 50).astype(dt) if dt != np.bool_ else rng.random(shape) < 0.5
    out[f'model/layer_with_weights-{i % 13}/sub/{i}/.ATTRIBUTES/VARIABLE_VALUE'] = np.asarray(a, dtype=dt)
  out['save_counter/.ATTRIBUTES/VARIABLE_VALUE'] = np.asarray(7, dtype=np.int64)
  return out


def test_bundle_round_trip_many_keys_multi_block(tmp_path):
  rng = np.random.default_rng(0)
  tensors = _tensors(rng, 400)      # ~40 kB of index entries: several 4 kB data blocks
  prefix = str(tmp_path / 'ckpt-1')
  B.write_bundle(prefix, tensors)
  assert os.path.getsize(prefix + '.index') > 3 * B.BLOCK_SIZE
  got = B.read_bundle(prefix, verify=True)
  assert set(got) == set(tensors)
  for k, v in tensors.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
    np.testing.assert_array_equal(got[k], v)
  # subset read
  some = sorted(tensors)[5:9]
  sub = B.read_bundle(prefix, keys=some)
  assert sorted(sub) == some
  # the table iterates keys in bytewise order, header first
  keys = [k for k, _ in B._read_table(prefix + '.index')]
  assert keys[0] == b'' and keys == sorted(keys)


def test_bundle_detects_corruption(tmp_path):
  prefix = str(tmp_path / 'c')
  B.write_bundle(prefix, {'a/.ATTRIBUTES/VARIABLE_VALUE': np.arange(100, dtype=np.float32)})
  raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  raw[17] ^= 0x40
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
  B.read_bundle(prefix)                       # unverified read still works
  with pytest.raises(ValueError, match='checksum'):
    B.read_bundle(prefix, verify=True)
  idx = bytearray(open(prefix + '.index', 'rb').read())
  idx[3] ^= 0x01
  open(prefix + '.index', 'wb').write(bytes(idx))
  with pytest.raises(ValueError, match='checksum'):
    B.read_bundle(prefix)
  idx[-1] ^= 0xff
  open(prefix + '.index', 'wb').write(bytes(idx))
  with pytest.raises(ValueError, match='magic'):
    B.read_bundle(prefix)
  with pytest.raises(FileNotFoundError):
    B.read_bundle(str(tmp_path / 'nope'))


def test_string_tensors_are_skipped(tmp_path):
  prefix = str(tmp_path / 's')
  B.write_bundle(prefix, {'w/.ATTRIBUTES/VARIABLE_VALUE': np.ones((2, 2), np.float32)})
  # splice an object-graph entry (DT_STRING) into the index the way TF writes one
  items = list(B._read_table(prefix + '.index'))
  items.append((b'_CHECKPOINTABLE_OBJECT_GRAPH', B._encode_entry(B._DT_STRING, (), 16, 5, 0)))
  B._write_table(prefix + '.index', sorted(items))
  got = B.read_bundle(prefix)
  assert list(got) == ['w/.ATTRIBUTES/VARIABLE_VALUE']


def test_generator_through_the_key_table(tmp_path):
  from se3ds_amd.models import image_models
  from se3ds_amd.utils import tf_checkpoint_keys
  G = image_models.ResNetGenerator(image_size=64, gen_dims=4, z_dim=4, device='cpu', seed=3)
  prefix = str(tmp_path / 'ckpt-7')
  B.save_generator(G, prefix)
  raw = B.read_bundle(prefix)
  table = tf_checkpoint_keys.generator_table(G)
  assert set(raw) == set(table.values())
  assert all(k.startswith('ema_generator/') and k.endswith('/.ATTRIBUTES/VARIABLE_VALUE') for k in raw)
  G2 = image_models.ResNetGenerator(image_size=64, gen_dims=4, z_dim=4, device='cpu', seed=11)
  assert not np.array_equal(G2.store.theta.numpy(), G.store.theta.numpy())
  v0 = G2.store.version
  assert B.load_generator(G2, prefix) == []
  assert G2.store.version > v0          # operand caches are invalidated
  a, b = G.store.to_dict(), G2.store.to_dict()
  assert set(a) == set(b)
  for k in a:
    np.testing.assert_array_equal(a[k], b[k], err_msg=k)
  # a checkpoint of the TRAINING generator has other keys: strict load says so, lenient one lists them
  B.save_generator(G, str(tmp_path / 'g'), root='generator')
  with pytest.raises(KeyError, match='missing'):
    B.load_generator(G2, str(tmp_path / 'g'))
  assert len(B.load_generator(G2, str(tmp_path / 'g'), strict=False)) == len(table)
  assert B.load_generator(G2, str(tmp_path / 'g'), root='generator') == []
