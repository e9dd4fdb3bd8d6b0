"""Reader / writer of TensorFlow's checkpoint "tensor bundle" (`<prefix>.index` +
`<prefix>.data-00000-of-00001`), without TensorFlow -- the file format behind
`tf.train.Checkpoint.save / restore`, which the reference uses for its trainer state
(trainers/gan_manager.py:333-349) and the inference wrapper for the public 17 GB checkpoint
(models/models.py:100-104, README.md:33).  Together with utils/tf_checkpoint_keys.py (object-graph
key of every variable) this turns such a checkpoint into the generator's ParamStore and back.

PARITY UNPINNED: TensorFlow is not installed and no real checkpoint ships with the reference, so
this follows the published format (tensorflow/core/util/tensor_bundle/tensor_bundle.cc,
tensorflow/core/lib/io/{table_builder,block,format}.cc, the LevelDB table format) from its
specification; tests/test_tf_bundle.py pins what can be pinned here: writer -> reader round trips
(multi-block indexes, prefix compression, every numeric dtype), the CRC-32C known-answer vectors
of RFC 3720, and a whole toy generator through the key table.

Format, as implemented:
  index file  = a LevelDB-style table, no compression: data blocks of prefix-compressed entries
                (varint shared, varint non_shared, varint value_len, key delta, value; restart
                offsets every 16 entries + their count as fixed32 at the block's end), each block
                followed by a 5-byte trailer (type 0, masked CRC-32C of contents + type); then a
                meta-index block, an index block (last key of a data block -> BlockHandle varints
                offset, size) and the 48-byte footer (two BlockHandles padded to 40 bytes + magic
                0xdb4775248b80fb57).
  entries     key ''  -> BundleHeaderProto {1: num_shards, 2: endianness, 3: VersionDef{1: producer}}
              key k   -> BundleEntryProto {1: dtype, 2: TensorShapeProto{2: Dim{1: size}},
                                           3: shard_id, 4: offset, 5: size, 6: fixed32 crc32c}
  data file   the tensors' raw little-endian bytes at (offset, size).
String tensors (the object graph proto itself) are skipped by the reader.
"""
import os
import struct
from typing import Dict, Tuple

import numpy as np

MAGIC = 0xdb4775248b80fb57
RESTART_INTERVAL = 16
BLOCK_SIZE = 4096   # table::Options default the bundle writer keeps

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8,
           9: np.int64, 10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_STRING = 7
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------- CRC-32C
def _crc_table():
  tab = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    tab.append(c)
  return np.array(tab, dtype=np.uint32)


_TAB = _crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
  """CRC-32C (Castagnoli), as tensorflow/core/lib/hash/crc32c.h."""
  c = crc ^ 0xffffffff
  tab = _TAB
  for b in data:
    c = int(tab[(c ^ b) & 0xff]) ^ (c >> 8)
  return c ^ 0xffffffff


def crc32c_array(a: np.ndarray) -> int:
  """The same over an array's little-endian bytes.  A Python byte loop (~5 MB/s): tensor
  checksums are for small files and tests; write_bundle(checksums=False) / read_bundle(verify=False)
  skip them for the 4.5 GB generator."""
  return crc32c(np.asarray(a, order='C').tobytes())


def mask_crc(crc: int) -> int:
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def unmask_crc(masked: int) -> int:
  rot = (masked - 0xa282ead8) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------ varints
def _put_varint(x: int) -> bytes:
  out = bytearray()
  while x >= 0x80:
    out.append((x & 0x7f) | 0x80)
    x >>= 7
  out.append(x)
  return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  shift = result = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


# ------------------------------------------------------------------- tiny protobuf codec
def _pb_fields(buf: bytes):
  """Yields (field number, wire type, value) of one message; value: int (varint / fixed) or bytes."""
  pos = 0
  while pos < len(buf):
    tag, pos = _get_varint(buf, pos)
    num, wt = tag >> 3, tag & 7
    if wt == 0:
      v, pos = _get_varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wt == 2:
      n, pos = _get_varint(buf, pos)
      v = bytes(buf[pos:pos + n])
      pos += n
    elif wt == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError(f'unsupported protobuf wire type {wt}')
    yield num, wt, v


def _pb_varint(num: int, v: int) -> bytes:
  return _put_varint(num << 3) + _put_varint(v & 0xffffffffffffffff)


def _pb_bytes(num: int, v: bytes) -> bytes:
  return _put_varint((num << 3) | 2) + _put_varint(len(v)) + v


def _encode_entry(dtype_code: int, shape, offset: int, size: int, crc: int) -> bytes:
  dims = b''.join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
  out = _pb_varint(1, dtype_code) + _pb_bytes(2, dims)
  if offset:
    out += _pb_varint(4, offset)
  out += _pb_varint(5, size)
  out += _put_varint((6 << 3) | 5) + struct.pack('<I', crc)
  return out


def _decode_entry(buf: bytes):
  dtype = shard = offset = size = crc = 0
  shape = []
  for num, _, v in _pb_fields(buf):
    if num == 1:
      dtype = v
    elif num == 2:
      for n2, _, dim in _pb_fields(v):
        if n2 == 2:
          sz = 0
          for n3, _, x in _pb_fields(dim):
            if n3 == 1:
              sz = x - (1 << 64) if x >= 1 << 63 else x
          shape.append(sz)
    elif num == 3:
      shard = v
    elif num == 4:
      offset = v
    elif num == 5:
      size = v
    elif num == 6:
      crc = v
  return dtype, tuple(shape), shard, offset, size, crc


# --------------------------------------------------------------------------- table blocks
def _build_block(items) -> bytes:
  """Prefix-compressed block of sorted (key, value) pairs + restart array."""
  out, restarts, last = bytearray(), [], b''
  for i, (k, v) in enumerate(items):
    shared = 0
    if i % RESTART_INTERVAL == 0:
      restarts.append(len(out))
    else:
      m = min(len(k), len(last))
      while shared < m and k[shared] == last[shared]:
        shared += 1
    out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v))
    out += k[shared:] + v
    last = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _parse_block(buf: bytes):
  n_restarts = struct.unpack_from('<I', buf, len(buf) - 4)[0]
  end = len(buf) - 4 - 4 * n_restarts
  pos, key = 0, b''
  while pos < end:
    shared, pos = _get_varint(buf, pos)
    non_shared, pos = _get_varint(buf, pos)
    vlen, pos = _get_varint(buf, pos)
    key = key[:shared] + bytes(buf[pos:pos + non_shared])
    pos += non_shared
    yield key, bytes(buf[pos:pos + vlen])
    pos += vlen


def _handle(offset: int, size: int) -> bytes:
  return _put_varint(offset) + _put_varint(size)


class _TableWriter:
  def __init__(self):
    self.out = bytearray()

  def add_block(self, contents: bytes) -> Tuple[int, int]:
    off = len(self.out)
    self.out += contents + b'\x00' + struct.pack('<I', mask_crc(crc32c(contents + b'\x00')))
    return off, len(contents)


def _write_table(path: str, items):
  tw = _TableWriter()
  index, cur, cur_bytes = [], [], 0
  def flush():
    nonlocal cur, cur_bytes
    if cur:
      off, size = tw.add_block(_build_block(cur))
      index.append((cur[-1][0], _handle(off, size)))
      cur, cur_bytes = [], 0
  for k, v in items:
    cur.append((k, v))
    cur_bytes += len(k) + len(v) + 3
    if cur_bytes >= BLOCK_SIZE:
      flush()
  flush()
  meta = tw.add_block(_build_block([]))
  idx = tw.add_block(_build_block(index))
  footer = _handle(*meta) + _handle(*idx)
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
  tw.out += footer
  with open(path, 'wb') as f:
    f.write(bytes(tw.out))


def _read_table(path: str, verify: bool = True):
  buf = open(path, 'rb').read()
  if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
    raise ValueError(f'{path}: not a TensorFlow checkpoint index (bad table magic)')
  foot = buf[-48:]
  _, pos = _get_varint(foot, 0)
  _, pos = _get_varint(foot, pos)
  idx_off, pos = _get_varint(foot, pos)
  idx_size, pos = _get_varint(foot, pos)
  def block(off, size):
    contents, typ = buf[off:off + size], buf[off + size]
    if typ != 0:
      raise NotImplementedError('compressed table blocks (snappy) are not supported')
    if verify:
      want = struct.unpack_from('<I', buf, off + size + 1)[0]
      if mask_crc(crc32c(contents + bytes([typ]))) != want:
        raise ValueError(f'{path}: block checksum mismatch at offset {off}')
    return contents
  for _, h in _parse_block(block(idx_off, idx_size)):
    off, p = _get_varint(h, 0)
    size, _ = _get_varint(h, p)
    yield from _parse_block(block(off, size))


# -------------------------------------------------------------------------------- bundle
def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], checksums: bool = True):
  """Writes `<prefix>.index` and `<prefix>.data-00000-of-00001` (one shard, little endian)."""
  data_path = prefix + '.data-00000-of-00001'
  items = [(b'', _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1)))]   # num_shards 1, version 1
  offset = 0
  with open(data_path, 'wb') as f:
    for name in sorted(tensors, key=lambda s: s.encode()):
      a = np.asarray(tensors[name], order='C')
      if a.dtype not in _CODES:
        raise ValueError(f'{name}: dtype {a.dtype} has no TensorFlow code here')
      raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
      crc = mask_crc(crc32c_array(a)) if checksums else 0
      f.write(raw)
      items.append((name.encode(), _encode_entry(_CODES[a.dtype], a.shape, offset, len(raw), crc)))
      offset += len(raw)
  _write_table(prefix + '.index', items)


def read_bundle(prefix: str, verify: bool = False, keys=None) -> Dict[str, np.ndarray]:
  """{key: array} of every numeric tensor of the checkpoint `prefix` (string tensors such as
  `_CHECKPOINTABLE_OBJECT_GRAPH` are skipped).  verify: also check the per-tensor CRC-32C (slow in
  pure Python: meant for small files).  keys: optional subset to read."""
  index = prefix + '.index'
  if not os.path.exists(index):
    raise FileNotFoundError(index)
  num_shards, out, files = 1, {}, {}
  want = set(keys) if keys is not None else None
  for k, v in _read_table(index):
    if k == b'':
      for num, _, x in _pb_fields(v):
        if num == 1:
          num_shards = x
        if num == 2 and x != 0:
          raise NotImplementedError('big-endian bundles are not supported')
      continue
    name = k.decode()
    if want is not None and name not in want:
      continue
    dtype, shape, shard, offset, size, crc = _decode_entry(v)
    if dtype == _DT_STRING or dtype not in _DTYPES:
      continue
    path = f'{prefix}.data-{shard:05d}-of-{num_shards:05d}'
    if path not in files:
      files[path] = np.memmap(path, dtype=np.uint8, mode='r')
    raw = np.array(files[path][offset:offset + size])     # one copy out of the map, writable
    a = raw.view(np.dtype(_DTYPES[dtype]).newbyteorder('<')).reshape(shape)
    if verify and crc and mask_crc(crc32c_array(a)) != crc:
      raise ValueError(f'{name}: tensor checksum mismatch')
    out[name] = a.astype(_DTYPES[dtype], copy=False)
  return out


# ------------------------------------------------------------------- the generator's state
def load_generator(model, prefix: str, root: str = 'ema_generator', strict: bool = True):
  """Restores `tf.train.Checkpoint(ema_generator=model)` (models/models.py:100-104) from the bundle
  `prefix` into a ResNetGenerator's ParamStore through the object-graph key table."""
  from se3ds_amd.utils import tf_checkpoint_keys
  table = tf_checkpoint_keys.generator_table(model, root)   # our name -> TF key
  got = read_bundle(prefix, keys=set(table.values()))
  missing = [k for k in table.values() if k not in got]
  if missing and strict:
    raise KeyError(f'{prefix}: {len(missing)} variables missing, e.g. {missing[:3]}')
  model.store.load_dict({ours: got[tf] for ours, tf in table.items() if tf in got})
  return sorted(missing)


def save_generator(model, prefix: str, root: str = 'ema_generator'):
  """Writes the generator's variables under their TensorFlow object-graph keys in the bundle
  layout `load_generator` / `read_bundle` read back (a host-side interchange file for THIS
  library: ParamStore -> file -> ParamStore).

  NOT a TensorFlow-restorable checkpoint, on purpose (VERDICT r3 #13 -- nothing here can
  be pinned until a real checkpoint or TensorFlow is at hand, so it is not grown further):
    * per-tensor CRC-32C values are only computed for small states (<= 64 MB; the pure-Python byte
      loop runs at ~5 MB/s) and stored as 0 otherwise -- TensorFlow's BundleReader::GetValue
      verifies them and reports DataLoss for 0;
    * no `_CHECKPOINTABLE_OBJECT_GRAPH` entry is written, which `tf.train.Checkpoint.restore`
      (models/models.py:101-103, with assert_existing_objects_matched) needs.
  """
  from se3ds_amd.utils import tf_checkpoint_keys
  table = tf_checkpoint_keys.generator_table(model, root)
  state = model.store.to_dict()
  nbytes = sum(int(np.asarray(state[ours]).nbytes) for ours in table)
  write_bundle(prefix, {tf: state[ours] for ours, tf in table.items()},
               checksums=nbytes <= (64 << 20))
