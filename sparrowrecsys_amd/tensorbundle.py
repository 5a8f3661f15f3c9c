"""TensorBundle (TF2 SavedModel ``variables/``) reader that needs no TensorFlow.

The reference exports its models with ``tf.keras.models.save_model``
(reference: TFRecModel/src/com/sparrowrecsys/offline/tensorflow/NeuralCF.py:97-105)
and ships trained checkpoints under
``src/main/resources/webroot/modeldata/{neuralcf,MLPRec}/*/variables/``.
This module reads those files so a reference-trained model can be served by the
HIP path without TensorFlow.

On-disk format (written from the published TensorBundle / LevelDB-table layout):

* ``variables.index`` is an SSTable: data blocks, a metaindex block, an index
  block and a 48-byte footer ``[metaindex handle][index handle][padding][magic]``
  where a handle is two varint64 (offset, size) and the magic is
  ``0xdb4775248b80fb57`` little-endian.
* A block is a run of prefix-compressed entries
  ``varint32 shared | varint32 non_shared | varint32 value_len | key delta | value``
  followed by a uint32 restart array and a uint32 restart count.  Each block is
  followed on disk by a 1-byte compression tag and a 4-byte CRC (not part of the
  handle's size).  TF writes bundles uncompressed.
* The index block maps separator keys to data-block handles.  In a data block the
  empty key holds ``BundleHeaderProto``; every other key is a tensor name and its
  value a ``BundleEntryProto``: 1=dtype 2=shape{2=dim{1=size}} 3=shard_id 4=offset
  5=size 6=crc32c(fixed32).
* ``variables.data-XXXXX-of-YYYYY`` holds raw little-endian tensor bytes at
  ``[offset, offset+size)``.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto DataType enum (subset that bundles hold)
_DTYPES = {
    1: np.dtype("<f4"),
    2: np.dtype("<f8"),
    3: np.dtype("<i4"),
    4: np.dtype("u1"),
    5: np.dtype("<i2"),
    6: np.dtype("i1"),
    9: np.dtype("<i8"),
    10: np.dtype("bool"),
    17: np.dtype("<u2"),
    19: np.dtype("<f2"),
    22: np.dtype("<u4"),
    23: np.dtype("<u8"),
}
_DT_STRING = 7


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError("block too small")
    (num_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    limit = len(block) - 4 - 4 * num_restarts
    if limit < 0:
        raise ValueError("bad restart array")
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        value = block[pos:pos + vlen]
        pos += vlen
        yield key, value


def _read_block(data: bytes, offset: int, size: int) -> bytes:
    if offset + size + 1 > len(data):
        raise ValueError("block handle out of range")
    tag = data[offset + size]
    if tag != 0:
        raise ValueError("compressed TensorBundle index blocks (tag %d) are not supported" % tag)
    return data[offset:offset + size]


def _parse_proto(buf: bytes) -> Dict[int, List]:
    """Minimal protobuf wire decoder: field number -> list of raw values."""
    out: Dict[int, List] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(val)
    return out


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for dim in _parse_proto(buf).get(2, []):
        d = _parse_proto(dim)
        size = d.get(1, [0])[0]
        if size >= 1 << 63:
            size -= 1 << 64
        dims.append(int(size))
    return tuple(dims)


class BundleEntry:
    __slots__ = ("name", "dtype", "shape", "shard", "offset", "size", "crc32c")

    def __init__(self, name, dtype, shape, shard, offset, size, crc32c):
        self.name = name
        self.dtype = dtype
        self.shape = shape
        self.shard = shard
        self.offset = offset
        self.size = size
        self.crc32c = crc32c

    def __repr__(self):
        return "BundleEntry(%r, dtype=%s, shape=%s, off=%d, size=%d)" % (
            self.name, self.dtype, self.shape, self.offset, self.size)


def read_index(index_path: str) -> Tuple[int, Dict[str, BundleEntry]]:
    """Parse ``variables.index`` -> (num_shards, {tensor name: BundleEntry})."""
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError("not a TensorBundle index: file shorter than the footer")
    footer = data[-48:]
    (magic,) = struct.unpack_from("<Q", footer, 40)
    if magic != _MAGIC:
        raise ValueError("not a TensorBundle index: bad table magic")
    pos = 0
    _mi_off, pos = _varint(footer, pos)
    _mi_size, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)

    entries: Dict[str, BundleEntry] = {}
    num_shards = 1
    for _sep, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, value in _block_entries(_read_block(data, boff, bsize)):
            if key == b"":
                hdr = _parse_proto(value)
                num_shards = int(hdr.get(1, [1])[0])
                continue
            msg = _parse_proto(value)
            dt = int(msg.get(1, [0])[0])
            shape = _parse_shape(msg[2][0]) if 2 in msg else ()
            entries[key.decode("utf-8")] = BundleEntry(
                key.decode("utf-8"), dt, shape,
                int(msg.get(3, [0])[0]), int(msg.get(4, [0])[0]),
                int(msg.get(5, [0])[0]), int(msg.get(6, [0])[0]))
    return num_shards, entries


def load_bundle(variables_dir: str, prefix: str = "variables",
                skip_missing_shards: bool = False) -> Dict[str, np.ndarray]:
    """Load every numeric tensor of a bundle into ``{name: ndarray}``.

    ``variables_dir`` is the SavedModel's ``variables/`` directory.  String
    tensors (object-graph bookkeeping) are skipped.  With
    ``skip_missing_shards`` a tensor whose data shard is absent is skipped
    instead of raising (some reference checkpoints are shipped without data).
    """
    num_shards, entries = read_index(os.path.join(variables_dir, prefix + ".index"))
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    for name, e in entries.items():
        if e.dtype == _DT_STRING or e.dtype not in _DTYPES:
            continue
        if e.shard not in shards:
            path = os.path.join(variables_dir, "%s.data-%05d-of-%05d" % (prefix, e.shard, num_shards))
            if not os.path.exists(path):
                if skip_missing_shards:
                    shards[e.shard] = None
                else:
                    raise FileNotFoundError(path)
            else:
                shards[e.shard] = np.memmap(path, dtype=np.uint8, mode="r")
        mm = shards[e.shard]
        if mm is None:
            continue
        dt = _DTYPES[e.dtype]
        count = int(np.prod(e.shape)) if e.shape else 1
        if count * dt.itemsize != e.size:
            raise ValueError("size mismatch for %s: shape %s dtype %s vs %d bytes" % (name, e.shape, dt, e.size))
        raw = np.frombuffer(mm[e.offset:e.offset + e.size].tobytes(), dtype=dt)
        out[name] = raw.reshape(e.shape).copy()
    return out


_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def model_variables(variables_dir: str) -> Dict[str, np.ndarray]:
    """Only the model's own variables (``layer_with_weights-*``), keyed by the
    object-graph path without the ``/.ATTRIBUTES/VARIABLE_VALUE`` suffix and
    without optimizer slots."""
    out = {}
    for name, arr in load_bundle(variables_dir).items():
        if not name.endswith(_SUFFIX) or ".OPTIMIZER_SLOT" in name:
            continue
        if not name.startswith("layer_with_weights-"):
            continue
        out[name[:-len(_SUFFIX)]] = arr
    return out
