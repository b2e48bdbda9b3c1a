"""TFRecord shards of training examples, after WIDERDataset.to_tfrecord
(/root/reference/dataset.py:56-126).

The reference writes them with TensorFlow (tf.train.Example, tf.io.serialize_tensor,
tf.io.TFRecordWriter).  This is the same file format written without TensorFlow: the
protocol-buffer wire encoding of the three messages involved and the record framing are
stated here directly.  Host-side file code; nothing in it touches the GPU.

  record    uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
            (little endian), masked(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32
  Example   { features: Features{ feature: map<string, Feature> } }
  Feature   oneof { bytes_list = 1, float_list = 2 (packed floats), int64_list = 3 (packed) }
  Tensor    TensorProto{ dtype = 1, tensor_shape = 2 { dim = 2 { size = 1 } }, tensor_content = 4 }

Map entries are written in sorted key order (what protobuf's deterministic serialisation
does; the order carries no meaning to a parser).
"""
import struct

import numpy as np

# ---- CRC-32C (Castagnoli), reflected polynomial 0x82F63B78 ----------------------------------
_TABLE = None


def _table():
    global _TABLE
    if _TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _TABLE = [int(v) for v in t]
    return _TABLE


def _crc32c_bytewise(data):
    """One table look-up per byte, in Python: the definition, kept for short inputs and as the
    cross-check of the library routine."""
    t = _table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data):
    """CRC-32C of a bytes-like object.  Records carry megabytes of tensor data, so anything
    beyond a few bytes goes through libymx's eight-bytes-per-step host routine (ymx_crc32c)."""
    data = bytes(data)
    if len(data) < 64:
        return _crc32c_bytewise(data)
    from . import _lib
    return int(_lib.lib().ymx_crc32c(data, len(data), 0))


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protocol-buffer wire encoding ---------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1                       # negative int64 values take ten bytes
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_field(number, payload):
    return _varint((number << 3) | 2) + _varint(len(payload)) + payload


_DTYPES = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.uint8): 4,
           np.dtype(np.int64): 9, np.dtype(np.float16): 19}


def serialize_tensor(array):
    """tf.io.serialize_tensor for a numeric numpy array: a TensorProto with the array's bytes
    in tensor_content."""
    a = np.asarray(array, order="C")
    if a.dtype not in _DTYPES:
        raise ValueError("serialize_tensor: unsupported dtype %s" % a.dtype)
    shape = b"".join(_len_field(2, (_varint(1 << 3) + _varint(d)) if d else b"") for d in a.shape)
    out = _varint(1 << 3) + _varint(_DTYPES[a.dtype])
    out += _len_field(2, shape)
    if a.size:
        out += _len_field(4, a.astype(a.dtype.newbyteorder("<")).tobytes())
    return out


def bytes_feature(value):
    """tf.train.Feature(bytes_list=BytesList(value=[value])) (dataset.py:56-59)."""
    return _len_field(1, _len_field(1, bytes(value)))


def floats_feature(values):
    """tf.train.Feature(float_list=FloatList(value=values)) (dataset.py:61-62)."""
    v = np.asarray(values).astype("<f4").reshape(-1)
    return _len_field(2, _len_field(1, v.tobytes()) if v.size else b"")


def int64_feature(values):
    v = np.asarray(values).astype(np.int64).reshape(-1)
    return _len_field(3, _len_field(1, b"".join(_varint(int(x)) for x in v)) if v.size else b"")


def example(features):
    """Serialised tf.train.Example from {name: encoded Feature}."""
    body = b""
    for key in sorted(features):
        entry = _len_field(1, key.encode("utf-8")) + _len_field(2, features[key])
        body += _len_field(1, entry)
    return _len_field(1, body)


def sparse_indices(dense):
    """What ``tf.io.serialize_sparse(tf.sparse.from_dense(x))[0]`` holds (dataset.py:75-83):
    the serialised int64 index matrix [nnz, rank] of the non-zero entries, row-major order.
    (The reference keeps element [0] of the three serialised parts, i.e. the indices only.)"""
    d = np.asarray(dense)
    idx = np.argwhere(d != 0).astype(np.int64).reshape(-1, max(d.ndim, 1))
    return serialize_tensor(idx)


# ---- record files ----------------------------------------------------------------------------------
class TFRecordWriter(object):
    """Context manager with ``write(bytes)``, as tf.io.TFRecordWriter."""

    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, data):
        data = bytes(data)
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def read_records(path):
    """Yields the payload of every record of a file, checking both checksums."""
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise ValueError("truncated record header")
            (n,) = struct.unpack("<Q", head)
            (c,) = struct.unpack("<I", f.read(4))
            if c != masked_crc32c(head):
                raise ValueError("record length checksum mismatch")
            data = f.read(n)
            (c,) = struct.unpack("<I", f.read(4))
            if len(data) != n or c != masked_crc32c(data):
                raise ValueError("record data checksum mismatch")
            yield data


def _read_varint(buf, pos):
    n, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _fields(buf):
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        wt = key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            val = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            val = bytes(buf[pos:pos + 8])
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.append((key >> 3, wt, val))
    return out


def parse_tensor(buf):
    """Inverse of serialize_tensor."""
    dtype, shape, content = None, [], b""
    for num, _, val in _fields(buf):
        if num == 1:
            dtype = val
        elif num == 2:
            for n2, _, dim in _fields(val):
                if n2 == 2:
                    size = 0
                    for n3, _, v in _fields(dim):
                        if n3 == 1:
                            size = v
                    shape.append(size)
        elif num == 4:
            content = val
    np_dtype = {v: k for k, v in _DTYPES.items()}[dtype]
    return np.frombuffer(content, np_dtype.newbyteorder("<")).reshape(shape).astype(np_dtype)


def parse_example(buf):
    """{name: bytes | float32 array | int64 array} of a serialised Example (one bytes value per
    bytes feature, as the writer above produces)."""
    out = {}
    for num, _, feats in _fields(buf):
        if num != 1:
            continue
        for n2, _, entry in _fields(feats):
            if n2 != 1:
                continue
            key, feat = None, b""
            for n3, _, v in _fields(entry):
                if n3 == 1:
                    key = v.decode("utf-8")
                elif n3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat):
                vals = [v for n4, _, v in _fields(lst) if n4 == 1]
                if kind == 1:
                    out[key] = vals[0] if vals else b""
                elif kind == 2:
                    out[key] = np.frombuffer(b"".join(vals), "<f4").astype(np.float32)
                elif kind == 3:
                    ints, raw, pos = [], b"".join(vals), 0
                    while pos < len(raw):
                        v, pos = _read_varint(raw, pos)
                        ints.append(v - (1 << 64) if v >> 63 else v)
                    out[key] = np.array(ints, np.int64)
    return out


def to_tfrecord(examples, filename, shards, use_sparse=False):
    """Writes ``shards`` files ``filename.shard{k}of{shards}`` (dataset.py:64-126).

    examples: a sized iterable of (image, targets) pairs as the dataset yields them at batch
    size 1: image [1,S,S,3], targets = three arrays [1, n_i].  Each example becomes one
    tf.train.Example with the serialised image tensor and the three target vectors as float
    lists (use_sparse: the serialised indices of their non-zero entries instead, as the
    reference stores them).  Every shard holds len(examples) // shards examples; the
    remainder is dropped, as in the reference.  Returns the paths written."""
    shard_size = len(examples) // shards
    if shard_size < 1:
        raise ValueError("fewer examples than shards")
    paths, pending, shard_index = [], [], 0
    for i, (image, targets) in enumerate(examples):
        small, medium, large = (np.asarray(t)[0] for t in targets)
        if use_sparse:
            feature = {"image": bytes_feature(serialize_tensor(np.asarray(image)[0])),
                       "small_targets": bytes_feature(sparse_indices(small)),
                       "medium_targets": bytes_feature(sparse_indices(medium)),
                       "large_targets": bytes_feature(sparse_indices(large))}
        else:
            feature = {"image": bytes_feature(serialize_tensor(np.asarray(image)[0])),
                       "small_targets": floats_feature(small),
                       "medium_targets": floats_feature(medium),
                       "large_targets": floats_feature(large)}
        pending.append(example(feature))
        if (i + 1) % shard_size == 0:
            path = "%s.shard%dof%d" % (filename, shard_index + 1, shards)
            with TFRecordWriter(path) as w:
                for e in pending:
                    w.write(e)
            paths.append(path)
            pending = []
            shard_index += 1
            if shard_index == shards:
                break
    return paths
