"""TensorFlow-free reader (and writer) of the TF "V2" checkpoint format (tensor bundle).

The reference restores the victim auto-encoder with `tf.train.Saver.restore(sess, 'models.ckpt-500')`
(src/adversary_autoencoder.py:42-51, src/neural_net.py:10,33-36).  TensorFlow is not available on the
MI355X image, so `restore_ae_model` here reads the two files of the bundle directly:

    <prefix>.index                  an SSTable in the LevelDB table format: sorted (key, value) pairs,
                                    key "" -> BundleHeaderProto, key <variable name> -> BundleEntryProto
    <prefix>.data-SSSSS-of-NNNNN    the raw little-endian tensor bytes, addressed by (shard, offset, size)

Restated from the published format (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc},
tensorflow/core/lib/io/{table,block,format}.cc, tensorflow/core/protobuf/tensor_bundle.proto at the
pinned tensorflow-gpu==1.13.2, requirements.txt:107).  No checkpoint ships with the reference
(download_models_and_data.sh fetches them), so this reader is **unpinned against a TF-written file**.
What can be pinned without TensorFlow is (tests/test_tf_checkpoint_pins.py): CRC-32C and its masking against the
published known answers (RFC 3720 B.4 = LevelDB's crc32c_test.cc), and the reader against a two-shard bundle whose
index BYTES are written out in the test from the published layouts (prefix-compressed keys with a restart array,
block trailers, protobuf wire format of BundleHeaderProto / BundleEntryProto, footer, a Snappy block with literal and
copy elements) -- not produced by `write_checkpoint` below, which the other tests round-trip.
Both the uncompressed blocks TF's BundleWriter emits and Snappy-compressed blocks are accepted.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48                      # 2 block handles padded to 40 bytes + 8 bytes of magic
BLOCK_TRAILER = 5                    # 1 byte compression type + 4 bytes masked crc32c
MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
          10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_ENUM = {np.dtype(v): k for k, v in DTYPES.items()}


# ---------------------------------------------------------------- crc32c (Castagnoli), table driven
def _make_table():
    tab = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab[i] = c
    return tab


_TAB = _make_table()
_TAB_L = [int(v) for v in _TAB]


def _crc_scalar(buf, c):
    tab = _TAB_L
    for b in bytes(buf):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object.  Long inputs are cut into K equal pieces whose registers advance in
    lockstep as one numpy vector; the pieces are then chained with the (GF(2)-linear) operator "advance the
    register over L zero bytes", tabulated as four 256-entry tables: state' = Z_L(state) ^ R(0, piece)."""
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    c = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    n = len(buf)
    if n < 1 << 14:
        return _crc_scalar(buf, c) ^ 0xFFFFFFFF
    K = 4096 if n >= 1 << 20 else 256
    L = n // K
    cols = np.ascontiguousarray(np.frombuffer(buf, np.uint8, K * L).reshape(K, L).T)
    r = np.zeros(K, np.uint32)
    z = np.array([b << (8 * j) for j in range(4) for b in range(256)], np.uint32)
    for j in range(L):
        r = _TAB[(r ^ cols[j]) & 0xFF] ^ (r >> 8)
        z = _TAB[z & 0xFF] ^ (z >> 8)
    T0, T1, T2, T3 = ([int(v) for v in z[256 * j:256 * (j + 1)]] for j in range(4))
    for piece in r.tolist():
        c = T0[c & 0xFF] ^ T1[(c >> 8) & 0xFF] ^ T2[(c >> 16) & 0xFF] ^ T3[c >> 24] ^ piece
    return _crc_scalar(buf[K * L:], c) ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf wire format
def _get_varint(buf, pos):
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise ValueError("truncated length-delimited field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf):
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:                                   # TensorShapeProto.dim
            size = 0
            for g, _, u in _proto_fields(v):
                if g == 1:
                    size = _signed64(u)
            dims.append(size)
        elif f == 3 and v:                           # unknown_rank
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, _, v in _proto_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] += 1
    return e


def _parse_header(buf):
    h = {"num_shards": 0, "endianness": 0, "version": None}
    for f, _, v in _proto_fields(buf):
        if f == 1:
            h["num_shards"] = v
        elif f == 2:
            h["endianness"] = v
    return h


# ---------------------------------------------------------------- snappy (raw format) decompressor
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                          # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---------------------------------------------------------------- table reading
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size + BLOCK_TRAILER]
    if len(raw) != size + BLOCK_TRAILER:
        raise ValueError("block handle points outside the index file")
    body, ctype = raw[:size], raw[size]
    if verify:
        want = unmask_crc(struct.unpack_from("<I", raw, size + 1)[0])
        if crc32c(raw[:size + 1]) != want:
            raise ValueError("index block checksum mismatch")
    if ctype == 0:
        return body
    if ctype == 1:
        return _snappy_decompress(body)
    raise ValueError("unknown block compression type %d" % ctype)


def _block_entries(block):
    """(key, value) pairs of one table block (prefix-compressed keys; the restart array is skipped)."""
    if len(block) < 4:
        raise ValueError("table block too short")
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    if end < 0:
        raise ValueError("corrupt restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("corrupt table entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_index(prefix, verify=True):
    """-> (header dict, {variable name: entry dict}) of `<prefix>.index`."""
    path = prefix + ".index"
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < FOOTER_LEN:
        raise ValueError("%s: too short for a table footer" % path)
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise ValueError("%s: not a TF V2 checkpoint index (bad table magic)" % path)
    _, p = _get_varint(footer, 0)                     # metaindex handle (unused by the bundle)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b"":
                header = _parse_header(value)
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise ValueError("%s: no bundle header entry" % path)
    if header["endianness"] != 0:
        raise ValueError("big-endian bundles are not supported")
    return header, entries


def shard_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_variables(prefix):
    """[(name, shape)] like tf.train.list_variables."""
    _, entries = read_index(prefix)
    return [(k, e["shape"]) for k, e in sorted(entries.items())]


def load_checkpoint(prefix, name_filter=None, verify=True):
    """All (or the filtered) variables of a V2 checkpoint as {name: numpy array}."""
    header, entries = read_index(prefix, verify)
    out, files = {}, {}
    try:
        for name, e in sorted(entries.items()):
            if name_filter is not None and not name_filter(name):
                continue
            if e["slices"]:
                raise ValueError("%s: partitioned variables are not supported" % name)
            if e["dtype"] not in DTYPES:
                raise ValueError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
            dt = np.dtype(DTYPES[e["dtype"]])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if count * dt.itemsize != e["size"]:
                raise ValueError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open(shard_path(prefix, sid, header["num_shards"]), "rb")
            f = files[sid]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("%s: data shard truncated" % name)
            if verify and e["crc32c"] is not None and crc32c(raw) != unmask_crc(e["crc32c"]):
                raise ValueError("%s: tensor checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def restore_ae_weights(model_path, epoch, ae_name="autoencoder", saver_id="models.ckpt", verify=True):
    """The variables `restore_ae_model` picks (src/adversary_autoencoder.py:42-51: names starting with the
    AE name) from `<model_path>/models.ckpt-<epoch>`, Adam slots and step counters dropped."""
    from . import weights as W
    prefix = os.path.join(model_path, "%s-%d" % (saver_id, int(epoch)))
    wanted = set(W.variable_names(ae_name))
    got = load_checkpoint(prefix, lambda n: n in wanted, verify)
    missing = sorted(wanted - set(got))
    if missing:
        raise KeyError("checkpoint %s lacks %d variables, e.g. %s" % (prefix, len(missing), missing[0]))
    return got


# ---------------------------------------------------------------- writer (tests, and exporting trained weights)
def _entry_proto(dtype_enum, shape, shard, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    out = b"\x08" + _put_varint(dtype_enum) + b"\x12" + _put_varint(len(dims)) + dims
    if shard:
        out += b"\x18" + _put_varint(shard)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


def _build_block(pairs, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_checkpoint(prefix, variables, block_size=4096):
    """Writes {name: array} as a single-shard V2 bundle that `load_checkpoint` (and TF's BundleReader,
    by the published layout) reads back."""
    names = sorted(variables, key=lambda s: s.encode("utf-8"))
    data_path = shard_path(prefix, 0, 1)
    pairs = [(b"", b"\x08\x01\x1a\x02\x08\x01")]      # num_shards = 1, little endian, version { producer: 1 }
    offset = 0
    with open(data_path, "wb") as f:
        for n in names:
            a = np.asarray(variables[n], order="C")                    # (ascontiguousarray would turn 0-d into 1-d)
            raw = a.tobytes()
            f.write(raw)
            pairs.append((n.encode("utf-8"), _entry_proto(DTYPE_ENUM[a.dtype], a.shape, 0, offset, len(raw),
                                                          mask_crc(crc32c(raw)))))
            offset += len(raw)
    file_bytes, index_pairs = bytearray(), []

    def emit(block):
        off = len(file_bytes)
        file_bytes.extend(block + b"\x00")
        file_bytes.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    cur, cur_bytes = [], 0
    for k, v in pairs:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index_pairs.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index_pairs.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_pairs, restart_interval=1))
    footer = meta + index
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(file_bytes) + footer)
