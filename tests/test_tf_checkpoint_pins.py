"""CPU: what can be pinned of tf_checkpoint.py WITHOUT TensorFlow (the reference restores `models.ckpt-500` with tf.train.Saver,
src/adversary_autoencoder.py:42-51; no checkpoint ships with it and TF is absent here, so no TF-written file exists to read).

* CRC-32C against the published known answers (RFC 3720 B.4; the same vectors are LevelDB's util/crc32c_test.cc StandardResults) and
  the masking of LevelDB's crc32c.h (rotate right by 15, add 0xa282ead8) against values worked out by hand;
* a table file whose BYTES are written out here -- block entries with prefix-compressed keys and a restart array, block trailers,
  BundleHeaderProto / BundleEntryProto in protobuf wire format, the 48-byte footer -- following the published layouts
  (leveldb/doc/table_format.md, tensorflow/core/protobuf/tensor_bundle.proto), NOT produced by the repo's writer;
* two data shards, a Snappy-compressed block (literal + copy elements, snappy/format_description.txt), and the refusals: big-endian
  bundles, partitioned variables, corrupt checksums, truncated shards, wrong magic."""
import os
import struct

import numpy as np
import pytest

from geometric_adv_amd import tf_checkpoint as T


def crc32c_bitwise(data):
    """Independent of the module under test: the reflected Castagnoli polynomial, one bit at a time."""
    c = 0xFFFFFFFF
    for b in bytes(data):
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def test_crc32c_published_known_answers():
    # RFC 3720 appendix B.4 / leveldb util/crc32c_test.cc (StandardResults)
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    iscsi_read = bytes([0x01, 0xC0, 0x00, 0x00] + [0x00] * 12 + [0x14, 0x00, 0x00, 0x00, 0x00, 0x00, 0x04, 0x00, 0x00, 0x00, 0x00, 0x14,
                        0x00, 0x00, 0x00, 0x18, 0x28, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x02, 0x00, 0x00, 0x00] + [0x00] * 4)
    assert len(iscsi_read) == 48 and T.crc32c(iscsi_read) == 0xD9963A56
    assert T.crc32c(b"123456789") == 0xE3069283               # the catalogue's check value of CRC-32C
    assert T.crc32c(b"") == 0
    # leveldb crc32c_test.cc Extend: Value("hello world") == Extend(Value("hello "), "world")
    assert T.crc32c(b"world", T.crc32c(b"hello ")) == T.crc32c(b"hello world") == crc32c_bitwise(b"hello world")


def test_crc32c_long_inputs_take_the_vectorised_path_and_agree_with_the_bitwise_form():
    rng = np.random.default_rng(5)
    for n in (1 << 14, (1 << 14) + 1, 100003, (1 << 20) + 12345):
        buf = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(buf) == crc32c_bitwise(buf), n
    buf = rng.integers(0, 256, 70001, dtype=np.uint8).tobytes()
    assert T.crc32c(buf[30000:], T.crc32c(buf[:30000])) == crc32c_bitwise(buf)


def test_crc_mask_known_answers_by_hand():
    # leveldb util/crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    assert T.mask_crc(0x00000000) == 0xA282EAD8
    assert T.mask_crc(0xFFFFFFFF) == 0xA282EAD7                # the rotation of all ones is all ones; + delta wraps
    assert T.mask_crc(0x00008000) == 0xA282EAD9                # bit 15 rotates to bit 0
    assert T.mask_crc(0x00000001) == 0xA284EAD8                # bit 0 rotates to bit 17
    # 0xE3069283 = 1110 0011 0000 0110 1001 0010 1000 0011: >> 15 = 0x0001C60D, << 17 = 0x25060000 -> 0x2507C60D; + delta = 0xC78AB0E5
    assert T.mask_crc(0xE3069283) == 0xC78AB0E5
    for c in (0, 1, 0x8000, 0xFFFFFFFF, 0xE3069283, 0x12345678):
        assert T.unmask_crc(T.mask_crc(c)) == c
    assert T.mask_crc(T.crc32c(b"foo")) != T.crc32c(b"foo")    # crc32c_test.cc Mask


def _trailer(block, ctype=0):
    """block trailer of table_format.md: 1 byte type + 4 bytes masked crc32c(block + type), little endian -- with THIS file's
    CRC and the masking written out, not the module's."""
    c = crc32c_bitwise(block + bytes([ctype]))
    m = (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    return bytes([ctype]) + struct.pack("<I", m)


def _masked(raw):
    c = crc32c_bitwise(raw)
    return struct.pack("<I", (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


A = np.arange(6, dtype="<f4").reshape(2, 3) * np.float32(0.5) - np.float32(1.25)      # autoencoder/a : float32 (2, 3), shard 0, offset 0
B = np.array([7, -9], dtype="<i4")                                                   # autoencoder/b : int32 (2,),  shard 1, offset 8
C0 = np.array(3.5, dtype="<f8")                                                      # step : float64 scalar, shard 0, offset 24


def _hand_assembled_index(endianness=0, compress_data_block=False, with_slice=False, bad_magic=False, flip=None):
    # --- BundleHeaderProto { num_shards = 2 (field 1 varint), endianness (field 2 varint), version { producer = 1 } (field 3) }
    header = b"\x08\x02" + (b"\x10\x01" if endianness else b"") + b"\x1a\x02\x08\x01"
    # --- BundleEntryProto: dtype (1), shape (2: TensorShapeProto { dim (2) { size (1) } }), shard_id (3), offset (4), size (5),
    #     crc32c (6, fixed32), slices (7)
    ent_a = b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03" + b"\x28\x18" + b"\x35" + _masked(A.tobytes())
    ent_b = b"\x08\x03" + b"\x12\x04" + b"\x12\x02\x08\x02" + b"\x18\x01" + b"\x20\x08" + b"\x28\x08" + b"\x35" + _masked(B.tobytes())
    if with_slice:
        ent_b += b"\x3a\x02\x0a\x00"                            # one TensorSliceProto: a partitioned variable
    ent_c = b"\x08\x02" + b"\x12\x00" + b"\x20\x18" + b"\x28\x08" + b"\x35" + _masked(C0.tobytes())
    # --- data block (table_format.md / block_builder.cc): entries = varint shared, varint non_shared, varint value_len, key delta,
    #     value; restart points every 16 entries -> one restart at 0; keys sorted bytewise: "", "autoencoder/a", "autoencoder/b", "step"
    def entry(shared, delta, value):
        assert len(value) < 128
        return bytes([shared, len(delta), len(value)]) + delta + value
    block = (entry(0, b"", header) + entry(0, b"autoencoder/a", ent_a) + entry(12, b"b", ent_b) + entry(0, b"step", ent_c)
             + struct.pack("<I", 0) + struct.pack("<I", 1))
    out = bytearray()
    if compress_data_block:
        # raw Snappy: varint uncompressed length, then elements.  First 40 bytes as ONE literal (tag (len-1) << 2), then the rest as
        # literals of <= 60 bytes -- and one COPY: the bytes "autoencoder/" of the second key repeat nothing earlier, so a copy is
        # placed where the block really repeats itself: the second dim record 12 02 08 .. follows 12 02 08 .. 4 bytes earlier in
        # ent_a (copy with 1-byte offset: tag = 01 | (len-4) << 2 | (offset >> 8) << 5, then offset & 0xff)
        i = block.index(b"\x12\x02\x08\x02\x12\x02\x08\x03")
        body = bytearray()
        body += bytes([len(block)]) if len(block) < 128 else bytes([(len(block) & 0x7F) | 0x80, len(block) >> 7])
        def literal(chunk):
            assert 1 <= len(chunk) <= 60
            return bytes([(len(chunk) - 1) << 2]) + chunk
        pos = 0
        while pos < i + 4:                                   # literals up to and including the first dim record
            take = min(60, i + 4 - pos)
            body += literal(block[pos:pos + take]); pos += take
        # the next three bytes 12 02 08 are a copy of the three bytes 4 back ... but a kind-1 copy is >= 4 bytes long: copy
        # "12 02 08" + whatever follows 4 back would be wrong, so copy exactly 4 bytes only if they match; otherwise fall back
        if block[pos:pos + 4] == block[pos - 4:pos]:
            body += bytes([0x01 | ((4 - 4) << 2) | (0 << 5), 4]); pos += 4
        else:                                                # (12 02 08 03 differs in its last byte): copy via a kind-2 element of length 3
            body += bytes([0x02 | ((3 - 1) << 2), 4, 0]); pos += 3
        while pos < len(block):
            take = min(60, len(block) - pos)
            body += literal(block[pos:pos + take]); pos += take
        stored, ctype = bytes(body), 1
    else:
        stored, ctype = block, 0
    data_off, data_size = 0, len(stored)
    out += stored + _trailer(stored, ctype)
    # --- metaindex block: empty (one restart)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)
    meta_off = len(out)
    out += meta + _trailer(meta)
    # --- index block: one entry, key >= last key of the data block ("step"), value = BlockHandle(varint offset, varint size)
    handle = bytes([data_off, data_size]) if data_size < 128 else bytes([data_off, (data_size & 0x7F) | 0x80, data_size >> 7])
    index = bytes([0, 4, len(handle)]) + b"step" + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    index_off = len(out)
    out += index + _trailer(index)
    # --- footer: metaindex handle, index handle, padding to 40 bytes, magic 0xdb4775248b80fb57 little endian
    def h(off, size):
        def v(x):
            return bytes([x]) if x < 128 else bytes([(x & 0x7F) | 0x80, x >> 7])
        return v(off) + v(size)
    footer = h(meta_off, len(meta)) + h(index_off, len(index))
    footer += b"\x00" * (40 - len(footer)) + (b"\x57\xfb\x80\x8b\x24\x75\x47\xdb" if not bad_magic else b"\x57\xfb\x80\x8b\x24\x75\x47\xdc")
    out += footer
    if flip is not None:
        out[flip] ^= 0x40
    return bytes(out)


def _write(tmp_path, **kw):
    prefix = str(tmp_path / "models.ckpt-500")
    with open(prefix + ".index", "wb") as f:
        f.write(_hand_assembled_index(**kw))
    with open(prefix + ".data-00000-of-00002", "wb") as f:
        f.write(A.tobytes() + C0.tobytes())
    with open(prefix + ".data-00001-of-00002", "wb") as f:
        f.write(b"\xee" * 8 + B.tobytes())
    return prefix


@pytest.mark.parametrize("compressed", [False, True])
def test_reads_a_hand_assembled_two_shard_bundle(tmp_path, compressed):
    prefix = _write(tmp_path, compress_data_block=compressed)
    header, entries = T.read_index(prefix)
    assert header["num_shards"] == 2 and header["endianness"] == 0
    assert sorted(entries) == ["autoencoder/a", "autoencoder/b", "step"]
    assert entries["autoencoder/a"]["shape"] == (2, 3) and entries["autoencoder/b"]["shard_id"] == 1 and entries["step"]["shape"] == ()
    got = T.load_checkpoint(prefix)
    assert got["autoencoder/a"].dtype == np.float32 and np.array_equal(got["autoencoder/a"], A)
    assert got["autoencoder/b"].dtype == np.int32 and np.array_equal(got["autoencoder/b"], B)
    assert got["step"].dtype == np.float64 and got["step"].shape == () and got["step"] == 3.5
    assert T.list_variables(prefix) == [("autoencoder/a", (2, 3)), ("autoencoder/b", (2,)), ("step", ())]
    only = T.load_checkpoint(prefix, lambda n: n.startswith("autoencoder"))   # restore_ae_model's filter (adversary_autoencoder.py:45-47)
    assert sorted(only) == ["autoencoder/a", "autoencoder/b"]


def test_refusals(tmp_path):
    with pytest.raises(ValueError, match="big-endian"):
        T.read_index(_write(tmp_path, endianness=1))
    with pytest.raises(ValueError, match="partitioned"):
        T.load_checkpoint(_write(tmp_path, with_slice=True))
    with pytest.raises(ValueError, match="bad table magic"):
        T.read_index(_write(tmp_path, bad_magic=True))
    with pytest.raises(ValueError, match="checksum"):
        T.read_index(_write(tmp_path, flip=20))                 # a byte of the data block: its trailer no longer matches
    T.read_index(_write(tmp_path, flip=20), verify=False)       # (the corruption sits in a key: unverified reading still parses)
    prefix = _write(tmp_path)
    with open(prefix + ".data-00001-of-00002", "r+b") as f:     # a flipped tensor byte: the per-tensor checksum catches it
        f.seek(9); f.write(b"\x5a")
    with pytest.raises(ValueError, match="tensor checksum"):
        T.load_checkpoint(prefix)
    prefix = _write(tmp_path)
    with open(prefix + ".data-00000-of-00002", "r+b") as f:
        f.truncate(28)
    with pytest.raises(ValueError, match="truncated"):
        T.load_checkpoint(prefix)
    os.remove(prefix + ".data-00001-of-00002")
    with pytest.raises(OSError):
        T.load_checkpoint(prefix, lambda n: n == "autoencoder/b")


def test_writer_output_is_what_the_hand_assembled_reader_rules_expect(tmp_path):
    """The repo's writer against the same independent pieces: its block trailers and tensor checksums verify under THIS file's CRC,
    its footer carries the published magic, and its header entry says one little-endian shard."""
    prefix = str(tmp_path / "w.ckpt-1")
    T.write_checkpoint(prefix, {"autoencoder/a": A, "autoencoder/b": B, "step": C0})
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == b"\x57\xfb\x80\x8b\x24\x75\x47\xdb"
    header, entries = T.read_index(prefix)
    assert header["num_shards"] == 1 and header["endianness"] == 0
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    for name, arr in (("autoencoder/a", A), ("autoencoder/b", B), ("step", C0)):
        e = entries[name]
        piece = data[e["offset"]:e["offset"] + e["size"]]
        assert piece == arr.tobytes() and struct.pack("<I", e["crc32c"]) == _masked(piece)
    # first block: starts at 0; find its size from the index block the footer points to
    _, p = T._get_varint(raw[-48:], 0); _, p = T._get_varint(raw[-48:], p)
    ioff, p = T._get_varint(raw[-48:], p); isize, p = T._get_varint(raw[-48:], p)
    assert raw[ioff + isize:ioff + isize + 5] == _trailer(raw[ioff:ioff + isize])
