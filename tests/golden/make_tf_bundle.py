#!/usr/bin/env python
"""Builds tests/golden/tf_bundle/*: a TensorFlow-1 "tensor bundle" checkpoint assembled byte by byte from the PUBLISHED formats,
by an encoder that shares no code with ophelia_amd/tf_checkpoint.py (it does not import the package at all) -- the pin of
SURVEY.md 8f row f-1's reader that TensorFlow itself cannot provide here (no TensorFlow in this image, no sample checkpoint in
the reference tree).  What train.py:296-305 (`tf.train.Saver().save`) leaves behind and synthesize.py:302-330 restores from:

    model_epoch_3.index                 a LevelDB table (tensorflow/core/lib/io/table_format.txt, leveldb doc/table_format.md):
                                          data blocks of prefix-compressed entries with restart points every 16 entries, block trailer =
                                          1 type byte + masked CRC32C, an (empty) metaindex block, an index block of shortest separators
                                          -> BlockHandle, a 48-byte footer with the magic 0xdb4775248b80fb57
                                        key ""  -> BundleHeaderProto, key <name> -> BundleEntryProto (tensor_bundle.proto),
                                        key OrderedCode(0, name, slice) -> the entry of one slice of a partitioned variable
                                          (tensorflow/core/util/saved_tensor_slice_util.cc: EncodeTensorNameSlice)
    model_epoch_3.data-0000S-of-00002   raw little-endian tensor bytes of shard S
    checkpoint                          CheckpointState text proto

Deliberately included: long shared key prefixes across restart points and across blocks, ONE snappy-compressed data block (type
byte 1; TensorFlow's BundleWriter never compresses, LevelDB readers must cope), two data shards, `global_step` (int64 scalar),
Adam slots and beta powers, a variable partitioned into two slices along axis 0, a float16 tensor outside the model scopes.
Everything the reader must return is also saved as tf_bundle_expected.npz.

    python tests/golden/make_tf_bundle.py          # regenerates the fixture (deterministic)
"""
import os
import struct

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle")
MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64, DT_HALF = 1, 3, 9, 19          # tensorflow/core/framework/types.proto


# ---------------------------------------------------------------- primitives (own implementations)
def crc32c_bitwise(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bit at a time."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            lsb = crc & 1
            crc >>= 1
            if lsb:
                crc ^= 0x82F63B78
    return crc ^ 0xFFFFFFFF


def masked(crc):
    """leveldb / tensorflow crc32c::Mask: rotate right by 15, add a constant."""
    rot = ((crc >> 15) | (crc << 17)) & 0xFFFFFFFF
    return (rot + 0xa282ead8) & 0xFFFFFFFF


def varint(n):
    assert n >= 0
    out = []
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def pb_key(field, wire):
    return varint((field << 3) | wire)


def pb_int(field, value):                       # int32 / int64 / enum (negative int64: 10-byte two's complement)
    return pb_key(field, 0) + varint(value & 0xFFFFFFFFFFFFFFFF if value < 0 else value)


def pb_bytes(field, payload):
    return pb_key(field, 2) + varint(len(payload)) + payload


def pb_fixed32(field, value):
    return pb_key(field, 5) + struct.pack("<I", value)


def shape_proto(shape):                         # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; } }
    return b"".join(pb_bytes(2, pb_int(1, d)) for d in shape)


def slice_proto(extents):                       # TensorSliceProto { repeated Extent extent = 1 { int64 start = 1; int64 length = 2; } }
    out = b""
    for start, length in extents:
        ext = b""
        if start:
            ext += pb_int(1, start)
        if length is not None:                  # the oneof is absent for a full extent
            ext += pb_int(2, length)
        out += pb_bytes(1, ext)
    return out


def entry_proto(dtype, shape, shard, offset, size, crc, slices=()):
    out = pb_int(1, dtype) + pb_bytes(2, shape_proto(shape))
    if shard:
        out += pb_int(3, shard)
    if offset:
        out += pb_int(4, offset)
    if size:
        out += pb_int(5, size)
    if crc is not None:
        out += pb_fixed32(6, crc)
    for sl in slices:
        out += pb_bytes(7, slice_proto(sl))
    return out


# ---------------------------------------------------------------- OrderedCode (tensorflow/core/lib/strings/ordered_code.cc)
def oc_num_increasing(n):
    body = b""
    while n > 0:
        body = bytes([n & 0xFF]) + body
        n >>= 8
    return bytes([len(body)]) + body


def oc_string(s):
    out = bytearray()
    for ch in s:
        if ch == 0x00:
            out += b"\x00\xff"
        elif ch == 0xff:
            out += b"\xff\x00"
        else:
            out.append(ch)
    return bytes(out) + b"\x00\x01"


def oc_signed_increasing(v):
    assert -64 <= v < 64, "the fixture only needs the one-byte form"
    return bytes([(0x80 + v) & 0xFF])


def slice_key(name, extents):
    """EncodeTensorNameSlice: 0, name, rank, then (start, length) per dimension; a full extent is (0, -1)."""
    out = oc_num_increasing(0) + oc_string(name.encode("utf-8")) + oc_num_increasing(len(extents))
    for start, length in extents:
        out += oc_signed_increasing(start) + oc_signed_increasing(-1 if length is None else length)
    return out


# ---------------------------------------------------------------- snappy (format_description.txt), a plain greedy compressor
def snappy_compress(data):
    out = bytearray(varint(len(data)))
    table = {}
    i = lit = 0

    def flush(upto):
        nonlocal lit
        while lit < upto:
            n = min(upto - lit, 60)             # literals of <= 60 bytes: length - 1 in the tag's upper six bits
            out.append((n - 1) << 2)
            out.extend(data[lit:lit + n])
            lit += n

    while i + 4 <= len(data):
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and 0 < i - j < 65536:
            n = 4
            while i + n < len(data) and data[j + n] == data[i + n] and n < 64:
                n += 1
            flush(i)
            out.append(((n - 1) << 2) | 2)      # copy with a 2-byte offset
            out.extend(struct.pack("<H", i - j))
            i += n
            lit = i
        else:
            i += 1
    flush(len(data))
    return bytes(out)


# ---------------------------------------------------------------- LevelDB table
def build_block(items, restart_interval):
    out, restarts, prev = bytearray(), [], b""
    for n, (key, value) in enumerate(items):
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            lim = min(len(prev), len(key))
            while shared < lim and prev[shared] == key[shared]:
                shared += 1
        out += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        prev = key
    if not restarts:
        restarts.append(0)
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def shortest_separator(a, b):
    """BytewiseComparator::FindShortestSeparator: a <= result < b."""
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    if n < min(len(a), len(b)) and a[n] < 0xff and a[n] + 1 < b[n]:
        return a[:n] + bytes([a[n] + 1])
    return a


def short_successor(a):
    for n, ch in enumerate(a):
        if ch != 0xff:
            return a[:n] + bytes([ch + 1])
    return a


def write_table(path, items, per_block, compress_block):
    items = sorted(items, key=lambda kv: kv[0])
    blocks = [items[i:i + per_block] for i in range(0, len(items), per_block)]
    handles = []
    with open(path, "wb") as f:
        pos = 0

        def put(contents, ctype):
            nonlocal pos
            trailer = bytes([ctype])
            f.write(contents + trailer + struct.pack("<I", masked(crc32c_bitwise(contents + trailer))))
            h = varint(pos) + varint(len(contents))
            pos += len(contents) + 5
            return h
        for n, chunk in enumerate(blocks):
            raw = build_block(chunk, 16)
            if n == compress_block:
                comp = snappy_compress(raw)
                assert len(comp) < len(raw)
                handles.append(put(comp, 1))
            else:
                handles.append(put(raw, 0))
        index = []
        for n, chunk in enumerate(blocks):
            last = chunk[-1][0]
            sep = shortest_separator(last, blocks[n + 1][0][0]) if n + 1 < len(blocks) else short_successor(last)
            index.append((sep, handles[n]))
        meta = put(build_block([], 16), 0)
        idx = put(build_block(index, 1), 0)
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))


# ---------------------------------------------------------------- the checkpoint
def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(20260929))
    variables = {}                               # name -> array, in the names the reference graph creates
    for i in range(4, 12):
        variables["Text2Mel/TextEnc/HC_%d/conv1d/kernel" % i] = rng.standard_normal((3, 6, 12)).astype("<f4")
        variables["Text2Mel/TextEnc/HC_%d/conv1d/bias" % i] = rng.standard_normal(12).astype("<f4")
        for half in ("H1", "H2"):
            variables["Text2Mel/TextEnc/HC_%d/%s/gamma" % (i, half)] = (1 + 0.05 * rng.standard_normal(6)).astype("<f4")
            variables["Text2Mel/TextEnc/HC_%d/%s/beta" % (i, half)] = (0.05 * rng.standard_normal(6)).astype("<f4")
    variables["Text2Mel/AudioDec/C_1/conv1d/kernel"] = rng.standard_normal((1, 12, 6)).astype("<f4")
    variables["SSRN/D_4/conv2d_transpose/kernel"] = rng.standard_normal((1, 3, 5, 5)).astype("<f4")
    variables["SSRN/D_4/conv2d_transpose/bias"] = rng.standard_normal(5).astype("<f4")
    variables["SSRN/C_1/normalize/gamma"] = np.ones(5, "<f4")
    table = rng.standard_normal((8, 6)).astype("<f4")           # embed_1's table, saved as two partitions along axis 0
    slots = {}
    for name, arr in list(variables.items()):
        if name.endswith("kernel"):
            slots[name + "/Adam"] = (0.01 * rng.standard_normal(arr.shape)).astype("<f4")
            slots[name + "/Adam_1"] = (0.001 * rng.random(arr.shape)).astype("<f4")
    bookkeeping = {"global_step": np.array(123456, "<i8"), "beta1_power": np.array(0.5 ** 7, "<f4"), "beta2_power": np.array(0.9 ** 7, "<f4"),
                   "Text2Mel/TextEnc/alignment_lengths": np.arange(5, dtype="<i4")}
    other = {"Other/half_precision_var": rng.standard_normal((2, 3)).astype("<f2")}

    shards = [bytearray(), bytearray()]
    items = [(b"", pb_int(1, 2) + pb_bytes(3, pb_int(1, 1)))]      # BundleHeaderProto: num_shards = 2, (endianness LITTLE = 0: proto3 default, not serialised), version.producer = 1

    def store(arr, shard):
        raw = arr.tobytes()
        off = len(shards[shard])
        shards[shard] += raw
        return off, len(raw), masked(crc32c_bitwise(raw))

    dtype_of = {"<f4": DT_FLOAT, "<i4": DT_INT32, "<i8": DT_INT64, "<f2": DT_HALF}
    everything = {}
    everything.update(variables); everything.update(slots); everything.update(bookkeeping); everything.update(other)
    for n, name in enumerate(sorted(everything)):
        arr = everything[name]
        shard = n % 2                              # alternate the shards
        off, size, crc = store(arr, shard)
        items.append((name.encode("utf-8"), entry_proto(dtype_of[arr.dtype.str], arr.shape, shard, off, size, crc)))
    # the partitioned variable: the full-name entry lists the slices and owns no bytes; each slice has its own entry
    pname = "Text2Mel/TextEnc/embed_1/lookup_table"
    parts = [((0, 5), (0, None)), ((5, 3), (0, None))]
    items.append((pname.encode("utf-8"), entry_proto(DT_FLOAT, table.shape, 0, 0, 0, None, slices=parts)))
    for k, ext in enumerate(parts):
        piece = np.ascontiguousarray(table[ext[0][0]:ext[0][0] + ext[0][1]])
        off, size, crc = store(piece, 1 - k)
        items.append((slice_key(pname, ext), entry_proto(DT_FLOAT, piece.shape, 1 - k, off, size, crc)))

    prefix = os.path.join(OUT, "model_epoch_3")
    write_table(prefix + ".index", items, per_block=19, compress_block=2)
    for s, blob in enumerate(shards):
        with open("%s.data-%05d-of-%05d" % (prefix, s, len(shards)), "wb") as f:
            f.write(bytes(blob))
    with open(os.path.join(OUT, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "model_epoch_3"\nall_model_checkpoint_paths: "model_epoch_1"\nall_model_checkpoint_paths: "model_epoch_3"\n')
    expected = dict(variables)
    expected[pname] = table
    expected["Text2Mel/TextEnc/alignment_lengths"] = bookkeeping["Text2Mel/TextEnc/alignment_lengths"]
    np.savez(os.path.join(os.path.dirname(OUT), "tf_bundle_expected.npz"), **{k.replace("/", "|"): v for k, v in expected.items()})
    print("wrote", prefix, "with", len(items), "index entries;", [len(b) for b in shards], "data bytes")


if __name__ == "__main__":
    main()
