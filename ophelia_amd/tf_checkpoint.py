"""TF-free reader (and minimal writer) for TensorFlow-1 checkpoints in the "tensor bundle" format, the
counterpart of tf.train.Saver.restore / tf.train.latest_checkpoint in the reference's
restore_latest_model_parameters / restore_archived_model_parameters (synthesize.py:302-330; files written
by train.py:296-305 as {logdir}-{t2m|ssrn}/model_epoch_{E}.{index,data-00000-of-00001} plus a `checkpoint`
state file).  SURVEY.md 8f row f-1.

PARITY STATUS: pinned against an INDEPENDENT encoder, not against TensorFlow.  Neither TensorFlow nor a sample
checkpoint exists in the reference tree or in this image; tests/golden/make_tf_bundle.py assembles a bundle byte by byte
from the published formats without importing this package (LevelDB table with restart points and prefix compression
across blocks, a snappy block, two data shards, int64 global_step, Adam slots, a partitioned variable with slices, a
float16 tensor) and tests/test_tf_bundle_fixture.py reads it back with this module; tests/test_tf_checkpoint.py adds the
structural checks (round trip through this module's own writer, magic / CRC / varint handling):
  * `<prefix>.index` is a LevelDB-style SSTable: data blocks of prefix-compressed (shared, non_shared,
    value_len) entries with a restart array, each block followed by a 1-byte compression type
    (0 none / 1 snappy) and a masked CRC32C; then metaindex block, index block (separator key ->
    BlockHandle varint64 offset,size) and a 48-byte footer ending in the magic 0xdb4775248b80fb57.
  * key "" holds BundleHeaderProto {num_shards=1, endianness=2, version=3}; every other key is a variable
    name holding BundleEntryProto {dtype=1, shape=2{dim=2{size=1}}, shard_id=3, offset=4, size=5,
    crc32c=6 (fixed32)}.
  * `<prefix>.data-0000S-of-0000N` holds the raw little-endian tensor bytes at [offset, offset+size).
  * a partitioned variable: its entry lists `slices` (TensorSliceProto, field 7) and owns no bytes; every slice is a second
    entry under the key OrderedCode(0, name, rank, (start, length) per dimension) -- saved_tensor_slice_util.cc:
    EncodeTensorNameSlice -- and is copied into its place of the full tensor here.
float32 / float64 / int32 / int64 tensors are returned; any other dtype among the SELECTED variables is refused loudly
(strict=False skips it), as is anything malformed.
"""
import os
import re
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}      # tensorflow DataType enum values
_DT_INV = {np.dtype(v): k for k, v in _DT.items()}

# ------------------------------------------------------------------ CRC32C (Castagnoli), masked as in leveldb
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data, crc=0):
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / tiny protobuf codec
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message (varint, fixed32/64, length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": []}
    for f, wt, v in _pb_fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:                                   # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1: size = v3
                    e["shape"].append(size)
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
        elif f == 7:                                          # TensorSliceProto: repeated Extent extent = 1 {start = 1, length = 2 (oneof: absent = all)}
            ext = []
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    start, length = 0, None
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1: start = v3
                        elif f3 == 2: length = v3
                    ext.append((start, length))
            e["slices"].append(ext)
    return e


# ------------------------------------------------------------------ OrderedCode keys of tensor slices (ordered_code.cc)
def _oc_read_num(buf, pos):
    n = buf[pos]; pos += 1
    if n > 8:
        raise ValueError("corrupt ordered-code number in a tensor-slice key")
    return int.from_bytes(buf[pos:pos + n], "big"), pos + n


def _oc_read_string(buf, pos):
    out = bytearray()
    while True:
        ch = buf[pos]; pos += 1
        if ch == 0x00:
            nx = buf[pos]; pos += 1
            if nx == 0x01:
                return bytes(out), pos
            if nx != 0xff:
                raise ValueError("corrupt ordered-code string in a tensor-slice key")
            out.append(0x00)
        elif ch == 0xff:
            nx = buf[pos]; pos += 1
            if nx != 0x00:
                raise ValueError("corrupt ordered-code string in a tensor-slice key")
            out.append(0xff)
        else:
            out.append(ch)


def _oc_read_signed(buf, pos):
    first = buf[pos]
    neg = not (first & 0x80)
    fb = first ^ (0xff if neg else 0)
    if fb == 0xff:
        raise ValueError("tensor-slice key with a >= 8-byte extent: not supported")
    ln = 7 - ((fb ^ 0xff).bit_length() - 1)              # leading one bits of fb = length of the encoding
    x = -1 if neg else 0
    for i in range(ln):
        x = (x << 8) | buf[pos + i]
    mask = ((0xff << (8 - ln)) & 0xff) << (8 * (ln - 1))  # the header bits: 0x80, 0xc000, 0xe00000, ...
    return x ^ mask, pos + ln


def _decode_slice_key(key):
    """(variable name, [(start, length or None)]) of a key written by EncodeTensorNameSlice."""
    zero, pos = _oc_read_num(key, 0)
    if zero != 0:
        raise ValueError("not a tensor-slice key")
    name, pos = _oc_read_string(key, pos)
    rank, pos = _oc_read_num(key, pos)
    ext = []
    for _ in range(rank):
        start, pos = _oc_read_signed(key, pos)
        length, pos = _oc_read_signed(key, pos)
        ext.append((start, None if length < 0 else length))
    if pos != len(key):
        raise ValueError("trailing bytes in a tensor-slice key")
    return name.decode("utf-8"), ext


# ------------------------------------------------------------------ snappy (raw format) decompression
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                        # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = struct.unpack_from("<H", buf, pos)[0]; pos += 2
        else:
            ln = (tag >> 2) + 1
            off = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        for _ in range(ln):                                  # may overlap itself
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block")
    return bytes(out)


# ------------------------------------------------------------------ SSTable reading
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    body, ctype, crc = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if verify and mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("checkpoint index block at %d fails its CRC" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return body


def _block_entries(block):
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen]); pos += vlen


def read_index(prefix, verify=True, with_slices=False):
    """{variable name: entry dict} and the header, from <prefix>.index (with_slices: also {(name, extents): entry} of the slices
    of partitioned variables)."""
    path = prefix + ".index"
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != MAGIC:
            raise ValueError("%s is not a TensorFlow checkpoint index (bad magic)" % path)
        _, p = _get_varint(footer, 0); _, p = _get_varint(footer, p)        # metaindex handle
        ioff, p = _get_varint(footer, p); isz, p = _get_varint(footer, p)  # index handle
        entries, header, pieces = {}, None, {}
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, q = _get_varint(handle, 0); bsz, q = _get_varint(handle, q)
            for key, val in _block_entries(_read_block(f, boff, bsz, verify)):
                if key == b"":
                    header = {fn: v for fn, _, v in _pb_fields(val)}
                elif key[:1] == b"\x00":                      # a slice of a partitioned variable
                    name, ext = _decode_slice_key(key)
                    pieces[(name, tuple(ext))] = _parse_entry(val)
                else:
                    entries[key.decode("utf-8")] = _parse_entry(val)
    return (entries, header, pieces) if with_slices else (entries, header)


def read_checkpoint(prefix, scope=None, verify_data=False, strict=True):
    """{variable name: ndarray} for every variable under `scope` (e.g. 'Text2Mel/'), skipping optimizer slots ('.../Adam',
    '.../Adam_1') and bookkeeping (global_step, beta*_power).  Partitioned variables are assembled from their slices.  A
    selected variable of a dtype other than float32 / float64 / int32 / int64 raises (strict=False: it is left out)."""
    entries, header, pieces = read_index(prefix, with_slices=True)
    nshards = (header or {}).get(1, 1) or 1
    if (header or {}).get(2, 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    out, files = {}, {}

    def payload(name, e, shape):
        if e["dtype"] not in _DT:
            raise ValueError("tensor %s has dtype enum %d: only float32 / float64 / int32 / int64 are supported" % (name, e["dtype"]))
        sid = e["shard_id"]
        if not 0 <= sid < nshards:
            raise ValueError("tensor %s names shard %d of %d" % (name, sid, nshards))
        if sid not in files:
            files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb")
        files[sid].seek(e["offset"])
        raw = files[sid].read(e["size"])
        want = int(np.prod(shape, dtype=np.int64)) * np.dtype(_DT[e["dtype"]]).itemsize
        if len(raw) != e["size"] or e["size"] != want:
            raise ValueError("tensor %s: %d bytes on disk, %d in the index, %d by its shape" % (name, len(raw), e["size"], want))
        if verify_data and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("tensor %s fails its CRC" % name)
        return np.frombuffer(raw, dtype=np.dtype(_DT[e["dtype"]]).newbyteorder("<")).reshape(shape)
    try:
        for name, e in entries.items():
            if scope and not name.startswith(scope):
                continue
            if re.search(r"(/Adam(_\d+)?$)|(^global_step$)|(beta\d_power$)", name):
                continue
            if e["dtype"] not in _DT:
                if strict:
                    raise ValueError("variable %s has dtype enum %d: only float32 / float64 / int32 / int64 are supported" % (name, e["dtype"]))
                continue
            if not e["slices"]:
                out[name] = payload(name, e, e["shape"]).astype(_DT[e["dtype"]])
                continue
            full = np.zeros(e["shape"], _DT[e["dtype"]])
            seen = np.zeros(e["shape"], bool)
            for ext in e["slices"]:
                if len(ext) != len(e["shape"]):
                    raise ValueError("variable %s: a slice of rank %d for a tensor of rank %d" % (name, len(ext), len(e["shape"])))
                pe = pieces.get((name, tuple(ext)))
                if pe is None:
                    raise ValueError("variable %s: the index lists slice %s but holds no entry for it" % (name, ext))
                idx = tuple(slice(st, None if ln is None else st + ln) for st, ln in ext)
                if full[idx].shape != tuple(pe["shape"]):
                    raise ValueError("variable %s: slice %s has shape %s, its entry says %s" % (name, ext, full[idx].shape, pe["shape"]))
                full[idx] = payload("%s%s" % (name, ext), pe, pe["shape"])
                seen[idx] = True
            if not seen.all():
                raise ValueError("variable %s: its slices do not cover the tensor" % name)
            out[name] = full
    finally:
        for f in files.values():
            f.close()
    return out


def latest_checkpoint(savepath):
    """Prefix named by `model_checkpoint_path` in <savepath>/checkpoint (tf.train.latest_checkpoint), or None."""
    state = os.path.join(savepath, "checkpoint")
    if not os.path.isfile(state):
        return None
    for line in open(state):
        m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
        if m:
            p = m.group(1)
            p = p if os.path.isabs(p) else os.path.join(savepath, p)
            return p if os.path.isfile(p + ".index") else None
    return None


# ------------------------------------------------------------------ minimal writer (tests, exporting)
def _pb_varint_field(f, v):
    return _put_varint(f << 3) + _put_varint(v)


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _make_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
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


def write_checkpoint(prefix, W, block_entries=64, data_crc=True):
    """Write {name: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001 (uncompressed blocks) and
    update <dir>/checkpoint.  Used by the tests and to export weights in the reference's format.
    data_crc=False skips the per-tensor CRC32C (pure-Python, ~4 MB/s): fine for this module's reader, but
    TensorFlow verifies it, so keep the default when exporting for TensorFlow."""
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    names = sorted(W, key=lambda s: s.encode("utf-8"))
    items, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as fd:
        for n in names:
            a = np.ascontiguousarray(W[n])
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in a.shape)
            entry = (_pb_varint_field(1, _DT_INV[a.dtype]) + _pb_bytes_field(2, shape) +
                     (_pb_varint_field(4, offset) if offset else b"") + _pb_varint_field(5, len(raw)) +
                     (_put_varint((6 << 3) | 5) + struct.pack("<I", mask_crc(crc32c(raw))) if data_crc else b""))
            items.append((n.encode("utf-8"), entry))
            fd.write(raw)
            offset += len(raw)
    header = _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))      # num_shards=1, version.producer=1
    items = [(b"", header)] + items
    with open(prefix + ".index", "wb") as fi:
        pos, index_items = 0, []

        def emit(block):
            nonlocal pos
            trailer = bytes([0])
            fi.write(block + trailer + struct.pack("<I", mask_crc(crc32c(block + trailer))))
            handle = _put_varint(pos) + _put_varint(len(block))
            pos += len(block) + 5
            return handle
        for i in range(0, len(items), block_entries):
            chunk = items[i:i + block_entries]
            index_items.append((chunk[-1][0], emit(_make_block(chunk))))
        meta = emit(_make_block([]))
        index = emit(_make_block(index_items, restart_interval=1))
        footer = meta + index
        fi.write(footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as fs:
        base = os.path.basename(prefix)
        fs.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
