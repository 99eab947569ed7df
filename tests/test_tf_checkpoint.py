"""TF-free TF-1 checkpoint ("tensor bundle") reader.  PARITY UNPINNED: no TensorFlow and no sample
checkpoint exist here, so these tests check the format handling structurally (round trip through the
module's own writer, prefix compression across blocks, CRCs, magic, snappy blocks, state file,
optimizer-slot filtering) -- not against bytes produced by TensorFlow."""
import os
import struct

import numpy as np
import pytest

from ophelia_amd import tf_checkpoint as T


def _weights():
    rng = np.random.default_rng(0)
    W = {}
    for i in range(40):                     # > one index block, long shared prefixes
        W["Text2Mel/TextEnc/HC_%d/conv1d/kernel" % i] = rng.standard_normal((3, 8, 16)).astype(np.float32)
        W["Text2Mel/TextEnc/HC_%d/conv1d/bias" % i] = rng.standard_normal(16).astype(np.float32)
        W["Text2Mel/TextEnc/HC_%d/conv1d/kernel/Adam" % i] = np.zeros((3, 8, 16), np.float32)
        W["Text2Mel/TextEnc/HC_%d/conv1d/kernel/Adam_1" % i] = np.zeros((3, 8, 16), np.float32)
    W["SSRN/D_4/conv2d_transpose/kernel"] = rng.standard_normal((1, 3, 4, 4)).astype(np.float32)
    W["global_step"] = np.array(1234, np.int64)
    W["beta1_power"] = np.array(0.5, np.float32)
    return W


def test_roundtrip_and_filtering(tmp_path):
    W = _weights()
    prefix = str(tmp_path / "train-t2m" / "model_epoch_7")
    T.write_checkpoint(prefix, W, block_entries=16)
    entries, header = T.read_index(prefix)
    assert set(entries) == set(W) and header[1] == 1
    got = T.read_checkpoint(prefix, verify_data=True)
    keep = {n for n in W if "Adam" not in n and n != "global_step" and not n.endswith("_power")}
    assert set(got) == keep
    for n in keep:
        assert got[n].dtype == W[n].dtype and got[n].shape == W[n].shape and np.array_equal(got[n], W[n])
    assert set(T.read_checkpoint(prefix, scope="SSRN/")) == {"SSRN/D_4/conv2d_transpose/kernel"}
    assert T.latest_checkpoint(str(tmp_path / "train-t2m")) == prefix
    assert T.latest_checkpoint(str(tmp_path / "nope")) is None


def test_bad_magic_and_crc(tmp_path):
    prefix = str(tmp_path / "m")
    T.write_checkpoint(prefix, {"a/b": np.arange(6, dtype=np.float32).reshape(2, 3)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    bad = bytearray(raw); bad[-1] ^= 0xFF
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="bad magic"):
        T.read_index(prefix)
    bad = bytearray(raw); bad[3] ^= 0x01            # flip a bit inside the first data block
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="CRC"):
        T.read_index(prefix)
    open(prefix + ".index", "wb").write(raw)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); data[0] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    with pytest.raises(ValueError, match="fails its CRC"):
        T.read_checkpoint(prefix, verify_data=True)


def test_crc32c_known_answers():
    # RFC 3720 test vectors for CRC32C
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"123456789") == 0xE3069283


def test_snappy_block_decoding():
    # literal "abcd" + copy(offset 4, len 8) => "abcdabcdabcd"; then 2-byte-offset copy
    stream = bytes([12]) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert T._snappy_decompress(stream) == b"abcdabcdabcd"
    stream = bytes([8]) + bytes([3 << 2]) + b"wxyz" + bytes([((4 - 1) << 2) | 2]) + struct.pack("<H", 4)
    assert T._snappy_decompress(stream) == b"wxyzwxyz"


def test_restore_functions_prefer_tf_bundle(tmp_path):
    from ophelia_amd import architectures as A

    class Sess:
        def __init__(self): self.W = {}
        def inventory(self, scope=None): return [("Text2Mel/A/conv1d/kernel", (1, 2, 3)), ("Text2Mel/A/conv1d/bias", (3,))]
        def assign(self, W): self.W.update(W)

    class hp: logdir = str(tmp_path / "train")
    W = {"Text2Mel/A/conv1d/kernel": np.ones((1, 2, 3), np.float32), "Text2Mel/A/conv1d/bias": np.zeros(3, np.float32),
         "Text2Mel/A/conv1d/kernel/Adam": np.zeros((1, 2, 3), np.float32)}
    T.write_checkpoint(hp.logdir + "-t2m/model_epoch_12", W)
    s = Sess()
    assert A.restore_latest_model_parameters(s, hp, "t2m") == "12"
    assert set(s.W) == {"Text2Mel/A/conv1d/kernel", "Text2Mel/A/conv1d/bias"}
    T.write_checkpoint(hp.logdir + "-t2m/archive/model_epoch_5", W)
    s = Sess(); A.restore_archived_model_parameters(s, hp, "t2m", 5)
    assert np.array_equal(s.W["Text2Mel/A/conv1d/kernel"], W["Text2Mel/A/conv1d/kernel"])
    del W["Text2Mel/A/conv1d/bias"]
    T.write_checkpoint(hp.logdir + "-t2m/archive/model_epoch_6", W)
    with pytest.raises(SystemExit, match="lacks 1 variables"):
        A.restore_archived_model_parameters(Sess(), hp, "t2m", 6)
