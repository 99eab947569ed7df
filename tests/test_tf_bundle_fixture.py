"""the TF-free checkpoint reader against a tensor bundle that was NOT written by this package
(tests/golden/tf_bundle/, assembled byte by byte by tests/golden/make_tf_bundle.py from the published LevelDB-table /
tensor_bundle.proto / OrderedCode / snappy formats with its own CRC, varint, protobuf and block code).  TensorFlow itself is not
available here, so this is the strongest pin there is: two independent readings of the same public formats must agree."""
import os
import shutil

import numpy as np
import pytest

from ophelia_amd import tf_checkpoint as T

HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(HERE, "golden", "tf_bundle")
PREFIX = os.path.join(BUNDLE, "model_epoch_3")


def _expected():
    return {k.replace("|", "/"): v for k, v in np.load(os.path.join(HERE, "golden", "tf_bundle_expected.npz")).items()}


def test_generator_shares_no_code_with_the_reader():
    src = open(os.path.join(HERE, "golden", "make_tf_bundle.py")).read()
    assert "ophelia_amd" not in src.replace("ophelia_amd/tf_checkpoint.py", "") and "tf_checkpoint" not in src.replace("ophelia_amd/tf_checkpoint.py", "")


def test_index_structure():
    entries, header, pieces = T.read_index(PREFIX, with_slices=True)
    assert header[1] == 2                                        # num_shards
    assert "global_step" in entries and entries["global_step"]["dtype"] == 9 and entries["global_step"]["shape"] == []
    assert {e["shard_id"] for e in entries.values()} == {0, 1}
    part = entries["Text2Mel/TextEnc/embed_1/lookup_table"]
    assert part["slices"] == [[(0, 5), (0, None)], [(5, 3), (0, None)]] and part["size"] == 0
    assert set(pieces) == {("Text2Mel/TextEnc/embed_1/lookup_table", ((0, 5), (0, None))),
                           ("Text2Mel/TextEnc/embed_1/lookup_table", ((5, 3), (0, None)))}
    # one block of the table is snappy-compressed (type byte 1): find it from the footer the way a LevelDB reader does
    raw = open(PREFIX + ".index", "rb").read()
    _, p = T._get_varint(raw[-48:], 0); _, p = T._get_varint(raw[-48:], p)
    ioff, p = T._get_varint(raw[-48:], p); isz, p = T._get_varint(raw[-48:], p)
    types = []
    for _, handle in T._block_entries(raw[ioff:ioff + isz]):
        boff, q = T._get_varint(handle, 0); bsz, q = T._get_varint(handle, q)
        types.append(raw[boff + bsz])
    assert sorted(set(types)) == [0, 1] and types.count(1) == 1 and len(types) >= 4


@pytest.mark.parametrize("scope", ["Text2Mel/", "SSRN/"])
def test_reader_returns_the_generators_arrays(scope):
    exp = {k: v for k, v in _expected().items() if k.startswith(scope)}
    got = T.read_checkpoint(PREFIX, scope=scope, verify_data=True)
    assert set(got) == set(exp)                                  # Adam slots, beta powers, global_step are filtered
    for k in exp:
        assert got[k].dtype == exp[k].dtype and got[k].shape == exp[k].shape and np.array_equal(got[k], exp[k]), k
    if scope == "Text2Mel/":
        assert got["Text2Mel/TextEnc/embed_1/lookup_table"].shape == (8, 6)          # assembled from its two slices
        assert got["Text2Mel/TextEnc/alignment_lengths"].dtype == np.int32


def test_unsupported_dtype_is_refused_loudly():
    with pytest.raises(ValueError, match="half_precision_var has dtype enum 19"):
        T.read_checkpoint(PREFIX)                                # no scope: the float16 tensor is among the selected ones
    loose = T.read_checkpoint(PREFIX, strict=False)
    assert "Other/half_precision_var" not in loose and "SSRN/D_4/conv2d_transpose/kernel" in loose


def test_state_file_and_restore_functions(tmp_path):
    assert T.latest_checkpoint(BUNDLE) == PREFIX
    from ophelia_amd import architectures as A
    exp = _expected()

    class Sess:
        def __init__(self): self.W = {}
        def inventory(self, scope=None): return [(k, v.shape) for k, v in exp.items() if k.startswith("SSRN/")]
        def assign(self, W): self.W.update(W)

    class hp: logdir = str(tmp_path / "work" / "train")
    shutil.copytree(BUNDLE, hp.logdir + "-ssrn")                 # the reference's layout: {logdir}-ssrn/{checkpoint, model_epoch_E.*}
    s = Sess()
    assert A.restore_latest_model_parameters(s, hp, "ssrn") == "3"
    assert set(s.W) == {k for k in exp if k.startswith("SSRN/")}
    assert np.array_equal(s.W["SSRN/D_4/conv2d_transpose/kernel"], exp["SSRN/D_4/conv2d_transpose/kernel"])


def test_damage_is_detected(tmp_path):
    d = str(tmp_path / "b")
    shutil.copytree(BUNDLE, d)
    prefix = os.path.join(d, "model_epoch_3")
    idx = bytearray(open(prefix + ".index", "rb").read())
    for at in (7, len(idx) // 2, len(idx) - 60):                 # first block, a middle block (the compressed one is among them), the index block
        bad = bytearray(idx); bad[at] ^= 0x20
        open(prefix + ".index", "wb").write(bad)
        with pytest.raises(ValueError):
            T.read_checkpoint(prefix, scope="Text2Mel/")
    open(prefix + ".index", "wb").write(idx)
    shard = prefix + ".data-00001-of-00002"
    data = bytearray(open(shard, "rb").read())
    open(shard, "wb").write(data[:-40])                          # truncated shard: sizes no longer add up
    with pytest.raises(ValueError, match="bytes on disk"):
        T.read_checkpoint(prefix, strict=False)
    data[100] ^= 1
    open(shard, "wb").write(data)
    with pytest.raises(ValueError, match="fails its CRC"):
        T.read_checkpoint(prefix, strict=False, verify_data=True)
    os.remove(shard)
    with pytest.raises((OSError, ValueError)):
        T.read_checkpoint(prefix, strict=False)


def test_ordered_code_extents_beyond_one_byte():
    """Slice extents >= 64 take the multi-byte signed form (ordered_code.cc kLengthToHeaderBits): 2-byte 0xc000 | v, 3-byte 0xe00000 | v;
    negative numbers are the complement.  Known answers worked out by hand from the published table."""
    assert T._oc_read_signed(bytes([0x80]), 0) == (0, 1) and T._oc_read_signed(bytes([0x7f]), 0) == (-1, 1)
    assert T._oc_read_signed(bytes([0xbf]), 0) == (63, 1) and T._oc_read_signed(bytes([0x40]), 0) == (-64, 1)
    assert T._oc_read_signed(bytes([0xc0, 0x40]), 0) == (64, 2) and T._oc_read_signed(bytes([0xd0, 0x00]), 0) == (4096, 2)
    assert T._oc_read_signed(bytes([0x3f, 0xbf]), 0) == (-65, 2)
    assert T._oc_read_signed(bytes([0xe0, 0x20, 0x00]), 0) == (8192, 3)
