"""Every decode flavour of the library must produce the same utterances: the whole-decode launch (default), two launches per
step, one launch per layer, and the cone variants (no cone head, no fused small levels).  The flavours are launch-path options of
oph_create_opts (the library reads no environment variable for them); each runs in its own interpreter on the same seeded model and text; the parent compares
the mel frames, alignments, stop steps and the attention trace -- with the default flavour and with the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import hp_from_snapshot
from oracle import ophelia_oracle as O          # weights / text generators only (seeded inputs)
from ophelia_amd.engine import Engine
hp = hp_from_snapshot("lj_tutorial.cfg")
hp.max_T = int(sys.argv[3]); B = int(sys.argv[4]); stop_mode = int(sys.argv[5])
W = O.random_weights(hp, 2)
L = O.random_text(hp, B, 3, min_len=6, max_len=24) if stop_mode == 0 else O.random_text(hp, B, 3, min_len=75, max_len=149)
ends = O.get_text_lengths(L)
eng = Engine(hp, device=0, options=json.loads(sys.argv[6])); eng.load_weights(W); eng.set_ssrn_precision(0)
K, V = eng.encode_text(L)
Y, t_ends, al, steps = eng.text2mel(K, V, ends, stop_mode=stop_mode)
Y2, t2, al2, steps2 = eng.text2mel(K, V, ends, stop_mode=stop_mode)         # a second decode on the same handle: state fully reset
assert steps2 == steps and np.array_equal(Y2, Y) and np.array_equal(al2, al), "second decode differs from the first"
Z = eng.ssrn(Y2)                 # the resident frames: SSRN streamed under the decode (or not: loop_nostream, runs, layers)
Z1 = eng.ssrn(np.array(Y2))      # a copy: uploaded, SSRN in one piece
assert np.array_equal(Z, Z1), "streamed SSRN differs from the one-shot evaluation"
np.savez(sys.argv[2], Y=Y, al=al, t_ends=np.asarray(t_ends), steps=steps, K=K, V=V, ends=ends, Zsum=Z.sum(axis=(1, 2)))
eng.close()
"""

FLAVOURS = {
    "loop": {},
    "runs": {"DECODE": "runs"},
    "layers": {"DECODE": "layers"},
    "loop_generic": {"NO_CHAIN": 1},                               # dec_loop, the generic whole-decode kernel (what non-standard geometries take)
    "loop_rows4": {"RUN_ROWS": 4},
    "loop_nohead": {"NO_CONE_HEAD": 1},
    "loop_nofused": {"NO_FUSED_CONE": 1},                          # the cone's levels as contraction + ln_rows launches (the recovery ladder's first rung)
    "loop_nofc": {"CONE_FC_ROWS": 0},
    "loop_fc256": {"CONE_FC_ROWS": 256, "CONE_FC_INSPLIT": 1},
    "loop_nostream": {"NO_STREAM_SSRN": 1},
    "loop_conefp32": {"CONE_PREC": 0, "TEXTENC_PREC": 0},        # fp32 MFMA for the cone's large levels and TextEnc (default: split-fp16 x3)
    "loop_conebf16": {"CONE_PREC": 1},                             # the split-bf16 experiment
}


# fp32-class flavours (fp32 MFMA, split-fp16 x3) differ at the level of summation order; the split-bf16 x3 cone experiment drops terms below 2^-16
TOL = {"loop_conebf16": 3e-4}
# flavours that only change how launches are cut, not what a row's arithmetic is
BITWISE = {"loop_nostream"}


def _run(tmp_path, name, env_extra, max_T, B, stop_mode):
    out = str(tmp_path / (name + ".npz"))
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, out, str(max_T), str(B), str(stop_mode), json.dumps(env_extra)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, "%s failed:\n%s" % (name, r.stdout[-3000:])
    return np.load(out)


@pytest.mark.parametrize("stop_mode,max_T,B", [(1, 96, 16), (0, 120, 5)])
def test_decode_flavours_agree(tmp_path, stop_mode, max_T, B):
    """stop_mode 1: fixed length (every step runs); stop_mode 0: the reference's early stop (short texts reach their end)."""
    ref = _run(tmp_path, "loop", FLAVOURS["loop"], max_T, B, stop_mode)
    if stop_mode == 0:
        assert int(ref["steps"]) < max_T, "the early-stop case must actually stop early"
    for name, env in FLAVOURS.items():
        if name == "loop":
            continue
        got = _run(tmp_path, name, env, max_T, B, stop_mode)
        assert int(got["steps"]) == int(ref["steps"]), name
        assert got["t_ends"].tolist() == ref["t_ends"].tolist(), name
        assert np.array_equal(got["al"].argmax(1), ref["al"].argmax(1)), "%s: attention trace differs" % name
        ey, ea = np.abs(got["Y"] - ref["Y"]).max(), np.abs(got["al"] - ref["al"]).max()
        print("%-16s vs loop: max-abs Y %.2e align %.2e" % (name, ey, ea))
        tol = TOL.get(name, 2e-5)
        assert ey < tol and ea < tol, name
        if name in BITWISE:
            assert np.array_equal(got["Y"], ref["Y"]) and np.array_equal(got["al"], ref["al"]), "%s: the same arithmetic in the same order must give the same bits" % name
        assert np.allclose(got["Zsum"], ref["Zsum"], rtol=1e-4), name
    # and the default flavour against the oracle (exact incremental algorithm) on the same K, V
    sys.path.insert(0, ROOT)
    from conftest import hp_from_snapshot
    from oracle import ophelia_oracle as O
    hp = hp_from_snapshot("lj_tutorial.cfg")
    hp.max_T = max_T
    W = O.random_weights(hp, 2)
    idx = np.arange(min(B, 4))
    Y0, t0, al0 = O.synth_codedtext2mel_incremental(hp, W, ref["K"][idx], ref["V"][idx], ref["ends"][idx], stop=False)
    n = int(ref["steps"]) if stop_mode == 0 else max_T
    # (early stop couples the batch: compare the frames every utterance has in common with a free-running oracle decode)
    ey = np.abs(ref["Y"][idx, :n] - Y0[:, :n]).max()
    print("loop vs oracle: max-abs Y %.2e over %d steps" % (ey, n))
    assert ey < 1e-4
