# -*- coding: utf-8 -*-
"""
Golden-vector GENERATION tool (build container only; never shipped, never imported by the product, never runs
on the GPU box).

Runs the REFERENCE's own host code -- /root/reference/synthesize.py: synthesize(), encode_text(),
synth_codedtext2mel(), synth_mel2mag(), get_text_lengths(), split_batch(); and calculate_CDP_Ain_Aout.py:
getCDP(), getAP() -- under Python 3.10, reading the sources where they lie at generation time:

  * synthesize.py is Python-2 syntax (tuple parameter at :40): it is converted IN MEMORY with lib2to3's
    `tuple_params` fixer, and every `/` of the file is routed through `_py2div` (an AST pass), because the file has
    no `from __future__ import division`: `max(1, len(Y) / batchsize)` (:254) is an integer division there.
  * calculate_CDP_Ain_Aout.py mixes tabs and spaces (TabError under py3): loaded through str.expandtabs(8).
  * tensorflow / soundfile / librosa are absent: `tensorflow` is tests/golden/tf_standin.py (eager, torch CPU
    primitives), the others are empty stubs -- nothing on the Text2Mel + SSRN path calls into them once
    `synth_wave` (Griffin-Lim) and `plot_alignment` are replaced by recorders.

The reference builds its graphs once and feeds/fetches them through a tf.Session; the stand-in is eager.  The
session below therefore REBUILDS the reference graph (architectures.Text2MelGraph / SSRNGraph, the reference's own
classes) on every `run`, handing the fed arrays to the placeholders in creation order (architectures.py:70-81).
`hp.norm` is captured when synthesize() constructs each graph, so its "SSRN always uses layer norm" switch
(synthesize.py:513-534) is honoured by the rebuilds.  K and V are fed tensors in synth_codedtext2mel (:172); the
rebuild recomputes them from the L of the preceding encode_text call and asserts they are the arrays that were fed.
"""
import ast
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"


def _py2div(a, b):
    """Python-2 `/`: floor division when both operands are (numpy) integers, true division otherwise."""
    if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)) and not isinstance(a, bool):
        return a // b
    return a / b


class _DivToPy2(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right],
                                              keywords=[]), node)
        return node


def _load_py2_module(name, path, extra_globals=None):
    from lib2to3 import refactor
    src = open(path, encoding="utf-8").read()
    tool = refactor.RefactoringTool(["lib2to3.fixes.fix_tuple_params"], {"print_function": True})
    src3 = str(tool.refactor_string(src if src.endswith("\n") else src + "\n", path))
    tree = _DivToPy2().visit(ast.parse(src3, filename=path))
    ast.fix_missing_locations(tree)
    mod = types.ModuleType(name)
    mod.__file__ = path
    mod.__dict__["_py2div"] = _py2div
    mod.__dict__.update(extra_globals or {})
    exec(compile(tree, path, "exec"), mod.__dict__)
    return mod


def load_cdp_module():
    """reference calculate_CDP_Ain_Aout.py, tabs expanded (its body is otherwise plain py2/py3-neutral code)."""
    path = os.path.join(REF, "calculate_CDP_Ain_Aout.py")
    src = open(path, encoding="utf-8").read().expandtabs(8)
    mod = types.ModuleType("calculate_CDP_Ain_Aout")
    mod.__file__ = path
    import matplotlib
    matplotlib.use("Agg")
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


class Sym(object):
    """A graph tensor handle: identity-hashed, so it can key a feed dict like a tf.Tensor does."""
    def __init__(self, graph, name):
        self.graph, self.name = graph, name

    def __repr__(self):
        return "<%s.%s>" % (self.graph.kind, self.name)


class FakeGraph(object):
    """What synthesize() gets from Text2MelGraph(hp, mode='synthesize') / SSRNGraph(...): handles only.  The
    hyper-parameters that synthesize() changes between the two constructions are captured here."""
    NAMES = ("L", "speakers", "durations", "mels", "prev_max_attentions", "K", "V", "Q", "R", "Y", "Y_logits",
             "alignments", "max_attentions", "Z", "Z_logits")

    def __init__(self, kind, hp):
        self.kind = kind
        self.norm_at_build = hp.norm
        for n in self.NAMES:
            setattr(self, n, Sym(self, n))


class FakeSession(object):
    def __init__(self, host):
        self.host = host

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        if fetches is self.host.INIT:
            return None
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        g = fl[0].graph
        assert all(f.graph is g for f in fl)
        feed = {} if feed_dict is None else feed_dict
        for k in feed:
            assert isinstance(k, Sym) and k.graph is g, k
        out = self.host.evaluate(g, {k.name: v for k, v in feed.items()}, [f.name for f in fl])
        return out[0] if single else out


class ReferenceHost(object):
    """Owns the converted reference modules and the record of one synthesize() run."""
    INIT = object()

    def __init__(self, tf, ref_arch):
        self.tf, self.ref_arch = tf, ref_arch
        # what the reference's synthesize.py imports at module level and this image lacks
        for name in ("soundfile", "librosa"):
            sys.modules.setdefault(name, types.ModuleType(name))
        tf.Session = lambda *a, **k: FakeSession(self)
        tf.global_variables_initializer = lambda: ReferenceHost.INIT
        self.cdp = load_cdp_module()
        sys.modules["calculate_CDP_Ain_Aout"] = self.cdp
        with contextlib.redirect_stdout(io.StringIO()):
            self.mod = _load_py2_module("ref_synthesize", os.path.join(REF, "synthesize.py"))
        self.rec = None
        self.hp = None

    # -- the eager rebuild behind sess.run --------------------------------------------------------------------
    def evaluate(self, g, feed, names):
        hp, tf, rec = self.hp, self.tf, self.rec
        B = len(rec["L"])
        if "speakers" in feed:                     # synth_mel2mag feeds no speaker codes; the SSRN graph still declares
            rec["speakers"] = np.asarray(feed["speakers"])      # the placeholder (architectures.py:70-81 via add_data)
        speakers = rec.get("speakers")
        saved_norm = hp.norm
        hp.norm = g.norm_at_build
        try:
            if g.kind == "t2m":
                if "L" in feed:
                    rec["L_fed"] = np.asarray(feed["L"])
                L = rec["L_fed"]
                mels = feed.get("mels", np.zeros((B, hp.max_T, hp.n_mels), np.float32))
                prev = feed.get("prev_max_attentions", np.zeros((B,), np.int32))
                q = [L]
                if hp.multispeaker:
                    q.append(speakers)
                if hp.use_external_durations:          # placeholder order of architectures.py:70-81
                    q.append(np.asarray(feed.get("durations", rec.get("durations")), np.float32))
                q += [np.asarray(mels), np.asarray(prev).astype(np.int32)]
                tf.PLACEHOLDER_QUEUE[:] = q
                with contextlib.redirect_stdout(io.StringIO()):
                    rg = self.ref_arch.Text2MelGraph(hp, mode='synthesize')
                assert not tf.PLACEHOLDER_QUEUE
                if "K" in feed:                        # fed intermediate tensors must be what the graph computes from L
                    assert np.array_equal(np.asarray(rg.K), feed["K"]) and np.array_equal(np.asarray(rg.V), feed["V"])
                    j = rec["decode_runs"]
                    rec["decode_runs"] += 1
                    rec["trace"].append(np.asarray(rg.max_attentions)[:, j].astype(np.int32).copy())
                    if j == 0:
                        rec["Q_step0"] = np.asarray(rg.Q).copy()
                if not rec["t2m_names"]:
                    rec["t2m_names"] = list(tf.REQUESTED)
            else:
                mels = np.asarray(feed["mels"])
                nb = len(mels)
                q = [np.zeros((nb, hp.max_N), np.int32)]
                if hp.multispeaker:
                    q.append(np.asarray(speakers)[:nb])
                if hp.use_external_durations:
                    q.append(np.zeros((nb, hp.max_T, hp.max_N), np.float32))
                q += [mels, np.zeros((nb,), np.int32)]
                tf.PLACEHOLDER_QUEUE[:] = q
                before = list(tf.REQUESTED)
                tf.REQUESTED[:] = []
                with contextlib.redirect_stdout(io.StringIO()):
                    rg = self.ref_arch.SSRNGraph(hp, mode='synthesize')
                rec["ssrn_names"] = list(tf.REQUESTED)
                tf.REQUESTED[:] = before
                rec["ssrn_batches"].append(nb)
        finally:
            hp.norm = saved_norm
        return [np.asarray(getattr(rg, n)).copy() for n in names]

    # -- one run of the reference's synthesize() --------------------------------------------------------------
    def synthesize(self, hp, L, bases, speaker_id="", durations=None, num_sentences=0):
        """Calls the reference's synthesize(hp, ...) with load_data / restore / plotting / Griffin-Lim replaced by
        recorders, and returns everything it computed."""
        mod = self.mod
        self.hp = hp
        rec = self.rec = dict(L=L, trace=[], decode_runs=0, t2m_names=[], ssrn_names=[], ssrn_batches=[], waves=[],
                              plots=[], cdp=[], ap=[], durations=durations, report_crashes=False)
        outroot = tempfile.mkdtemp(prefix="ref_synth_")
        ds = {"fpaths": ["/nowhere/%s.wav" % b for b in bases], "texts": L}
        if durations is not None:
            ds["durations"] = durations
        mod.load_data = lambda hp_, mode="synthesis": ds                      # f-2 is pinned separately (frontend.npz)
        mod.restore_latest_model_parameters = lambda sess, hp_, kind: {"t2m": "1000", "ssrn": "900"}[kind]
        mod.restore_archived_model_parameters = lambda sess, hp_, kind, ep: None
        mod.Text2MelGraph = lambda hp_, mode=None: FakeGraph("t2m", hp_)
        mod.SSRNGraph = lambda hp_, mode=None: FakeGraph("ssrn", hp_)
        mod.tqdm = lambda x: x
        mod.plot_alignment = lambda hp_, al, utt_idx, t2m_epoch, dir="", outfile="": rec["plots"].append(
            (os.path.relpath(outfile, outroot), np.asarray(al).shape))
        real_cdp, real_ap = self.cdp.getCDP, self.cdp.getAP
        mod.getCDP = lambda A: rec["cdp"].append(float(real_cdp(A))) or rec["cdp"][-1]

        def _ap(A):
            # An utterance whose attention reaches `ends` at step 0 has lengths[i] == 0: its trimmed alignment is empty and
            # the reference's own getEnt divides by A.shape[0] == 0 (calculate_CDP_Ain_Aout.py:39) -- synthesize() dies in
            # its report loop there (wiring_lj_stop has such an utterance).  Recorded as NaN so that the rest of the run
            # (trimming, file naming) still yields goldens; rec["report_crashes"] says the reference itself would not.
            try:
                r = real_ap(A)
            except ZeroDivisionError:
                rec["report_crashes"] = True
                r = (float("nan"), float("nan"))
            rec["ap"].append((float(r[0]), float(r[1])))
            return r
        mod.getAP = _ap
        mod.synth_wave = lambda hp_, mag, outfile: rec["waves"].append((os.path.relpath(outfile, outroot), np.asarray(mag).copy()))
        wrapped = {}
        for fn in ("encode_text", "synth_codedtext2mel", "synth_mel2mag", "get_text_lengths"):
            real = getattr(mod, fn)

            def w(*a, __real=real, __fn=fn, **k):
                r = __real(*a, **k)
                rec[__fn] = r
                return r
            wrapped[fn] = real
            setattr(mod, fn, w)
        saved = hp.sampledir if hasattr(hp, "sampledir") else None
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                mod.synthesize(hp, speaker_id=speaker_id, num_sentences=num_sentences, ncores=1, topoutdir=outroot)
        finally:
            for fn, real in wrapped.items():
                setattr(mod, fn, real)
            if saved is not None:
                hp.sampledir = saved
        rec["stdout"] = buf.getvalue()
        rec["outdirs"] = sorted(os.listdir(outroot))
        return rec
