"""Model handles with the reference's names: counterpart of architectures.py
Graph 12-25 / add_data 69-81 (synthesis placeholders), SSRNGraph 133-142,
Text2MelGraph 181-239, plus the session object the synthesis functions drive.

In the reference, `g` (a TF graph object) and `sess` (a tf.Session) are opaque handles
passed to encode_text / synth_codedtext2mel / synth_mel2mag.  Here both graphs are
views on ONE Engine (one libophelia_hip handle = one GPU): Session owns it, loads the
variables by TF name and finalises them on first use."""
import os
import sys

from .engine import Engine
from . import tf_checkpoint
from . import weights as WT


class InvalidArgumentError(ValueError):
    """tf.errors.InvalidArgumentError's place in this session: a fetch that needs a placeholder the feed dict lacks."""


class Tensor(object):
    """A named graph tensor handle: what the reference's code holds as `g.K`, `g.mels`, ... and passes to
    sess.run as a fetch or as a feed-dict key (identity-hashed, like a tf.Tensor)."""

    def __init__(self, graph, name, shape):
        self.graph, self.name, self.shape = graph, name, shape

    def get_shape(self):
        return self.shape

    def __repr__(self):
        return "<%s/%s %s>" % (self.graph.scope, self.name, self.shape)


class Graph(object):
    def __init__(self, hp, mode="train", reuse=None):
        assert mode in ["train", "synthesize", "generate_attention"]
        if mode != "synthesize":
            raise NotImplementedError("only mode='synthesize' is on the hot path")
        self.mode, self.training, self.reuse, self.hp = mode, False, reuse, hp
        self.scope = None
        self.add_data(reuse=reuse)

    def add_data(self, reuse=None):
        """The synthesis placeholders of architectures.py:69-81 (both graphs declare all of them)."""
        hp = self.hp
        self.L = Tensor(self, "L", (None, hp.max_N))
        if hp.multispeaker:
            self.speakers = Tensor(self, "speakers", (None, None))
        if getattr(hp, "use_external_durations", False):
            self.durations = Tensor(self, "durations", (None, None, None))
        self.mels = Tensor(self, "mels", (None, hp.max_T, hp.n_mels))
        self.prev_max_attentions = Tensor(self, "prev_max_attentions", (None,))


class Text2MelGraph(Graph):
    """The tensors the reference feeds and fetches (architectures.py:188-239): K, V from TextEnc; Q from AudioEnc;
    R, alignments, max_attentions from Attention; Y_logits, Y from AudioDec.  Evaluated by Session.run."""

    def __init__(self, hp, mode="train", reuse=None):
        Graph.__init__(self, hp, mode, reuse)
        self.scope = "Text2Mel"
        d = hp.d
        self.K = Tensor(self, "K", (None, hp.max_N, d))
        self.V = Tensor(self, "V", (None, hp.max_N, d))
        self.Q = Tensor(self, "Q", (None, hp.max_T, d))
        self.R = Tensor(self, "R", (None, hp.max_T, 2 * d))
        self.alignments = Tensor(self, "alignments", (None, hp.max_N, hp.max_T))
        self.max_attentions = Tensor(self, "max_attentions", (None, hp.max_T))
        self.Y_logits = Tensor(self, "Y_logits", (None, hp.max_T, hp.n_mels))
        self.Y = Tensor(self, "Y", (None, hp.max_T, hp.n_mels))


class SSRNGraph(Graph):
    """architectures.py:133-142: Z_logits, Z = SSRN(mels)."""

    def __init__(self, hp, mode="train", reuse=None):
        Graph.__init__(self, hp, mode, reuse)
        self.scope = "SSRN"
        self.Z_logits = Tensor(self, "Z_logits", (None, hp.max_T * hp.r, hp.full_dim))
        self.Z = Tensor(self, "Z", (None, hp.max_T * hp.r, hp.full_dim))


class Session(object):
    """Stands in for tf.Session(): owns the device engine and the variable store."""

    def __init__(self, hp, device=0, engine=None):
        """engine: an existing Engine (weights loaded) to run on instead of creating one -- one GPU should carry ONE
        handle: each owns three CU-masked streams, and more than three of those alive time-slice."""
        self.hp = hp
        self._own = engine is None
        self.engine = Engine(hp, device=device) if engine is None else engine
        self.device = device
        self._pending = {}
        self._ready = engine is not None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._own:
            self.engine.close()

    def inventory(self, scope=None):
        return [(n, s) for n, s in self.engine.inventory() if scope is None or n.startswith(scope)]

    def assign(self, W):
        """Provide variables {TF name: array} (any subset; all must be present before the first run)."""
        if self._ready:
            raise RuntimeError("variables already finalised on the device")
        self._pending.update(W)

    def initialize_random(self, seed=0):
        """Counterpart of tf.global_variables_initializer() (synthesize.py:537)."""
        self.assign(WT.random_weights(self.inventory(), seed))

    def run(self, fetches, feed_dict=None):
        """sess.run for the three fetch sets of the synthesis path (synthesize.py:182-183, 237-239, 257) and, as a
        debug fetch, every other tensor the graphs declare:
          [g.K, g.V]                                <- {g.L [, g.speakers]}
          [g.Y, g.max_attentions, g.alignments, g.Q, g.R, g.Y_logits]
                                                    <- {g.K, g.V, g.mels, g.prev_max_attentions [, g.speakers]}
          g.Z / g.Z_logits                          <- {g.mels}
        One call evaluates the whole graph at the fed values (the Text2Mel graph over all max_T positions), exactly
        what the reference session does; the fast path of the loop is synth_codedtext2mel."""
        import numpy as np
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        feed = dict(feed_dict or {})
        g = fl[0].graph
        for t in list(fl) + list(feed):
            if not isinstance(t, Tensor) or t.graph is not g:
                raise ValueError("fetches and feeds must be tensors of one graph: %r" % (t,))
        fed = {t.name: v for t, v in feed.items()}
        eng = self.ensure_ready()
        hp = self.hp
        if isinstance(g, SSRNGraph):
            if "mels" not in fed:
                raise ValueError("SSRN graph: feed g.mels")
            if "ssrn_input" in (hp.multispeaker or []):
                # networks.py:457-465 reads g.speakers: TensorFlow refuses the run when the placeholder is not fed -- which is what
                # happens to the reference's own synth_mel2mag (synthesize.py:257 feeds g.mels alone)
                if "speakers" not in fed:
                    raise InvalidArgumentError("You must feed a value for placeholder tensor 'speakers' ('ssrn_input' in hp.multispeaker)")
                want = any(t.name == "Z_logits" for t in fl)
                r = eng.ssrn(fed["mels"], speaker_data=fed["speakers"], logits=want)
                vals = {"Z": r[0], "Z_logits": r[1]} if want else {"Z": r}
            elif any(t.name == "Z_logits" for t in fl):
                Z, Zl = eng.ssrn_logits(fed["mels"])         # the device's own pre-squash rows (networks.py:527-534)
                vals = {"Z": Z, "Z_logits": Zl}
            else:
                vals = {"Z": eng.ssrn(fed["mels"])}
        elif "L" in fed and all(t.name in ("K", "V") for t in fl):
            K, V = eng.encode_text(fed["L"], fed.get("speakers"))
            self._last_L = np.asarray(fed["L"])
            vals = {"K": K, "V": V}
        else:
            if "K" in fed and "V" in fed:
                K, V = fed["K"], fed["V"]
            elif "L" in fed:
                K, V = eng.encode_text(fed["L"], fed.get("speakers"))
                self._last_L = np.asarray(fed["L"])
            else:
                raise ValueError("Text2Mel graph: feed g.L, or g.K and g.V")
            if getattr(hp, "use_external_durations", False):
                raise NotImplementedError("graph evaluation with external durations: use synth_codedtext2mel")
            B = len(K)
            mels = fed.get("mels", np.zeros((B, hp.max_T, hp.n_mels), np.float32))
            prev = fed.get("prev_max_attentions", np.zeros((B,), np.int32))
            ends = None
            if getattr(hp, "turn_off_monotonic_for_synthesis", False):
                ends = np.asarray(hp.text_lengths) - 1                       # hp.text_lengths = text lengths + 1 (synthesize.py:505-507)
            vals = eng.text2mel_graph(K, V, mels, prev, ends=ends, speaker_data=fed.get("speakers"))
            vals["K"], vals["V"] = np.asarray(K), np.asarray(V)
        out = [vals[t.name] for t in fl]
        return out[0] if single else out

    def ensure_ready(self):
        if not self._ready:
            self.engine.load_weights(self._pending)
            self._pending = {}
            self._ready = True
        return self.engine


MODEL_TYPES = {"t2m": "Text2Mel", "ssrn": "SSRN"}


def _load_scope(sess, path, scope):
    """path: a TF-1 checkpoint prefix (<path>.index + .data-*) or an .npz keyed by TF variable names."""
    W = WT.load_npz(path) if path.endswith(".npz") else tf_checkpoint.read_checkpoint(path, scope=scope + "/")
    missing = [n for n, _ in sess.inventory(scope) if n not in W]
    if missing:
        sys.exit("checkpoint %s lacks %d variables of scope %s (first: %s)" % (path, len(missing), scope, missing[0]))
    sess.assign({n: W[n] for n, _ in sess.inventory(scope)})


def latest_checkpoint(savepath):
    """Latest model_epoch_{E}.npz under savepath (by epoch number), or None."""
    best, best_e = None, -1
    if os.path.isdir(savepath):
        for f in os.listdir(savepath):
            if f.startswith("model_epoch_") and f.endswith(".npz"):
                try:
                    e = int(f[len("model_epoch_"):-4])
                except ValueError:
                    continue
                if e > best_e:
                    best, best_e = os.path.join(savepath, f), e
    return best


def restore_latest_model_parameters(sess, hp, model_type):
    """synthesize.py:302-316.  Looks in {hp.logdir}-{t2m|ssrn}/ first for a TF-1 checkpoint (the `checkpoint`
    state file names model_epoch_{E}; read TF-free by tf_checkpoint.py), then for model_epoch_{E}.npz keyed by
    the TF variable names."""
    scope = MODEL_TYPES[model_type]
    savepath = hp.logdir + "-" + model_type
    ckpt = tf_checkpoint.latest_checkpoint(savepath) or latest_checkpoint(savepath)
    if ckpt is None:
        sys.exit("No %s at %s?" % (model_type, savepath))
    latest_epoch = ckpt.strip("/ ").split("/")[-1].replace("model_epoch_", "").replace(".npz", "")
    _load_scope(sess, ckpt, scope)
    print("Model of type %s restored from latest epoch %s" % (model_type, latest_epoch))
    return latest_epoch


def restore_archived_model_parameters(sess, hp, model_type, epoch_number):
    """synthesize.py:319-330."""
    scope = MODEL_TYPES[model_type]
    desired = hp.logdir + "-" + model_type + "/archive/model_epoch_" + str(epoch_number)
    if os.path.isfile(desired + ".index"):
        _load_scope(sess, desired, scope)
    elif os.path.isfile(desired + ".npz"):
        _load_scope(sess, desired + ".npz", scope)
    else:
        sys.exit("No %s at %s?" % (model_type, desired))
    print("Model of type %s restored from archived epoch %s" % (model_type, epoch_number))
