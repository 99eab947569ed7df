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


class Graph(object):
    def __init__(self, hp, mode="train", reuse=None):
        assert mode in ["train", "synthesize", "generate_attention"]
        if mode != "synthesize":
            raise NotImplementedError("only mode='synthesize' is on the hot path")
        self.mode, self.training, self.reuse, self.hp = mode, False, reuse, hp
        self.scope = None
        self.session = None          # bound by Session.bind()


class Text2MelGraph(Graph):
    """Exposes the tensors the reference fetches/feeds: K, V, Y, alignments, max_attentions."""
    scope = "Text2Mel"

    def __init__(self, hp, mode="train", reuse=None):
        Graph.__init__(self, hp, mode, reuse)
        self.scope = "Text2Mel"


class SSRNGraph(Graph):
    scope = "SSRN"

    def __init__(self, hp, mode="train", reuse=None):
        Graph.__init__(self, hp, mode, reuse)
        self.scope = "SSRN"


class Session(object):
    """Stands in for tf.Session(): owns the device engine and the variable store."""

    def __init__(self, hp, device=0):
        self.hp = hp
        self.engine = Engine(hp, device=device)
        self.device = device
        self._pending = {}
        self._ready = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
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
