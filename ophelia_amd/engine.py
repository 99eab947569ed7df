"""Engine: one handle of libophelia_hip.so (= one GPU, one host thread).  It plays the
role of the reference's (graph, session) pair (synthesize.py:511-537): built from `hp`,
loaded with variables by TF name, then driven through encode_text / text2mel / ssrn."""
import ctypes as C
import weakref

import numpy as np

from . import _lib


class _PinnedBlock(object):
    """One pinned host buffer (oph_host_alloc) exposed through the array interface; it goes back to the pool when the
    last array viewing it is collected."""

    def __init__(self, pool, ptr, nbytes):
        self.pool, self.ptr, self.nbytes = pool, ptr, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self.pool._release(self.nbytes, self.ptr)
        except Exception:
            pass


class PinnedPool(object):
    """Result arrays in pinned host memory: device-to-host copies into them are asynchronous DMA at the link's rate (a
    pageable NumPy array is copied through a bounce buffer, several times slower for the 52 MB of a batch's magnitudes).
    Buffers are recycled by size; at most `keep` idle ones per size stay allocated."""

    def __init__(self, keep=3):
        self.keep = keep
        self.idle = {}

    def empty(self, shape, dtype=np.float32):
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape)) * dtype.itemsize)
        lst = self.idle.get(nbytes)
        if lst:
            ptr = lst.pop()
        else:
            out = C.c_void_p()
            if _lib.load().oph_host_alloc(nbytes, C.byref(out)) != 0 or not out.value:
                return np.empty(shape, dtype)            # no pinned memory left: an ordinary array works everywhere, slower
            ptr = out.value
        block = _PinnedBlock(self, ptr, nbytes)
        return np.asarray(block)[:int(np.prod(shape)) * dtype.itemsize].view(dtype).reshape(shape)

    def _release(self, nbytes, ptr):
        lst = self.idle.setdefault(nbytes, [])
        if len(lst) < self.keep:
            lst.append(ptr)
        else:
            _lib.load().oph_host_free(ptr)

    def drain(self):
        for lst in self.idle.values():
            while lst:
                _lib.load().oph_host_free(lst.pop())


PINNED = PinnedPool()


def dims_from_hp(hp, max_N=None, max_T=None):
    ms = list(getattr(hp, "multispeaker", []) or [])
    positions = {"audio_decoder_input": _lib.FLAG_SPK_AUDIO_DECODER_INPUT,
                 "text_encoder_input": _lib.FLAG_SPK_TEXT_ENCODER_INPUT,
                 "text_encoder_towards_end": _lib.FLAG_SPK_TEXT_ENCODER_TOWARDS_END,
                 "learn_channel_contributions": _lib.FLAG_LCC,
                 "audio_encoder_input": _lib.FLAG_SPK_AUDIO_ENCODER_INPUT,
                 "ssrn_input": _lib.FLAG_SPK_SSRN_INPUT,
                 # host-side only (data_load.load_vocab / phones_normalize, data_load.py:42-46, 61-64): the graphs do not change
                 "speaker_dependent_phones": 0}
    unsupported = [p for p in ms if p not in positions]
    if unsupported:
        raise NotImplementedError("multispeaker positions %s are not supported (supported: %s)"
                                  % (unsupported, sorted(positions)))
    norm = getattr(hp, "norm", "layer")
    if norm not in ("layer", None):
        raise NotImplementedError("hp.norm=%r is not supported ('layer' or None)" % (norm,))
    # SURVEY.md section 2 OUT-OF-SCOPE: the Merlin-label text encoder, the alternative history types and label inputs
    for attr, want in (("text_encoder_type", "DCTTS_standard"),
                       ("history_type", "DCTTS_standard"), ("merlin_label_dir", "")):
        if getattr(hp, attr, want) != want:
            raise NotImplementedError("hp.%s=%r is outside the hot-path scope" % (attr, getattr(hp, attr)))
    d = _lib.OphDims()
    d.vocab = len(hp.vocab)
    d.e, d.d, d.c = hp.e, hp.d, hp.c
    d.n_mels, d.full_dim, d.r = hp.n_mels, hp.full_dim, hp.r
    d.max_N = hp.max_N if max_N is None else max_N
    d.max_T = hp.max_T if max_T is None else max_T
    d.attention_win_size = hp.attention_win_size
    d.nspeakers = getattr(hp, "nspeakers", 0) if ms else 0
    d.speaker_embedding_size = getattr(hp, "speaker_embedding_size", 0) if ms else 0
    d.flags = (_lib.FLAG_NORM_NONE if norm is None else 0)
    if not getattr(hp, "concatenate_query", True):
        d.flags |= _lib.FLAG_NO_CONCAT_QUERY            # networks.py:317-321
    if not getattr(hp, "squash_output_t2m", True):
        d.flags |= _lib.FLAG_NO_SQUASH_T2M              # networks.py:430-433
    if not getattr(hp, "squash_output_ssrn", True):
        d.flags |= _lib.FLAG_NO_SQUASH_SSRN             # networks.py:533-536
    if getattr(hp, "turn_off_monotonic_for_synthesis", False):
        if d.max_N > 256:
            raise NotImplementedError("turn_off_monotonic_for_synthesis is supported up to max_N = 256")
        d.flags |= _lib.FLAG_NO_MONOTONIC
    for p in ms:
        d.flags |= positions[p]
    return d


class Engine(object):
    """resident_results (default True, or hp.resident_results): the K, V and Y arrays a call returns are READ-ONLY views of
    pinned memory, and handing the very same arrays to the next call skips the upload (K, V and Y never leave HBM, SSRN has been
    streaming while the decoder ran).  The reference returns ordinary writable arrays (synthesize.py:166, 240): a caller that
    post-processes them in place (`Y[i, t_end:] = 0`) either copies first (`np.array(Y)`: uploaded like any other array, same
    results) or builds the engine with resident_results=False: every result is then a plain writable array and every input is
    uploaded, exactly the reference's data flow."""

    def __init__(self, hp, device=0, max_N=None, max_T=None, resident_results=None, options=None):
        """options: launch-path / arithmetic options of oph_create_opts (include/ophelia_hip.h) as a dict {"DECODE": "runs", "NO_CHAIN": 1}
        or the string itself; default hp.engine_options, else none.  The library reads no environment variable for them."""
        self.lib = _lib.load()
        self.dims = dims_from_hp(hp, max_N, max_T)
        self.hp = hp
        self.device = int(device)
        self._h = C.c_void_p()
        if options is None:
            options = getattr(hp, "engine_options", None)
        if isinstance(options, dict):
            options = " ".join("%s=%s" % (k, v) for k, v in options.items())
        self.options = options or ""
        rc = self.lib.oph_create_opts(C.byref(self.dims), int(device), self.options.encode() or None, C.byref(self._h))
        if rc != 0:
            raise _lib.OpheliaHipError("oph_create failed (%d): %s" % (rc, self.lib.oph_last_error(None).decode()))
        self.multispeaker = bool(self.dims.flags & (_lib.FLAG_SPK_AUDIO_DECODER_INPUT | _lib.FLAG_SPK_TEXT_ENCODER_INPUT |
                                                    _lib.FLAG_SPK_TEXT_ENCODER_TOWARDS_END | _lib.FLAG_LCC |
                                                    _lib.FLAG_SPK_AUDIO_ENCODER_INPUT))
        self.B = 0
        self._kv_token = None          # (K, V) arrays of the last encode_text whose values are still in HBM
        self._y_token = None           # Y array of the last decode whose values are still in HBM
        self._z_spec = None            # pinned array the last decode's speculative SSRN has been copying its rows into
        self.resident_results = bool(getattr(hp, "resident_results", True) if resident_results is None else resident_results)

    # Residency between the three session calls (include/ophelia_hip.h): the arrays a call returns are read-only and the
    # engine remembers them; handing the very same (still read-only) arrays to the next call skips the upload -- K,V never
    # leave HBM, SSRN has been streaming over Y while the decoder ran.  Any other array (a copy, a modified or a made-writeable
    # one) is uploaded and used as it is.
    def _seal(self, a):
        if self.resident_results:
            a.flags.writeable = False
        return a

    def _drop_spec(self):
        """A different batch (or a different kind of call) starts: the magnitudes the last decode's speculative SSRN streamed to the
        host belong to nobody any more.  Its copies must have landed before the pinned buffer goes back to the pool."""
        if self._z_spec is not None:
            self.lib.oph_synchronize(self._h)
            self.lib.oph_set_mag_destination(self._h, None)
            self._z_spec = None

    def _is_resident(self, token, *arrays):
        if token is None or len(token) != len(arrays) or not self.resident_results:
            return False
        for ref, a in zip(token, arrays):
            if ref() is not a or a.flags.writeable:
                return False
        return True

    def is_resident_mel(self, Y):
        return isinstance(Y, np.ndarray) and self._is_resident(self._y_token, Y)

    # -- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise _lib.OpheliaHipError("libophelia_hip error %d: %s" % (rc, self.lib.oph_last_error(self._h).decode()))

    def close(self):
        if self._h:
            self.lib.oph_destroy(self._h)          # synchronises every stream first
            self._h = C.c_void_p()
            self._z_spec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights
    def inventory(self):
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        rank = C.c_int()
        for i in range(self.lib.oph_num_weights(self._h)):
            self._chk(self.lib.oph_weight_info(self._h, i, name, 256, shape, C.byref(rank)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(rank.value))))
        return out

    def load_weights(self, W):
        for name, shape in self.inventory():
            if name not in W:
                raise KeyError("missing variable %s" % name)
            a = np.ascontiguousarray(W[name], dtype=np.float32)
            shp = (C.c_int64 * 4)(*a.shape)
            self._chk(self.lib.oph_set_weight(self._h, name.encode(), _lib.fptr(a), shp, a.ndim))
        self._chk(self.lib.oph_finalize_weights(self._h))

    def load_weights_device(self, device_ptr, n_floats):
        """Every variable at once from a float32 buffer on this engine's GPU (inventory order, back to back): repacked in place by
        device kernels (oph_set_weights_device).  The buffer must stay alive until this call returns."""
        self._chk(self.lib.oph_set_weights_device(self._h, C.c_void_p(int(device_ptr)), int(n_floats)))
        self._chk(self.lib.oph_finalize_weights(self._h))

    # -- the three session calls (host arrays in/out), synthesize.py:232-260
    def _spk(self, speaker_data, B):
        if not self.multispeaker:
            return None, None
        if speaker_data is None:
            raise ValueError("multispeaker model: speaker_data (B,1) required (synthesize.py:496-503)")
        s = np.ascontiguousarray(np.asarray(speaker_data).reshape(B), dtype=np.int32)
        return s, _lib.iptr(s)

    def encode_text(self, L, speaker_data=None):
        L = np.ascontiguousarray(L, dtype=np.int32)
        B, N = L.shape
        assert N == self.dims.max_N, (N, self.dims.max_N)
        K = PINNED.empty((B, N, self.dims.d))
        V = PINNED.empty((B, N, self.dims.d))
        s, sp = self._spk(speaker_data, B)
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_encode_text(self._h, _lib.iptr(L), sp, B, _lib.fptr(K), _lib.fptr(V)))
        self._seal(K), self._seal(V)
        self._kv_token = (weakref.ref(K), weakref.ref(V))
        self.B = B
        return K, V

    def text2mel(self, K, V, ends, speaker_data=None, stop_mode=_lib.STOP_REFERENCE):
        resident = isinstance(K, np.ndarray) and isinstance(V, np.ndarray) and self._is_resident(self._kv_token, K, V)
        if not resident:
            K = np.ascontiguousarray(K, dtype=np.float32)
            V = np.ascontiguousarray(V, dtype=np.float32)
        B = K.shape[0]
        assert K.shape == V.shape == (B, self.dims.max_N, self.dims.d)
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        Y = PINNED.empty((B, self.dims.max_T, self.dims.n_mels))
        t_ends = np.empty((B,), np.int32)
        al = PINNED.empty((B, self.dims.max_N, self.dims.max_T))
        steps = C.c_int32()
        s, sp = self._spk(speaker_data, B)
        self._y_token = None
        if not resident:
            self._kv_token = None
        # the magnitudes this decode's speculative SSRN produces go straight into the array a following ssrn(Y) returns
        self._drop_spec()                     # nobody asked for the previous decode's magnitudes
        if self.resident_results:
            self._z_spec = PINNED.empty((B, self.dims.max_T * self.dims.r, self.dims.full_dim))
            self._chk(self.lib.oph_set_mag_destination(self._h, _lib.fptr(self._z_spec)))
        try:
            self._chk(self.lib.oph_text2mel(self._h, None if resident else _lib.fptr(K), None if resident else _lib.fptr(V),
                                            _lib.iptr(ends), sp, B, int(stop_mode),
                                            _lib.fptr(Y), _lib.iptr(t_ends), _lib.fptr(al), C.byref(steps)))
        except Exception:
            self.lib.oph_set_mag_destination(self._h, None)
            self._z_spec = None
            raise
        self._y_token = (weakref.ref(self._seal(Y)),)
        self.B = B
        return Y, t_ends, al, steps.value

    def text2mel_durations(self, K, V, durations, speaker_data=None, n_steps=0):
        """synth_codedtext2mel with hp.use_external_durations: `durations` (B, max_T, max_N) hard selection matrices
        (data_load.py:243-251).  K may be None (FixedAttention never reads it).  n_steps > 0 overrides the
        min(max_T, max(t_ends)+1) steps of the reference's break rule (sharded batches)."""
        V = np.ascontiguousarray(V, dtype=np.float32)
        B = V.shape[0]
        assert V.shape == (B, self.dims.max_N, self.dims.d)
        Kp = None
        if K is not None:
            K = np.ascontiguousarray(K, dtype=np.float32)
            assert K.shape == V.shape
            Kp = _lib.fptr(K)
        D = np.ascontiguousarray(durations, dtype=np.float32)
        assert D.shape == (B, self.dims.max_T, self.dims.max_N), D.shape
        Y = PINNED.empty((B, self.dims.max_T, self.dims.n_mels))
        t_ends = np.empty((B,), np.int32)
        al = PINNED.empty((B, self.dims.max_N, self.dims.max_T))
        steps = C.c_int32()
        s, sp = self._spk(speaker_data, B)
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_text2mel_durations(self._h, Kp, _lib.fptr(V), _lib.fptr(D), sp, B, int(n_steps),
                                                  _lib.fptr(Y), _lib.iptr(t_ends), _lib.fptr(al), C.byref(steps)))
        self._y_token = (weakref.ref(self._seal(Y)),)
        self.B = B
        return Y, t_ends, al, steps.value

    def text2mel_graph(self, K, V, mels, prev_max_attentions, ends=None, speaker_data=None):
        """ONE evaluation of the synthesis graph at fixed feeds = one sess.run of the reference's loop
        (synthesize.py:172,181-183): every max_T position under the one mask of prev_max_attentions.  Returns a dict
        with Q, R, Y_logits, Y, alignments, max_attentions (architectures.py:188-239).  O(max_T) per call: the fetch
        surface for validation and feed/fetch-style callers; the decode loop proper is text2mel()."""
        K = np.ascontiguousarray(K, dtype=np.float32)
        V = np.ascontiguousarray(V, dtype=np.float32)
        B = K.shape[0]
        d = self.dims
        assert K.shape == V.shape == (B, d.max_N, d.d)
        mels = np.ascontiguousarray(mels, dtype=np.float32)
        assert mels.shape == (B, d.max_T, d.n_mels), mels.shape
        pm = np.ascontiguousarray(np.asarray(prev_max_attentions).reshape(B), dtype=np.int32)
        ep = None
        if ends is not None:
            ends = np.ascontiguousarray(np.asarray(ends).reshape(B), dtype=np.int32)
            ep = _lib.iptr(ends)
        out = dict(Q=np.empty((B, d.max_T, d.d), np.float32), R=np.empty((B, d.max_T, 2 * d.d), np.float32),
                   Y_logits=np.empty((B, d.max_T, d.n_mels), np.float32), Y=np.empty((B, d.max_T, d.n_mels), np.float32),
                   alignments=np.empty((B, d.max_N, d.max_T), np.float32), max_attentions=np.empty((B, d.max_T), np.int32))
        s, sp = self._spk(speaker_data, B)
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_text2mel_graph(self._h, _lib.fptr(K), _lib.fptr(V), _lib.fptr(mels), _lib.iptr(pm), ep, sp, B,
                                              _lib.fptr(out["Q"]), _lib.fptr(out["R"]), _lib.fptr(out["Y_logits"]),
                                              _lib.fptr(out["Y"]), _lib.fptr(out["alignments"]), _lib.iptr(out["max_attentions"])))
        return out

    def ssrn(self, Y, speaker_data=None, logits=False):
        """Z = g.Z of the SSRN graph at g.mels = Y.  speaker_data (B,) / (B, 1): g.speakers, for configurations with 'ssrn_input'
        in hp.multispeaker (networks.py:457-465) -- the reference's synth_mel2mag cannot feed them (synthesize.py:257); this is
        the graph surface.  logits=True returns (Z, Z_logits)."""
        if speaker_data is not None:
            Y = np.ascontiguousarray(Y, dtype=np.float32)
            B, T, nm = Y.shape
            assert nm == self.dims.n_mels
            spk = np.ascontiguousarray(np.asarray(speaker_data).reshape(B), dtype=np.int32)
            Z = PINNED.empty((B, T * self.dims.r, self.dims.full_dim))
            Zl = PINNED.empty((B, T * self.dims.r, self.dims.full_dim)) if logits else None
            self._kv_token = self._y_token = None
            self._drop_spec()
            self._chk(self.lib.oph_ssrn_speakers(self._h, _lib.fptr(Y), _lib.iptr(spk), B, T, _lib.fptr(Z), None if Zl is None else _lib.fptr(Zl)))
            return (Z, Zl) if logits else Z
        if logits:
            return self.ssrn_logits(Y)
        resident = self.is_resident_mel(Y)
        if not resident:
            Y = np.ascontiguousarray(Y, dtype=np.float32)
        B, T, nm = Y.shape
        assert nm == self.dims.n_mels
        spec = getattr(self, "_z_spec", None)
        if resident and spec is not None and spec.shape == (B, T * self.dims.r, self.dims.full_dim):
            Z = spec                                     # most of it arrived while the decoder was running
        else:
            Z = PINNED.empty((B, T * self.dims.r, self.dims.full_dim))
        if not resident:
            self._kv_token = self._y_token = None        # the batch workspaces are reused
            if Z is not spec:
                self._drop_spec()
        self._chk(self.lib.oph_ssrn(self._h, None if resident else _lib.fptr(Y), B, T, _lib.fptr(Z)))
        if Z is spec:
            self._z_spec = None                          # handed to the caller: the next decode gets a fresh buffer
            self.lib.oph_set_mag_destination(self._h, None)
        return Z

    def ssrn_logits(self, Y):
        """(Z, Z_logits) = the SSRN graph's two fetchable tensors (networks.py:527-534)."""
        Y = np.ascontiguousarray(Y, dtype=np.float32)
        B, T, nm = Y.shape
        assert nm == self.dims.n_mels
        Z = PINNED.empty((B, T * self.dims.r, self.dims.full_dim))
        Zl = PINNED.empty((B, T * self.dims.r, self.dims.full_dim))
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_ssrn_logits(self._h, _lib.fptr(Y), B, T, _lib.fptr(Z), _lib.fptr(Zl)))
        return Z, Zl

    def counters(self):
        """What the pipeline did since the handle was created (oph_get_counters)."""
        v = (C.c_int64 * 11)()
        self._chk(self.lib.oph_get_counters(self._h, v, 11))
        return dict(zip(("textenc", "preenc_used", "chunks_streamed", "loop_decodes", "loop_fallbacks", "tile_resumes", "cone_loops", "fp16_guard",
                         "masked_streams", "recoveries", "degraded_left"), [int(x) for x in v]))

    def set_streaming(self, on=True):
        """SSRN over the frames a running decode has already produced (default on); off: SSRN only when asked for.
        An integer >= 2 also sets the number of mel frames per streamed chunk."""
        self._chk(self.lib.oph_set_streaming(self._h, int(on)))

    # -- device-resident pipeline
    def stage_text(self, L, ends, speaker_data=None):
        L = np.ascontiguousarray(L, dtype=np.int32)
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        B = L.shape[0]
        s, sp = self._spk(speaker_data, B)
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_stage_text(self._h, _lib.iptr(L), _lib.iptr(ends), sp, B))
        self.B = B

    def stage_text_next(self, L, ends, speaker_data=None):
        """The text of the batch after the staged / running one (second text slot): the next run_resident / run_host
        switches to it, and the decode before that pre-encodes it on the SSRN partition."""
        L = np.ascontiguousarray(L, dtype=np.int32)
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        s, sp = self._spk(speaker_data, L.shape[0])
        self._chk(self.lib.oph_stage_text_next(self._h, _lib.iptr(L), _lib.iptr(ends), sp, L.shape[0]))

    def run_host(self, stop_mode=_lib.STOP_REFERENCE, want_kv=False):
        """The staged batch host -> host in one call (oph_run_host): returns a dict with Y, t_ends, alignments, Z, steps
        (and K, V if asked for) in pinned arrays; the copies run under the decode."""
        d, B = self.dims, self.B
        out = dict(Y=PINNED.empty((B, d.max_T, d.n_mels)), t_ends=PINNED.empty((B,), np.int32),
                   alignments=PINNED.empty((B, d.max_N, d.max_T)), Z=PINNED.empty((B, d.max_T * d.r, d.full_dim)))
        if want_kv:
            out["K"], out["V"] = PINNED.empty((B, d.max_N, d.d)), PINNED.empty((B, d.max_N, d.d))
        steps = C.c_int32()
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_run_host(self._h, int(stop_mode), _lib.fptr(out["K"]) if want_kv else None,
                                        _lib.fptr(out["V"]) if want_kv else None, _lib.fptr(out["Y"]), _lib.iptr(out["t_ends"]),
                                        _lib.fptr(out["alignments"]), _lib.fptr(out["Z"]), C.byref(steps)))
        out["steps"] = steps.value
        return out

    def run_resident(self, stop_mode=_lib.STOP_NEVER, run_ssrn=True, pipelined=False):
        """encode -> decode -> SSRN on the staged batch, everything staying in HBM.  pipelined=True queues
        this batch's SSRN on its own CU partition without joining it, so the next call overlaps it."""
        steps = C.c_int32()
        mode = 2 if (run_ssrn and pipelined) else int(bool(run_ssrn))
        self._kv_token = self._y_token = None
        self._drop_spec()
        self._chk(self.lib.oph_run_resident(self._h, int(stop_mode), mode, C.byref(steps)))
        return steps.value

    def decode_steps(self, t_begin, t_end, stop_mode):
        steps = C.c_int32()
        if int(t_begin) == 0:
            self._kv_token = self._y_token = None
            self._drop_spec()                            # a new decode; a resume (t_begin > 0) continues the batch the magnitudes belong to
        else:
            self._y_token = None                         # the frames in HBM change: an earlier fetch no longer names them
        self._chk(self.lib.oph_decode_steps(self._h, int(t_begin), int(t_end), int(stop_mode), C.byref(steps)))
        return steps.value

    def run_ssrn_resident(self):
        self._chk(self.lib.oph_run_ssrn_resident(self._h))

    def fetch_kv(self):
        K = np.empty((self.B, self.dims.max_N, self.dims.d), np.float32)
        V = np.empty_like(K)
        self._chk(self.lib.oph_fetch_kv(self._h, _lib.fptr(K), _lib.fptr(V)))
        return K, V

    def fetch_mel(self):
        Y = PINNED.empty((self.B, self.dims.max_T, self.dims.n_mels))
        t_ends = np.empty((self.B,), np.int32)
        al = PINNED.empty((self.B, self.dims.max_N, self.dims.max_T))
        self._chk(self.lib.oph_fetch_mel(self._h, _lib.fptr(Y), _lib.iptr(t_ends), _lib.fptr(al)))
        self._y_token = (weakref.ref(self._seal(Y)),)       # these ARE the frames in HBM: ssrn(Y) may continue from them
        return Y, t_ends, al

    def fetch_mag(self):
        Z = PINNED.empty((self.B, self.dims.max_T * self.dims.r, self.dims.full_dim))
        self._chk(self.lib.oph_fetch_mag(self._h, _lib.fptr(Z)))
        return Z

    def set_ssrn_precision(self, mode):
        """2 = split-fp16 x3 contractions with fp32 accumulate (default, fp32-class accuracy), 1 = split-bf16 x3,
        0 = fp32 MFMA."""
        self._chk(self.lib.oph_set_ssrn_precision(self._h, int(mode)))

    def set_precision(self, which, mode):
        """which: "ssrn" | "cone" | "textenc"; mode as set_ssrn_precision (oph_set_precision)."""
        self._chk(self.lib.oph_set_precision(self._h, {"ssrn": 0, "cone": 1, "textenc": 2}[which], int(mode)))

    def synchronize(self):
        self._chk(self.lib.oph_synchronize(self._h))

    # -- measurement
    def timer_start(self):
        self._chk(self.lib.oph_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.oph_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def loop_clock(self, reset=False):
        """(launches, total_us) of the whole-decode launches as the kernel itself clocked them (oph_loop_clock)."""
        n, us = C.c_int64(), C.c_double()
        self._chk(self.lib.oph_loop_clock(self._h, C.byref(n), C.byref(us), int(bool(reset))))
        return n.value, us.value

    def profile_enable(self, on=True):
        self._chk(self.lib.oph_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._chk(self.lib.oph_profile_reset(self._h))

    def profile(self):
        out = []
        name = C.create_string_buffer(64)
        n, ms, by, fl = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        for i in range(self.lib.oph_profile_count(self._h)):
            self._chk(self.lib.oph_profile_get(self._h, i, name, 64, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl)))
            out.append(dict(name=name.value.decode(), launches=n.value, total_ms=ms.value,
                            alg_bytes=by.value, alg_flops=fl.value))
        return out
