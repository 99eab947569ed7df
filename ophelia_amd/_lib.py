"""ctypes binding of libophelia_hip.so (C ABI: include/ophelia_hip.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) into
ophelia_amd/lib/.  Loading fails loudly when it is missing: the product path has
no CPU fallback."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libophelia_hip.so")
SOURCES = ["oph_kernels.hip", "oph_planegemm.hip", "oph_decrun.hip", "oph_decchain.hip", "oph_hcfused.hip", "oph_conehead.hip",
           "oph_pack.hip", "oph_model.hip", "oph_nets.hip", "oph_cone.hip", "oph_decode.hip", "oph_api.hip", "oph_ops.hip"]

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


class OphDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab", "e", "d", "c", "n_mels", "full_dim", "r", "max_N", "max_T",
        "attention_win_size", "nspeakers", "speaker_embedding_size", "flags")]


FLAG_SPK_AUDIO_DECODER_INPUT = 1
FLAG_NORM_NONE = 2
FLAG_NO_MONOTONIC = 4
FLAG_SPK_TEXT_ENCODER_INPUT = 8
FLAG_SPK_TEXT_ENCODER_TOWARDS_END = 16
FLAG_LCC = 32
FLAG_SPK_AUDIO_ENCODER_INPUT = 64
FLAG_NO_CONCAT_QUERY = 128
FLAG_NO_SQUASH_T2M = 256
FLAG_NO_SQUASH_SSRN = 512
FLAG_SPK_SSRN_INPUT = 1024
STOP_REFERENCE, STOP_NEVER = 0, 1

# name -> (restype, argtypes); every symbol declared in include/ophelia_hip.h
SIGNATURES = {
    "oph_abi_version": (C.c_int, []),
    "oph_create": (C.c_int, [C.POINTER(OphDims), C.c_int, C.POINTER(C.c_void_p)]),
    "oph_create_opts": (C.c_int, [C.POINTER(OphDims), C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "oph_destroy": (C.c_int, [C.c_void_p]),
    "oph_last_error": (C.c_char_p, [C.c_void_p]),
    "oph_num_weights": (C.c_int, [C.c_void_p]),
    "oph_weight_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i64p, C.POINTER(C.c_int)]),
    "oph_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, c_f32p, c_i64p, C.c_int]),
    "oph_set_weights_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "oph_finalize_weights": (C.c_int, [C.c_void_p]),
    "oph_encode_text": (C.c_int, [C.c_void_p, c_i32p, c_i32p, C.c_int, c_f32p, c_f32p]),
    "oph_text2mel": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_i32p, c_i32p, C.c_int, C.c_int,
                               c_f32p, c_i32p, c_f32p, c_i32p]),
    "oph_text2mel_durations": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, C.c_int, C.c_int,
                                         c_f32p, c_i32p, c_f32p, c_i32p]),
    "oph_ssrn": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.c_int, c_f32p]),
    "oph_text2mel_graph": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_i32p, C.c_int,
                                     c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p]),
    "oph_stage_text": (C.c_int, [C.c_void_p, c_i32p, c_i32p, c_i32p, C.c_int]),
    "oph_stage_text_next": (C.c_int, [C.c_void_p, c_i32p, c_i32p, c_i32p, C.c_int]),
    "oph_run_host": (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_i32p]),
    "oph_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "oph_host_free": (C.c_int, [C.c_void_p]),
    "oph_ssrn_speakers": (C.c_int, [C.c_void_p, c_f32p, c_i32p, C.c_int, C.c_int, c_f32p, c_f32p]),
    "oph_ssrn_logits": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.c_int, c_f32p, c_f32p]),
    "oph_set_streaming": (C.c_int, [C.c_void_p, C.c_int]),
    "oph_set_mag_destination": (C.c_int, [C.c_void_p, c_f32p]),
    "oph_get_counters": (C.c_int, [C.c_void_p, c_i64p, C.c_int]),
    "oph_run_resident": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_i32p]),
    "oph_decode_steps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_i32p]),
    "oph_run_ssrn_resident": (C.c_int, [C.c_void_p]),
    "oph_fetch_kv": (C.c_int, [C.c_void_p, c_f32p, c_f32p]),
    "oph_fetch_mel": (C.c_int, [C.c_void_p, c_f32p, c_i32p, c_f32p]),
    "oph_fetch_mag": (C.c_int, [C.c_void_p, c_f32p]),
    "oph_synchronize": (C.c_int, [C.c_void_p]),
    "oph_device_mag": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i64p, c_i32p]),
    "oph_set_ssrn_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "oph_set_precision": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "oph_timer_start": (C.c_int, [C.c_void_p]),
    "oph_timer_stop": (C.c_int, [C.c_void_p, c_f32p]),
    "oph_loop_clock": (C.c_int, [C.c_void_p, c_i64p, c_f64p, C.c_int]),
    "oph_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "oph_profile_reset": (C.c_int, [C.c_void_p]),
    "oph_profile_count": (C.c_int, [C.c_void_p]),
    "oph_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_i64p, c_f64p, c_f64p, c_f64p]),
    "oph_op_embed": (C.c_int, [C.c_int, c_i32p, C.c_int64, c_f32p, C.c_int, C.c_int, c_f32p]),
    "oph_op_layernorm": (C.c_int, [C.c_int, c_f32p, C.c_int64, C.c_int, c_f32p, c_f32p, c_f32p]),
    "oph_op_conv1d": (C.c_int, [C.c_int, c_f32p] + [C.c_int] * 7 + [c_f32p] * 4 + [C.c_int, c_f32p]),
    "oph_op_hc": (C.c_int, [C.c_int, c_f32p] + [C.c_int] * 6 + [c_f32p] * 7),
    "oph_op_conv1d_transpose": (C.c_int, [C.c_int, c_f32p] + [C.c_int] * 4 + [c_f32p] * 5),
    "oph_op_conv1d_transpose_prec": (C.c_int, [C.c_int, c_f32p] + [C.c_int] * 4 + [c_f32p] * 4 + [C.c_int, c_f32p]),
    "oph_op_attention": (C.c_int, [C.c_int, c_f32p, c_f32p, c_f32p, c_i32p] + [C.c_int] * 5 + [c_f32p, c_f32p, c_i64p]),
    "oph_bench_conv1d_transpose": (C.c_int, [C.c_int] * 8 + [c_f64p, c_f64p, c_f64p]),
    "oph_op_last_error": (C.c_char_p, []),
}



class OphGLParams(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop_length", C.c_int32), ("win_length", C.c_int32), ("n_iter", C.c_int32),
                ("power", C.c_double), ("preemphasis", C.c_double), ("max_db", C.c_double), ("ref_db", C.c_double)]


# every symbol declared in include/ophelia_vocoder.h
VOCODER_LIBPATH = os.path.join(LIBDIR, "libophelia_vocoder.so")
VOCODER_SOURCES = ["oph_vocoder.hip"]
VOCODER_SIGNATURES = {
    "oph_vocoder_abi_version": (C.c_int, []),
    "oph_vocoder_create": (C.c_int, [C.POINTER(OphGLParams), C.c_int, C.POINTER(C.c_void_p)]),
    "oph_vocoder_destroy": (C.c_int, [C.c_void_p]),
    "oph_vocoder_last_error": (C.c_char_p, [C.c_void_p]),
    "oph_spectrogram2wav": (C.c_int, [C.c_void_p, c_f32p, c_i32p, C.c_int, c_f32p]),
    "oph_spectrogram2wav_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_i32p, C.c_int, c_f32p]),
    "oph_vocoder_griffin_lim": (C.c_int, [C.c_void_p, c_f32p, c_i32p, C.c_int, C.c_int, c_f32p]),
    "oph_vocoder_stft": (C.c_int, [C.c_void_p, c_f32p, C.c_int64, c_f32p]),
    "oph_vocoder_istft": (C.c_int, [C.c_void_p, c_f32p, C.c_int, c_f32p]),
    "oph_vocoder_deemphasis": (C.c_int, [C.c_void_p, c_f32p, C.c_int64, c_f32p]),
    "oph_vocoder_set_backend": (C.c_int, [C.c_void_p, C.c_int]),
    "oph_vocoder_last_device_ms": (C.c_int, [C.c_void_p, c_f32p]),
}

_lib = None
_vlib = None


class OpheliaHipError(RuntimeError):
    pass


def libpath():
    """The library this process builds and loads.  OPH_HIPCC_FLAGS (measurement builds, e.g. -DOPH_ABLATE) selects a file of its
    own, named after the flags: a measurement build never replaces the production library, and is itself up to date when its
    sources are older (every rank of a multi-rank run finds it built)."""
    more = os.environ.get("OPH_HIPCC_FLAGS", "").split()
    if not more:
        return LIBPATH
    import hashlib
    return os.path.join(LIBDIR, "libophelia_hip.%s.so" % hashlib.md5(" ".join(more).encode()).hexdigest()[:8])


def _hipcc_shared(out, srcs, deps, extra, verbose):
    # OPH_HIPCC_FLAGS: extra compiler flags (measurement builds); they go into a file of their own (libpath())
    more = os.environ.get("OPH_HIPCC_FLAGS", "").split() if os.path.basename(out).startswith("libophelia_hip") else []
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    extra = list(extra) + more
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # several ranks may find the library stale at the same time: each builds into its own temporary file and renames
    # it into place (atomic on one filesystem), so a reader never maps a half-written library
    tmp = "%s.%d.tmp" % (out, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result"] + srcs + extra + ["-o", tmp]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise OpheliaHipError("hipcc failed building %s" % os.path.basename(out))
    os.replace(tmp, out)
    return out


def build(verbose=False):
    """Compile the HIP extensions for gfx950 in-tree (cross-compiles without a GPU)."""
    os.makedirs(LIBDIR, exist_ok=True)
    inc = os.path.join(os.path.dirname(HERE), "include")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "oph_internal.h"), os.path.join(CSRC, "oph_device.h"), os.path.join(CSRC, "oph_loopdev.h"), os.path.join(CSRC, "oph_host.h"),
            os.path.join(inc, "ophelia_hip.h")]
    _hipcc_shared(libpath(), srcs, srcs + hdrs, ["-fvisibility=hidden"], verbose)
    vsrcs = [os.path.join(CSRC, s) for s in VOCODER_SOURCES]
    _hipcc_shared(VOCODER_LIBPATH, vsrcs, vsrcs + [os.path.join(inc, "ophelia_vocoder.h")],
                  ["-L/opt/rocm/lib", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"], verbose)
    return libpath()


def load():
    """Load the shared library and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = libpath()
    if not os.path.exists(path):
        raise OpheliaHipError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.oph_abi_version() != 1:
        raise OpheliaHipError("ABI version mismatch")
    _lib = lib
    return lib


def load_vocoder():
    """Load libophelia_vocoder.so (Griffin-Lim).  Raises if it is not built."""
    global _vlib
    if _vlib is not None:
        return _vlib
    if not os.path.exists(VOCODER_LIBPATH):
        raise OpheliaHipError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % VOCODER_LIBPATH)
    lib = C.CDLL(VOCODER_LIBPATH)
    for name, (res, args) in VOCODER_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.oph_vocoder_abi_version() != 1:
        raise OpheliaHipError("vocoder ABI version mismatch")
    _vlib = lib
    return lib


def fptr(a):
    return a.ctypes.data_as(c_f32p)


def iptr(a):
    return a.ctypes.data_as(c_i32p)
