"""ctypes binding of libbfir_hip.so (the C ABI declared in include/bfir_hip.h).

There is no fallback: if the library is missing or does not load, importing
the binding raises."""
import ctypes as C
import os

from . import _build

# codes from include/bfir_hip.h
SAMPLE_FORMAT_FLOAT_LE = 8
SAMPLE_FORMAT_FLOAT64_LE = 10
MIXMODE_INPUT, MIXMODE_INPUT_ADD, MIXMODE_OUTPUT = 1, 2, 3
OK, ERR_NONFINITE, ERR_COEFF, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED, ERR_IO = (
    0, -1, -2, -3, -4, -5, -6, -7, -8)
K_STAGE_IN, K_FWD, K_MAC, K_INV, K_STAGE_OUT = range(5)
KERNEL_NAMES = ("k_stage_in", "k_fwd", "k_mac", "k_inv", "k_stage_out")


class Overflow(C.Structure):
    """bfir_overflow == bfoverflow_t (brutefir/global.h:96-102)."""
    _fields_ = [("n_overflows", C.c_uint), ("intlargest", C.c_int32),
                ("largest", C.c_double), ("max", C.c_double)]


class SampleFormat(C.Structure):
    """bfir_sample_format == sample_format_t (brutefir/global.h:39-47)."""
    _fields_ = [("isfloat", C.c_bool), ("swap", C.c_bool), ("bytes", C.c_int),
                ("sbytes", C.c_int), ("scale", C.c_double), ("format", C.c_int)]


class BufferFormat(C.Structure):
    """bfir_buffer_format == buffer_format_t (brutefir/global.h:49-54)."""
    _fields_ = [("sf", SampleFormat), ("sample_spacing", C.c_int), ("byte_offset", C.c_int)]


class DitherState(C.Structure):
    """bfir_dither_state == dither_state_t (brutefir/global.h:63-69)."""
    _fields_ = [("randtab_ptr", C.c_int), ("randtab", C.POINTER(C.c_int8)), ("sf", C.c_float * 2),
                ("sd", C.c_double * 2)]


LOG_FN = C.CFUNCTYPE(None, C.c_char_p)

_vp, _ci, _cd, _cl = C.c_void_p, C.c_int, C.c_double, C.c_int64   # 64-bit fields are int64_t in the header, never `long`
_pi = C.POINTER(C.c_int)

# every symbol include/bfir_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bfir_set_log_callback": (None, [LOG_FN]),
    "bfir_strerror": (C.c_char_p, [_ci]),
    "bfir_device_count": (_ci, []),
    "bfir_version": (C.c_char_p, []),
    "bfir_engine_create": (_vp, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _pi]),
    "bfir_engine_create_batch": (_vp, [_ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _pi]),
    "bfir_engine_destroy": (None, [_vp]),
    "bfir_engine_is_initialized": (_ci, [_vp]),
    "bfir_engine_set_coeff": (_ci, [_vp, C.POINTER(_vp), _ci, _ci, _ci, _cd]),
    "bfir_engine_set_coeff_at": (_ci, [_vp, _ci, C.POINTER(_vp), _ci, _ci, _ci, _cd]),
    "bfir_engine_run": (_ci, [_vp, _vp, _vp, _ci]),
    "bfir_engine_run_device": (_ci, [_vp, _vp, _cl, _vp, _cl, _ci, _vp]),
    "bfir_engine_sync": (_ci, [_vp]),
    "bfir_engine_reset": (None, [_vp]),
    "bfir_engine_get_overflow": (_ci, [_vp, _ci, C.POINTER(Overflow)]),
    "bfir_engine_set_chunk": (_ci, [_vp, _ci]),
    "bfir_engine_set_profiling": (_ci, [_vp, _ci]),
    "bfir_engine_get_profile": (_ci, [_vp, _ci, C.POINTER(_cd), C.POINTER(_cl)]),
    "bfir_engine_read_coeff": (_ci, [_vp, _ci, _ci, _vp]),
    "bfir_convolver_create": (_vp, [_ci, _ci, _ci, _pi]),
    "bfir_convolver_destroy": (None, [_vp]),
    "bfir_convolver_cbufsize": (_ci, [_vp]),
    "bfir_convolver_raw2cbuf": (_ci, [_vp, _vp, _vp, _vp, C.POINTER(BufferFormat)]),
    "bfir_convolver_time2freq": (_ci, [_vp, _vp, _vp]),
    "bfir_convolver_mixnscale": (_ci, [_vp, C.POINTER(_vp), _vp, C.POINTER(_cd), _ci, _ci]),
    "bfir_convolver_convolve_inplace": (_ci, [_vp, _vp, _vp]),
    "bfir_convolver_convolve": (_ci, [_vp, _vp, _vp, _vp]),
    "bfir_convolver_convolve_add": (_ci, [_vp, _vp, _vp, _vp]),
    "bfir_convolver_freq2time": (_ci, [_vp, _vp, _vp]),
    "bfir_convolver_cbuf2raw": (_ci, [_vp, _vp, _vp, C.POINTER(BufferFormat), C.POINTER(Overflow)]),
    "bfir_dither_create": (_vp, [_ci, _ci, _ci, _ci, _ci, C.POINTER(DitherState), _ci, _pi]),
    "bfir_dither_destroy": (None, [_vp]),
    "bfir_dither_table_size": (_ci, [_vp]),
    "bfir_dither_table": (C.POINTER(C.c_int8), [_vp]),
    "bfir_dither_preloop_real2int_hp_tpdf": (None, [_vp, C.POINTER(DitherState), _ci]),
    "bfir_convolver_cbuf2raw_dither": (_ci, [_vp, _vp, _vp, _vp, C.POINTER(BufferFormat), C.POINTER(DitherState),
                                             C.POINTER(Overflow)]),
    "bfir_convolver_coeffs2cbuf": (_vp, [_vp, _vp, _ci, _cd, _vp]),
    "bfir_convolver_runtime_coeffs2cbuf": (_ci, [_vp, _vp, _vp]),
    "bfir_convolver_dirac_convolve": (_ci, [_vp, _vp, _vp]),
    "bfir_convolver_dirac_convolve_inplace": (_ci, [_vp, _vp]),
    "bfir_convolver_convolve_eval": (_ci, [_vp, _vp, _vp, _vp]),
    "bfir_convolver_crossfade_inplace": (_ci, [_vp, _vp, _vp, _vp]),
    "bfir_convolver_verify_cbuf": (_ci, [_vp, C.POINTER(_vp), _ci]),
    "bfir_convolver_debug_dump_cbuf": (_ci, [_vp, C.c_char_p, C.POINTER(_vp), _ci]),
    "bfir_td_block_length": (_ci, [_ci]),
    "bfir_td_new": (_vp, [_vp, _ci, _ci, _ci, _pi]),
    "bfir_td_destroy": (None, [_vp]),
    "bfir_td_blocklen": (_ci, [_vp]),
    "bfir_td_coeffs": (_vp, [_vp]),
    "bfir_td_convolve": (_ci, [_vp, _vp]),
    "bfir_fft_plan_create": (_vp, [_ci, _ci, _ci, _ci, _ci, _pi]),
    "bfir_fft_plan_destroy": (None, [_vp]),
    "bfir_fft_plan_execute": (_ci, [_vp, _vp, _vp]),
    "bfir_fft_plan_length": (_cl, [_vp]),
    "bfir_equalizer_render": (_ci, [_vp, _ci, C.POINTER(_cd), C.POINTER(_cd), C.POINTER(_cd), _vp]),
    "bfir_pinned_malloc": (_vp, [C.c_size_t]),
    "bfir_pinned_free": (None, [_vp]),
    "bfir_aligned_malloc": (_vp, [C.c_size_t, C.c_size_t]),
    "bfir_aligned_free": (None, [_vp]),
}

_lib = None


def library_path():
    # BFIR_LIB_OVERRIDE: load another build of the library (A/B timing of two builds in one gpurun call)
    return os.environ.get("BFIR_LIB_OVERRIDE") or _build.LIB


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the
    same SONAME as /opt/rocm's).  If libbfir_hip.so is loaded first it binds the
    system copy, a later `import torch` maps the bundled copy as a second
    runtime, and that one finds no GPU.  Mapping torch's copy first (without
    importing torch) makes both resolve to the same runtime, so device
    pointers, streams and synchronisation are shared.  BFIR_HIP_RUNTIME=system
    skips this (for processes that never import torch)."""
    import sys
    if os.environ.get("BFIR_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # no torch, or an unusual install: the system runtime is used


def load():
    """dlopen libbfir_hip.so and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libbfir_hip.so is not built (%s); run __graft_entry__.build() -- "
            "there is no CPU fallback for the convolution engine" % path)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class BfirError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = load().bfir_strerror(code).decode()
        super().__init__("%s%s (%d)" % (what + ": " if what else "", msg, code))
