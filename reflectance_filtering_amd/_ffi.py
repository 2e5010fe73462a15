"""ctypes binding of librf_hip.so -- the only way the Python host code reaches the GPU.

The C ABI is declared in include/reflectance_filtering.h.  PyTorch is used purely as the
device-buffer container (allocation, H2D/D2H copies, the current HIP stream); no torch op
computes anything on the path.  There is NO CPU fallback: if the library is missing, cannot
be loaded, or no GPU is visible, every entry point raises.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one library this package loads.  (A/B timing of two builds: tools/gf_c5_exp.py --lib PATH sets
# this attribute before the first call - announced on stderr, version checked; no environment
# variable redirects the shipped loader.)
LIB_PATH = os.path.join(_HERE, "librf_hip.so")

RF_OK, RF_E_BADARG, RF_E_UNSUPPORTED, RF_E_WORKSPACE, RF_E_HIP = 0, -1, -2, -3, -4
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = range(5)
BORDER_DEFAULT = BORDER_REFLECT_101
JBF_TRUE_DIVISION = 1
JBF_FORCE_GENERIC = 2
JBF_GREY_AS_BGR = 4
CNN_NPARAMS = 4513
CNN_NPACKED = 4673          # RF_CNN_NPACKED: floats of rf_cnn_pack_weights' output

EXPORTS = ("rf_version", "rf_last_error", "rf_shutdown", "rf_jbf_u8", "rf_gf_workspace_bytes",
           "rf_gf_u8", "rf_cnn_reflectance_u8", "rf_cnn_pack_weights",
           "rf_cnn_reflectance_packed_u8", "rf_colorize_workspace_bytes",
           "rf_colorize_srgb_u8", "rf_whdr_f32", "rf_jbf_f32_workspace_bytes", "rf_jbf_f32",
           "rf_gf_f32_workspace_bytes", "rf_gf_f32")

# include/reflectance_filtering_debug.h: test / benchmark switches, not part of the boundary
DEBUG_EXPORTS = ("rf_debug_option", "rf_debug_clock_probe", "rf_debug_build_info")
# switches that leave work out (wrong results, timing experiments only); all others keep the bytes
RESULT_CHANGING_OPTIONS = ("jbf_stage_only", "gf_exp_skip")

_lib = None
_lock = threading.Lock()


class RFError(RuntimeError):
    """A librf_hip.so entry point returned a negative code."""


def load_library():
    """dlopen librf_hip.so and declare the prototypes.  Raises if it was not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own libamdhip64 / libhsa-runtime64.  The device pointers and the
        # stream we are handed belong to THAT runtime instance, so it has to be in the process
        # before librf_hip.so resolves its libamdhip64.so.7 dependency (same SONAME -> shared).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RFError("%s not found: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C reflectance_filtering_amd/csrc` "
                          "(there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp, ci, cd, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
        lib.rf_version.argtypes = []
        lib.rf_version.restype = ci
        lib.rf_last_error.argtypes = []
        lib.rf_last_error.restype = ctypes.c_char_p
        lib.rf_shutdown.argtypes = []
        lib.rf_shutdown.restype = ci
        lib.rf_jbf_u8.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, cd, ci, ci, vp]
        lib.rf_jbf_u8.restype = ci
        lib.rf_gf_workspace_bytes.argtypes = [ci, ci, ci, ci, ci, ci]
        lib.rf_gf_workspace_bytes.restype = sz
        lib.rf_gf_u8.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, ci, vp, sz, vp]
        lib.rf_gf_u8.restype = ci
        lib.rf_cnn_reflectance_u8.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
        lib.rf_cnn_reflectance_u8.restype = ci
        lib.rf_cnn_pack_weights.argtypes = [vp, vp, vp]
        lib.rf_cnn_pack_weights.restype = ci
        lib.rf_cnn_reflectance_packed_u8.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
        lib.rf_cnn_reflectance_packed_u8.restype = ci
        u64 = ctypes.c_ulonglong
        lib.rf_colorize_workspace_bytes.argtypes = [ci]
        lib.rf_colorize_workspace_bytes.restype = sz
        lib.rf_colorize_srgb_u8.argtypes = [vp, vp, vp, vp, ci, ci, ci, u64, u64, vp, vp, sz, vp]
        lib.rf_colorize_srgb_u8.restype = ci
        lib.rf_jbf_f32_workspace_bytes.argtypes = [ci, ci]
        lib.rf_jbf_f32_workspace_bytes.restype = sz
        lib.rf_jbf_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, cd, ci, vp, sz, vp]
        lib.rf_jbf_f32.restype = ci
        lib.rf_gf_f32_workspace_bytes.argtypes = [ci, ci, ci, ci, ci, ci]
        lib.rf_gf_f32_workspace_bytes.restype = sz
        lib.rf_gf_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, ci, vp, sz, vp]
        lib.rf_gf_f32.restype = ci
        lib.rf_whdr_f32.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, cd, vp, vp]
        lib.rf_whdr_f32.restype = ci
        lib.rf_debug_option.argtypes = [ctypes.c_char_p, ci]
        lib.rf_debug_option.restype = ci
        lib.rf_debug_clock_probe.argtypes = [vp, ci, vp]
        lib.rf_debug_clock_probe.restype = ci
        lib.rf_debug_build_info.argtypes = []
        lib.rf_debug_build_info.restype = ctypes.c_char_p
        # RF_DEBUG_OPTIONS="name=value,...": preset the test / benchmark switches of
        # include/reflectance_filtering_debug.h for a whole process (timing experiments only).
        # Every preset is announced on stderr - loudly for the switches that change results.
        import sys
        for item in filter(None, os.environ.get("RF_DEBUG_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            name = name.strip()
            try:
                number = int(value or 1)
            except ValueError:
                raise RFError("RF_DEBUG_OPTIONS: %r is not an integer (option %r)" % (value, name))
            if lib.rf_debug_option(name.encode(), number) < 0:
                raise RFError("RF_DEBUG_OPTIONS: unknown debug option %r" % name)
            if name in RESULT_CHANGING_OPTIONS and number:
                sys.stderr.write("reflectance_filtering_amd: WARNING: RF_DEBUG_OPTIONS sets %s=%d, a "
                                 "TIMING-ONLY switch - results of this process are WRONG\n"
                                 % (name, number))
            else:
                sys.stderr.write("reflectance_filtering_amd: RF_DEBUG_OPTIONS sets %s=%d\n"
                                 % (name, number))
        _lib = lib
        return lib


class debug_options:
    """Context manager around rf_debug_option (include/reflectance_filtering_debug.h):
    ``with debug_options(gf_two_kernel=1): ...`` selects the alternative kernels that tests and
    timing tools compare with the default ones, and restores the previous values on exit."""

    def __init__(self, **opts):
        self.opts = opts
        self.prev = {}

    def __enter__(self):
        lib = load_library()
        for name, value in self.opts.items():
            old = lib.rf_debug_option(name.encode(), int(value))
            if old < 0:
                raise ValueError("unknown debug option %r" % name)
            self.prev[name] = old
        return self

    def __exit__(self, *exc):
        lib = load_library()
        for name, old in self.prev.items():
            lib.rf_debug_option(name.encode(), old)
        return False


def check(rc, what):
    if rc == RF_OK:
        return
    msg = load_library().rf_last_error().decode("utf-8", "replace")
    if rc in (RF_E_BADARG, RF_E_UNSUPPORTED):
        raise ValueError("%s: %s" % (what, msg))
    raise RFError("%s failed (%d): %s" % (what, rc, msg))


def require_gpu():
    """torch with a visible HIP device, or a loud failure (never a silent CPU path)."""
    import torch
    if not torch.cuda.is_available():
        raise RFError("no HIP device visible: reflectance_filtering_amd runs on MI355X only "
                      "and has no CPU fallback")
    return torch


def current_stream_ptr(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
