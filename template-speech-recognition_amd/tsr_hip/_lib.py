"""ctypes binding of libtsr_hip.so (include/tsr.h).  No CPU fallback: if the
library or a gfx950 device is missing, calls raise."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("TSR_HIP_LIB") or os.path.join(_PKG, "lib", "libtsr_hip.so")

TSR_OK, TSR_ERR_INVALID, TSR_ERR_HIP, TSR_ERR_NO_DEVICE, TSR_ERR_UNSUPPORTED, TSR_ERR_NONFINITE, \
    TSR_ERR_RCCL = range(7)
TSR_EXACT, TSR_FAST, TSR_W24, TSR_BF16, TSR_MX = 0, 1, 2, 4, 8
TSR_PARTS_AUTO, TSR_PARTS_VALU, TSR_PARTS_MFMA = 0, 1, 2
# context options (include/tsr.h: tsr_ctx_set_option)
TSR_OPT_FAST_KERNEL, TSR_OPT_EXACT_KERNEL, TSR_OPT_PART_LAYOUT, TSR_OPT_XCD_MAP = 1, 2, 3, 4
TSR_OPT_AUTO = 0
TSR_INFO_EXACT_MFMA = 5
TSR_INFO_LAST_BANK_KERNEL = 6
TSR_KERNEL_EXACT_VALU, TSR_KERNEL_EXACT_MFMA, TSR_KERNEL_INTERLEAVED, TSR_KERNEL_TAP2, TSR_KERNEL_MX = 1, 2, 3, 4, 5
TSR_FAST_KERNEL_TAP2, TSR_FAST_KERNEL_INTERLEAVED = 1, 2
TSR_EXACT_KERNEL_VALU, TSR_EXACT_KERNEL_MFMA = 1, 2
TSR_PART_LAYOUT_BYTES, TSR_PART_LAYOUT_BITS = 1, 2

_i32, _i64, _int, _p, _f = C.c_int32, C.c_int64, C.c_int, C.c_void_p, C.c_float
_pp = C.POINTER(C.c_void_p)

# name -> argtypes, exactly the declarations of include/tsr.h (restype is int unless noted)
SIGNATURES = {
    "tsr_version": [],
    "tsr_pool_trim": [],
    "tsr_device_count": [C.POINTER(_int)],
    "tsr_ctx_create": [_int, _pp],
    "tsr_ctx_destroy": [_p],
    "tsr_ctx_sync": [_p],
    "tsr_ctx_set_option": [_p, _int, _int],
    "tsr_ctx_get_option": [_p, _int, C.POINTER(_int)],
    "tsr_timer_start": [_p],
    "tsr_timer_stop": [_p, C.POINTER(_f)],
    "tsr_device_info": [_p, C.c_char_p, _int, C.POINTER(_int), C.POINTER(_i64)],
    "tsr_detect_u8": [_p, _p, _i64, _i64, _p, _i64, _p],
    "tsr_detect_f32": [_p, _p, _i64, _i64, _p, _i64, _p],
    "tsr_detect_f32_max": [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p],
    "tsr_max_detect_examples_u8": [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _p],
    "tsr_bank_create": [_p, _p, _p, _p, _i32, _i64, _pp],
    "tsr_bank_destroy": [_p],
    "tsr_bank_info": [_p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64)],
    "tsr_score_bank_u8": [_p, _p, _p, _p, _i64, _int, _p, _p],
    "tsr_score_bank_all_u8": [_p, _p, _p, _p, _i64, _int, _p],
    "tsr_corpus_create": [_p, _p, _p, _i64, _i64, _pp],
    "tsr_corpus_destroy": [_p],
    "tsr_corpus_score": [_p, _p, _p, _int],
    "tsr_corpus_score_all": [_p, _p, _p, _int, _p],
    "tsr_corpus_download": [_p, _p, _p, _p],
    "tsr_corpus_frames": [_p, C.POINTER(_i64)],
    "tsr_code_parts": [_p, _p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _int, _int, _p],
    "tsr_code_parts_mag_aux": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p],
    "tsr_spread_patches": [_p, _p, _i64, _i64, _int, _int, _i64, _p],
    "tsr_spread_graph_patches": [_p, _p, _i64, _i64, _i64, _p],
    "tsr_corpus_classify_windows": [_p, _p, _p, _p, _p, _i64, _int, _p, _p, _p],
    "tsr_classify_float_windows": [_p, _p, _p, _p, _p, _i64, _p, _p, _p],
    "tsr_corpus_classify_isolated": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p],
    "tsr_peaks": [_p, _p, _p, _p, _i64, _f, _int, _p, _p],
    "tsr_detect_clusters": [_p, _p, _p, _p, _i64, _f, _f, _f, _p, _p, _p, _p, _p],
    "tsr_cluster_times": [_p, _p, _i64, _f, _f, _p],
    "tsr_cluster_times_template_lengths": [_p, _p, _p, _i64, _p, _i64, _p],
    "tsr_cluster_times_variable_length": [_p, _p, _p, _i64, _p],
    "tsr_window_peak_max": [_p, _p, _p, _p, _i64, _p, _p, _p, _i64, _p],
    "tsr_edge_corpus_create": [_p, _p, _p, _i64, _i64, _int, _int, C.c_double, _p, _pp],
    "tsr_edge_corpus_create_masked": [_p, _p, _p, _p, _i64, _i64, _int, _int, C.c_double, _p, _pp],
    "tsr_threshold_edgemap": [_p, _p, _p, _i32, _i64, C.c_double, _p, _i32, _p],
    "tsr_spread_edgemap": [_p, _p, _p, _i32, _i64, _p, _i32],
    "tsr_magnitude_features": [_p, _p, _i64, _i64, _i32, _i32, C.c_double, _i32, _i32, _i32, _i32, _p],
    "tsr_corpus_weighted_sums": [_p, _p, _p, _i64, _p],
    "tsr_mixture_scores": [_p, _p, _p, _p, _i32, _p],
    "tsr_mixture_e_step": [_p, _p, _p, _p, _i32, _p, _i32, C.POINTER(C.c_double), _p],
    "tsr_mixture_m_step": [_p, _p, _p, _p],
    "tsr_corpus_download_affinities": [_p, _p, _p],
    "tsr_parts_create": [_p, _p, _p, _i64, _i64, _i64, _i64, _pp],
    "tsr_parts_destroy": [_p],
    "tsr_parts_set_mode": [_p, _int],
    "tsr_parts_info": [_p, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_i64)],
    "tsr_part_features": [_p, _p, _p, _i64, _i64, _int, _int, _int, _int, _p, _int, _int, _p, _p],
    "tsr_corpus_part_features": [_p, _p, _p, _int, _int, _int, _int, _p, _int, _int, _pp],
    "tsr_corpus_download_feats": [_p, _p, _p],
    "tsr_corpus_shape": [_p, C.POINTER(_i64), C.POINTER(_i64), _p],
    "tsr_corpus_layout": [_p, C.POINTER(_int), C.POINTER(_i64)],
    "tsr_dist_unique_id": [C.c_char_p],
    "tsr_dist_init": [_p, C.c_char_p, _int, _int],
    "tsr_dist_finalize": [_p],
    "tsr_bank_broadcast": [_p, _p, _int],
}
_STR_FUNCS = {"tsr_strerror": [_int], "tsr_last_error": []}

_lib = None


class TsrError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "libtsr_hip: %s" % message)
        self.status = status


def lib():
    """Load libtsr_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtsr_hip.so is missing at %s: build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _int
        for name, args in _STR_FUNCS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_char_p
        _lib = l
    return _lib


def check(status):
    """Map a C status to the Python exception the reference's callers expect."""
    if status == TSR_OK:
        return
    l = lib()
    msg = l.tsr_last_error().decode() or l.tsr_strerror(status).decode()
    if status == TSR_ERR_INVALID:
        raise ValueError(msg)
    raise TsrError(status, "%s [%s]" % (msg, l.tsr_strerror(status).decode()))


def ptr(a):
    return None if a is None else a.ctypes.data_as(_p)
