"""tsr_hip -- MI355X-native sliding-window template scoring (host side).

NumPy + ctypes over libtsr_hip.so (include/tsr.h).  The reference's own module
names live next door in `template_speech_rec/`; this package holds the batched,
device-resident API they are built on.
"""
from ._lib import (TSR_EXACT, TSR_FAST, TSR_W24, TSR_BF16, TSR_MX, TSR_PARTS_AUTO, TSR_PARTS_VALU, TSR_PARTS_MFMA,  # noqa: F401
                   TSR_OPT_FAST_KERNEL, TSR_OPT_EXACT_KERNEL, TSR_OPT_PART_LAYOUT, TSR_OPT_XCD_MAP, TSR_OPT_AUTO,
                   TSR_FAST_KERNEL_TAP2, TSR_FAST_KERNEL_INTERLEAVED, TSR_EXACT_KERNEL_VALU, TSR_EXACT_KERNEL_MFMA,
                   TSR_PART_LAYOUT_BYTES, TSR_PART_LAYOUT_BITS, TSR_INFO_EXACT_MFMA, TSR_INFO_LAST_BANK_KERNEL,
                   TSR_KERNEL_EXACT_VALU, TSR_KERNEL_EXACT_MFMA, TSR_KERNEL_INTERLEAVED, TSR_KERNEL_TAP2, TSR_KERNEL_MX, TsrError, LIB_PATH)
from .api import (Bank, Context, Corpus, Parts, default_ctx, detect_clusters, device_count,  # noqa: F401
                  pack_filters, peaks, pool_trim, unique_id, window_peak_max)
