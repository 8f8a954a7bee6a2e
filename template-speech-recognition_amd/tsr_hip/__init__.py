"""tsr_hip -- MI355X-native sliding-window template scoring (host side).

NumPy + ctypes over libtsr_hip.so (include/tsr.h).  The reference's own module
names live next door in `template_speech_rec/`; this package holds the batched,
device-resident API they are built on.
"""
from ._lib import (TSR_EXACT, TSR_FAST, TSR_W24, TSR_BF16, TSR_PARTS_AUTO, TSR_PARTS_VALU, TSR_PARTS_MFMA,  # noqa: F401
                   TsrError, LIB_PATH)
from .api import (Bank, Context, Corpus, Parts, default_ctx, detect_clusters, device_count,  # noqa: F401
                  pack_filters, peaks, unique_id, window_peak_max)
