"""Drop-in for the reference's cluster_times.pyx (the detection clustering right after the scorer).

    cluster_times(times, C0, C1)   :133-222

`times` must be a uint16 array of strictly increasing peak times (what get_max_above_threshold returns);
the result is the reference's partition code array, uint8 [len(times) + 1].  The batched form --
every utterance's peaks and clusters in one call, nothing but (start, end) pairs leaving the device -- is
roc_functions._get_detect_clusters_single_threshold / tsr_hip.detect_clusters.
cluster_times_variable_length and cluster_times_template_lengths are not provided.
"""
import numpy as np

from tsr_hip import _lib
from tsr_hip.api import default_ctx
from ._buffers import buffer


def cluster_times(times, C0, C1):
    times = buffer(times, np.uint16, 1, "times", "uint16_t")
    out = np.empty(times.shape[0] + 1, np.uint8)
    _lib.check(_lib.lib().tsr_cluster_times(default_ctx()._h, _lib.ptr(times), times.shape[0], float(C0), float(C1),
                                            _lib.ptr(out)))
    return out
