"""Drop-in for the reference's cluster_times.pyx (the detection clustering right after the scorer).

    cluster_times_variable_length(times, detection_lengths)                         :73-131
    cluster_times(times, C0, C1)                                                    :133-222
    cluster_times_template_lengths(times, template_lengths, template_components)    :221-310

`times` must be a uint16 array of strictly increasing peak times (what get_max_above_threshold returns);
the result is the reference's partition code array, uint8 [len(times) + 1].  The batched form --
every utterance's peaks and clusters in one call, nothing but (start, end) pairs leaving the device -- is
roc_functions._get_detect_clusters_single_threshold / tsr_hip.detect_clusters.
cluster_times_variable_length takes the reference's `int` buffers (C long under its Python 2 toolchain, C int
under Cython 3: int64 and int32 are both accepted here); its split loop never changes the partition
(split_cluster, :47-70, re-marks the end of the cluster it is asked to split), so what it returns -- and what is
computed here -- is the link partition: 1 where the gap to the previous detection is >= that detection's length.
An empty list returns an empty partition (the reference writes partition[0] of an empty array).
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


def cluster_times_template_lengths(times, template_lengths, template_components):
    times = buffer(times, np.uint16, 1, "times", "uint16_t")
    template_lengths = buffer(template_lengths, np.uint16, 1, "template_lengths", "uint16_t")
    template_components = buffer(template_components, np.uint16, 1, "template_components", "uint16_t")
    if template_components.shape[0] != times.shape[0]:
        raise ValueError("cluster_times_template_lengths: %d times but %d template_components"
                         % (times.shape[0], template_components.shape[0]))
    out = np.empty(times.shape[0] + 1, np.uint8)
    _lib.check(_lib.lib().tsr_cluster_times_template_lengths(
        default_ctx()._h, _lib.ptr(times), _lib.ptr(template_lengths), template_lengths.shape[0],
        _lib.ptr(template_components), times.shape[0], _lib.ptr(out)))
    return out


def cluster_times_variable_length(times, detection_lengths):
    def ints(a, name):
        a = np.asarray(a)
        if a.ndim != 1:
            raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
        if a.dtype not in (np.dtype(np.int64), np.dtype(np.int32)):
            raise ValueError("Buffer dtype mismatch, expected 'int' but got '%s'" % a.dtype.name)
        if a.size and (a.min() < -2**31 or a.max() >= 2**31):
            raise ValueError("cluster_times_variable_length: %s does not fit 32 bits" % name)
        return np.ascontiguousarray(a, np.int32)
    times, detection_lengths = ints(times, "times"), ints(detection_lengths, "detection_lengths")
    if detection_lengths.shape[0] != times.shape[0]:
        raise ValueError("cluster_times_variable_length: %d times but %d detection_lengths"
                         % (times.shape[0], detection_lengths.shape[0]))
    out = np.zeros(times.shape[0], np.uint8)
    _lib.check(_lib.lib().tsr_cluster_times_variable_length(default_ctx()._h, _lib.ptr(times), _lib.ptr(detection_lengths),
                                                            times.shape[0], _lib.ptr(out)))
    return out
