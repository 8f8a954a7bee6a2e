"""The two scans of the reference's roc_functions.py that run over whole score arrays -- the step
right after the scoring path -- on the GPU:

    get_max_detection_in_syllable_windows   :88-139
    get_max_above_threshold                 :177-187
    get_cluster_starts_ends / perform_detection_time_clustering   :189-221
    _get_detect_clusters_single_threshold / get_detect_clusters_threshold_array   :356-398

and the clustering of the peak times (cluster_times.pyx) behind them: one call turns a scored corpus
into the clustered (start, end) detections of every utterance.  The ROC bookkeeping downstream stays
on the host.
"""
import numpy as np

from tsr_hip import api


def get_max_above_threshold(detection_vals, detection_length, threshold, C0):
    """:177-187 -- indices (uint16, ascending) of the local maxima >= threshold among the first
    l = detection_length - C0 + 1 scores of one utterance."""
    detection_vals = np.asarray(detection_vals, dtype=np.float32)
    T = int(detection_length)
    if T > detection_vals.shape[0]:
        raise ValueError("detection_length exceeds the score row")
    out = api.peaks(threshold, C0, vals=detection_vals[:T], offsets=[0, T])[0]
    return out.astype(np.uint16)


def get_peaks_above_threshold(detection_array, detection_lengths, threshold, C0):
    """get_max_above_threshold for every row of a detection array in one launch (the loop of
    _get_detect_clusters_single_threshold, :371-383) -> tuple of uint16 index arrays."""
    detection_array = np.asarray(detection_array, dtype=np.float32)
    lengths = np.asarray(detection_lengths, dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    vals = np.concatenate([detection_array[n, :lengths[n]] for n in range(len(lengths))]) if len(lengths) \
        else np.zeros(0, np.float32)
    return tuple(p.astype(np.uint16) for p in api.peaks(threshold, C0, vals=vals, offsets=offsets))


def get_max_detection_in_syllable_windows(detection_array, example_start_end_times, detection_lengths,
                                          window_start, window_end, verbose=False,
                                          return_argsort_idx=False):
    """:88-139 -- for every labelled example the largest local maximum of the detection scores inside
    its detection window [start+window_start, start+window_end) (clipped to the array), -inf if the
    window holds none; returned sorted, like the reference (optionally with the argsort and the
    utterance id of every example).  `detection_array` is the zero-padded [N, width] array of the
    scoring loop; all windows are reduced in one launch."""
    detection_array = np.ascontiguousarray(detection_array, dtype=np.float32)
    N, width = detection_array.shape
    utt, starts, ends = [], [], []
    for example_idx, start_end_times in enumerate(example_start_end_times):
        for start_time, end_time in start_end_times:
            utt.append(example_idx)
            starts.append(max(start_time + window_start, 0))
            ends.append(min(start_time + window_end, width))
    offsets = np.arange(N + 1, dtype=np.int64) * width
    max_detect_vals = api.window_peak_max(utt, starts, ends, vals=detection_array, offsets=offsets)
    utterance_ids = np.array(utt)
    if return_argsort_idx:
        return np.sort(max_detect_vals), np.argsort(max_detect_vals), utterance_ids
    return np.sort(max_detect_vals)


def get_cluster_starts_ends(potential_detect_times, cluster_partition):
    """:189-210 -- (first time, last time + 1) of every cluster of the partition."""
    d, t, cluster_tuples = len(potential_detect_times), 0, []
    while t < d:
        cur_cluster_start = potential_detect_times[t]
        t += 1
        while cluster_partition[t] != 2:
            t += 1
        cluster_tuples.append((cur_cluster_start, potential_detect_times[t - 1] + 1))
    return tuple(cluster_tuples)


def perform_detection_time_clustering(potential_detect_times, C0, C1):
    """:212-221 -- potential_detect_times: uint16, strictly increasing."""
    from . import cluster_times
    return get_cluster_starts_ends(potential_detect_times,
                                   cluster_times.cluster_times(potential_detect_times, C0, C1))


def _get_detect_clusters_single_threshold(threshold, detection_array, detection_lengths, C0, C1):
    """:356-383 -- for every row of the detection array the clustered detections above the threshold, as a
    tuple of (start, end) tuples; peaks and clusters of all utterances in one launch each."""
    detection_array = np.asarray(detection_array, dtype=np.float32)
    lengths = np.asarray(detection_lengths, dtype=np.int64)
    offsets = np.concatenate(([0], np.cumsum(lengths)))
    vals = np.concatenate([detection_array[n, :lengths[n]] for n in range(len(lengths))]) if len(lengths) \
        else np.zeros(0, np.float32)
    return api.detect_clusters(threshold, C0, C1, vals=vals, offsets=offsets)


def get_detect_clusters_threshold_array(thresholds, detection_array, detection_lengths, C0, C1):
    """:385-398."""
    return tuple(_get_detect_clusters_single_threshold(threshold, detection_array, detection_lengths, C0, C1)
                 if threshold > -np.inf else () for threshold in thresholds)
