#!/usr/bin/env python3
"""Golden vectors for the detection clustering: g14 = cluster_times.cluster_times
(cluster_times.pyx:133-222), g16 = cluster_times.cluster_times_template_lengths (:221-310) and g17 =
cluster_times.cluster_times_variable_length (:73-131), all run by the
reference's own compiled module (oracle/_ref) on seeded peak-time lists.
Build container only:   python tests/golden/make_golden_cluster.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

assert build_ref.build(), "oracle/_ref could not be built (no /root/reference?)"
ct = build_ref.load_extra("cluster_times")
rng = np.random.default_rng(1400)
d = {}
n_cases = 60
for i in range(n_cases):
    n = int(rng.integers(0, 30)) if i > 2 else i          # includes the empty and the one-peak list
    T = int(rng.integers(n + 2, 500))
    times = np.sort(rng.choice(np.arange(1, T), size=n, replace=False)).astype(np.uint16)
    C0, C1 = float(rng.integers(1, 45)), float(rng.integers(1, 90))
    d["t%d" % i], d["C%d" % i] = times, np.array([C0, C1], np.float32)
    d["p%d" % i] = ct.cluster_times(times, C0, C1)
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g14_cluster_times.npz")
np.savez_compressed(path, **d)
print("g14_cluster_times %.1f KB" % (os.path.getsize(path) / 1024))

# g16: per-detection template lengths
rng = np.random.default_rng(1600)
d = {}
for i in range(n_cases):
    n = int(rng.integers(0, 30)) if i > 2 else i
    T = int(rng.integers(n + 2, 500))
    times = np.sort(rng.choice(np.arange(1, T), size=n, replace=False)).astype(np.uint16)
    nl = int(rng.integers(1, 7))
    lengths = rng.integers(1, 60, size=nl).astype(np.uint16)
    comps = rng.integers(0, nl, size=n).astype(np.uint16)
    d["t%d" % i], d["l%d" % i], d["c%d" % i] = times, lengths, comps
    d["p%d" % i] = ct.cluster_times_template_lengths(times, lengths, comps)
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g16_cluster_times_template_lengths.npz")
np.savez_compressed(path, **d)
print("g16_cluster_times_template_lengths %.1f KB" % (os.path.getsize(path) / 1024))

# g17: cluster_times_variable_length (n >= 1: the reference writes partition[0] of an empty array)
rng = np.random.default_rng(1700)
d = {}
for i in range(n_cases):
    n = int(rng.integers(1, 30)) if i > 1 else i + 1
    T = int(rng.integers(n + 2, 500))
    times = np.sort(rng.choice(np.arange(1, T), size=n, replace=False)).astype(np.intc)
    lengths = rng.integers(1, 60, size=n).astype(np.intc)
    d["t%d" % i], d["l%d" % i] = times, lengths
    d["p%d" % i] = ct.cluster_times_variable_length(times, lengths)
d["n"] = np.int64(n_cases)
path = os.path.join(HERE, "g17_cluster_times_variable_length.npz")
np.savez_compressed(path, **d)
print("g17_cluster_times_variable_length %.1f KB" % (os.path.getsize(path) / 1024))
