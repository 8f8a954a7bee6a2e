"""E-step of the Bernoulli mixture on the device vs NumPy (the reference's expression)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "template-speech-recognition_amd"))
import numpy as np
import template_speech_rec.bernoulli_mixture as bm
rng = np.random.default_rng(2)
N, shape, K = 4000, (32, 45, 8), 8
data = (rng.random((N,) + shape) < 0.25).astype(np.uint8)
mix = bm.BernoulliMixture(K, data, min_probability=0.05)
mix.M_step()
for _ in range(2):
    t0 = time.perf_counter(); got = mix.get_template_loglikelihoods(); t1 = time.perf_counter()
X = mix.data_mat.astype(np.float32)
t2 = time.perf_counter()
ref = np.dot(X, mix.log_templates.T) + np.dot(1 - X, mix.log_invtemplates.T)
t3 = time.perf_counter()
ref64 = X.astype(np.float64) @ mix.log_templates.astype(np.float64).T + (1 - X.astype(np.float64)) @ mix.log_invtemplates.astype(np.float64).T
print("E-step scoring, %d examples x D=%d x %d components: device %.2f ms (table upload + kernel + download), NumPy float32 %.1f ms"
      % (N, mix.data_length, K, (t1 - t0) * 1e3, (t3 - t2) * 1e3))
print("max relative deviation from the float64 value: device %.2e, NumPy float32 %.2e"
      % (np.max(np.abs(got - ref64) / np.abs(ref64)), np.max(np.abs(ref - ref64) / np.abs(ref64))))
mix._compute_loglikelihoods()
for _ in range(2):
    t0 = time.perf_counter(); ws = mix._corpus.weighted_sums(mix.affinities); t1 = time.perf_counter()
t2 = time.perf_counter(); ws_np = np.dot(mix.affinities.T, mix.data_mat).astype(np.float32); t3 = time.perf_counter()
ws64 = mix.affinities.astype(np.float64).T @ mix.data_mat.astype(np.float64)
print("M-step product affinities^T . data: device %.2f ms (upload of the affinities + kernels + download), NumPy %.1f ms; "
      "max deviation from float64 relative to the value: device %.1e, NumPy %.1e"
      % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, np.max(np.abs(ws - ws64) / np.maximum(ws64, 1)), np.max(np.abs(ws_np - ws64) / np.maximum(ws64, 1))))
