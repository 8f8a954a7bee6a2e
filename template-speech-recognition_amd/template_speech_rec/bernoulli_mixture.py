"""Bernoulli mixture whose EM runs on the GPU.

Same public surface as the reference's bernoulli_mixture.BernoulliMixture (constructor arguments, attribute
names, run_EM / M_step / _compute_loglikelihoods / get_template_loglikelihoods / set_templates / save,
bernoulli_mixture.py:73-312), different inside: the 0/1 data matrix is uploaded once and stays resident; an
E-step is ONE library call (tsr_mixture_e_step) that scores every row against every component --
log P(x_n | k) = sum_d log(1-T_k[d]) + sum_d x[n,d] (log T_k - log(1-T_k))[d], one wave per row, the K log-odds
rows shared through the caches -- and finishes on the device
with the max-shifted log-sum-exp, the log-likelihood and the posterior component probabilities
(:263-285).  The posteriors never leave the device between the E-step and the M-step (tsr_mixture_m_step:
the column means and affinities^T . data_mat of :175-179); `affinities` is fetched only when somebody reads it.
The reference evaluates these sums with float32 BLAS in no defined order, so agreement is to float32
rounding (golden g21 holds the outputs of the reference's own class).
"""
import collections
import ctypes as C

import numpy as np

from tsr_hip import api
from tsr_hip._lib import check, lib, ptr

BernoulliMixtureSimple = collections.namedtuple('BernoulliMixtureSimple', 'log_templates log_invtemplates weights')


def _log_tables(prob):
    """(log p, log(1 - p)) of a probability table."""
    prob = np.asarray(prob)
    return np.log(prob), np.log(1 - prob)


class BernoulliMixture:
    def __init__(self, num_mix, data_mat, init_type='unif_rand', init_seed=0, init_centers=None,
                 min_probability=.01, ctx=None):
        data_mat = np.asarray(data_mat)
        self.num_mix = int(num_mix)
        self.num_data = int(data_mat.shape[0])
        self.data_shape = tuple(data_mat.shape[1:])
        self.data_length = int(np.prod(self.data_shape, dtype=np.int64))
        self.data_mat = data_mat.reshape(self.num_data, self.data_length)
        if ((self.data_mat != 0) & (self.data_mat != 1)).any():
            raise ValueError("BernoulliMixture needs 0/1 data")
        self.not_data_mat = 1 - self.data_mat
        self.iterations = 0
        self.seed = init_seed
        np.random.seed(self.seed)            # (the reference seeds NumPy here; its initialisations draw nothing from it)
        self.min_probability = np.float32(min_probability)
        self.weights = np.full(self.num_mix, 1. / self.num_mix, dtype=np.float32)
        self.hard_assignment = False
        # device side: the data matrix as a corpus of one-frame utterances, uploaded once
        self._ctx = ctx or api.default_ctx()
        self._corpus = api.Corpus(self.data_mat.astype(np.uint8), np.arange(self.num_data + 1, dtype=np.int64),
                                  D=self.data_length, ctx=self._ctx)
        self._aff = None                     # host copy of the posteriors (None: fetch from the device on demand)
        self._aff_on_device = False          # the device holds the posteriors of the latest E-step
        self.init_affinities_templates(init_type, init_centers=init_centers)

    # ---- posteriors: resident on the device, mirrored on the host lazily ---------------------------------------
    @property
    def affinities(self):
        if self._aff is None:
            if not self._aff_on_device:
                raise AttributeError("affinities: no E-step has run and none were set")
            out = np.empty((self.num_data, self.num_mix), np.float32)
            check(lib().tsr_corpus_download_affinities(self._ctx._h, self._corpus._h, ptr(out)))
            self._aff = out
        return self._aff

    @affinities.setter
    def affinities(self, value):
        self._aff = np.asarray(value)
        self._aff_on_device = False          # the next M-step uploads these instead

    # ---- initialisation (bernoulli_mixture.py:195-237) -----------------------------------------------------------
    def init_affinities_templates(self, init_type, init_centers=None):
        K, N = self.num_mix, self.num_data
        if init_type == 'preset' and init_centers is not None:
            self.num_mix = K = int(init_centers.shape[0])
            self.work_templates = np.asarray(init_centers).reshape(K, self.data_length)
        elif init_type in ('unif_rand', 'specific'):
            # round-robin hard assignment of the first K * (N // K) rows ('specific': of a single row per
            # component, the reference's `np.arange(N / K)[1]`); a component starts as the mean of its rows
            rows = np.arange(N // K) if init_type == 'unif_rand' else np.arange(N // K)[1:2]
            start = np.zeros((N, K))
            for k in range(K):
                start[K * rows + k, k] = 1.
            members = start.sum(0)
            self.work_templates = (start.T @ self.data_mat.astype(np.float64)) / members[:, None]
            self.affinities = start
        self.threshold_templates()
        self.log_templates, self.log_invtemplates = _log_tables(self.work_templates)

    def threshold_templates(self):
        lo = self.min_probability
        self.work_templates = np.clip(self.work_templates, lo, 1 - lo)

    def init_templates(self):
        """:239-243."""
        self.work_templates = np.zeros((self.num_mix, self.data_length))
        self.templates = np.zeros((self.num_mix, self.data_length))

    def set_templates(self):
        """:245-248: the tables in the shape of a data entry."""
        self.templates = self.work_templates.reshape((self.num_mix,) + self.data_shape)
        self.log_templates, self.log_invtemplates = _log_tables(self.templates)

    def reset_templates(self):
        """:250-256: back to flat [num_mix, data_length] tables."""
        self.templates = self.work_templates.reshape(self.num_mix, self.data_length)
        self.log_templates, self.log_invtemplates = _log_tables(self.templates)

    def set_weights(self, new_weights):
        new_weights = np.asarray(new_weights)
        if new_weights.shape != (self.num_mix,):
            raise AssertionError("weights must have one entry per component")
        np.testing.assert_approx_equal(np.sum(new_weights), 1.)
        self.weights = new_weights

    def get_bernoulli_mixture_named_tuple(self):
        return BernoulliMixtureSimple(log_templates=self.log_templates, log_invtemplates=self.log_invtemplates,
                                      weights=self.weights)

    # ---- EM (:108-184, :263-291) -----------------------------------------------------------------------------------
    def _component_tables(self):
        """Per component k the log-odds row log T_k - log(1 - T_k) (f32 [K, D]) and c_k = sum_d log(1 - T_k[d]) (f64)."""
        lt = np.asarray(self.log_templates, np.float64).reshape(self.num_mix, self.data_length)
        lit = np.asarray(self.log_invtemplates, np.float64).reshape(self.num_mix, self.data_length)
        return np.ascontiguousarray((lt - lit).astype(np.float32)), np.ascontiguousarray(lit.sum(axis=1))

    def get_template_loglikelihoods(self):
        """[num_data, num_mix]: log P(x_n | component k) (:287-291), one pass over the resident data matrix."""
        odds, consts = self._component_tables()
        out = np.empty((self.num_mix, self.num_data), np.float32)
        check(lib().tsr_mixture_scores(self._ctx._h, self._corpus._h, ptr(odds), ptr(consts), self.num_mix, ptr(out)))
        return out.T.astype(np.float64)

    def _compute_loglikelihoods(self):
        """E-step (:263-285): returns the data log-likelihood under the current tables and weights and leaves the
        posteriors (one-hot on the best-scoring component with hard assignment) on the device."""
        odds, consts = self._component_tables()
        with np.errstate(divide='ignore'):
            log_w = np.log(np.asarray(self.weights, np.float64)).reshape(-1)
        if log_w.shape[0] != self.num_mix:
            # (a 'preset' initialisation with another number of centers changes num_mix but not the weights: the
            # reference's np.tile(np.log(self.weights), ...) + scores then fails to broadcast, :265)
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) (%d,%d) "
                             % (self.num_data, self.num_mix, self.num_data, log_w.shape[0]))
        total = C.c_double()
        check(lib().tsr_mixture_e_step(self._ctx._h, self._corpus._h, ptr(odds), ptr(consts), self.num_mix,
                                       ptr(np.ascontiguousarray(log_w)), int(bool(self.hard_assignment)), C.byref(total), None))
        self._aff, self._aff_on_device = None, True
        return float(total.value)

    def M_step(self):
        """:174-182: weights = mean posterior, tables = posterior-weighted means of the data, clipped."""
        K = self.num_mix
        if self._aff_on_device:
            w = np.empty(K, np.float64)
            sums = np.empty((K, self.data_length), np.float32)
            check(lib().tsr_mixture_m_step(self._ctx._h, self._corpus._h, ptr(w), ptr(sums)))
            self.weights = w.astype(np.float32)
        else:                                # posteriors assigned from outside: upload them for the product
            post = np.asarray(self.affinities)
            self.weights = post.mean(axis=0)
            sums = self._corpus.weighted_sums(post)
        self.work_templates = sums / np.float32(self.num_data) / self.weights.reshape(K, 1).astype(np.float32)
        self.threshold_templates()
        lt, lit = _log_tables(self.work_templates)
        self.log_templates, self.log_invtemplates = lt.astype(np.float32), lit.astype(np.float32)

    def run_EM(self, tol, min_probability=0.05, debug_plot=False, hard_assignment=False, rand_seed=None,
               use_templates=None, max_iter=500, min_iter=10):
        """Iterate M-step / E-step until the relative change of the log-likelihood is at most `tol` (and at
        least `min_iter`, at most `max_iter` times); returns the log-likelihood before the last update."""
        self.hard_assignment = hard_assignment
        self.min_probability = np.float32(min_probability)
        current = self._compute_loglikelihoods()
        previous = current / (1 - tol) - 10          # the reference's way of forcing the first iteration
        self.iterations = 0

        def moving():
            return abs((previous - current) / previous) > tol

        while self.iterations < min_iter or (self.iterations < max_iter and moving()):
            previous = current
            self.M_step()
            current = self._compute_loglikelihoods()
            self.iterations += 1
        self.set_templates()
        return previous

    def save(self, filename, save_affinities=False):
        entries = {'templates': self.templates, 'weights': self.weights}
        if save_affinities:
            entries['affinities'] = self.affinities
        np.savez(filename, **entries)
