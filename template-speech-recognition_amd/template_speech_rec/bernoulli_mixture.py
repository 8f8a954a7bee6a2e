"""Bernoulli mixture with the E-step's scoring on the GPU (the GEMM half of it).

Mirrors the reference's bernoulli_mixture.BernoulliMixture (constructor arguments, attributes, run_EM /
M_step / _compute_loglikelihoods / get_template_loglikelihoods, :9-312).  The one heavy expression of
the EM loop,

    get_template_loglikelihoods = data_mat . log_templates^T + (1 - data_mat) . log_invtemplates^T   (:287-291)

is, for binary data, exactly the bank score of one-frame "utterances": sum_d log_invtemplates[k, d] +
sum_d x[n, d] * (log_templates - log_invtemplates)[k, d].  It runs on the matrix cores with exact integer
accumulation of 23-bit fixed-point weights (TSR_FAST | TSR_W24); the data matrix is uploaded once and
stays resident across the EM iterations.  The reference evaluates the same expression with a float32
BLAS product, so the two agree to float32 rounding, not bit for bit (no summation order is defined
there).  The M-step product affinities^T . data_mat (:179) runs on the same resident matrix
(tsr_corpus_weighted_sums); the bookkeeping stays the reference's NumPy expressions.
"""
import collections

import numpy as np

from tsr_hip import api

BernoulliMixtureSimple = collections.namedtuple('BernoulliMixtureSimple', 'log_templates log_invtemplates weights')


class BernoulliMixture:
    def __init__(self, num_mix, data_mat, init_type='unif_rand', init_seed=0, init_centers=None,
                 min_probability=.01, ctx=None):
        self.num_mix = num_mix
        self.num_data = data_mat.shape[0]
        self.data_shape = data_mat.shape[1:]
        self.data_length = int(np.prod(data_mat.shape[1:]))
        self.data_mat = data_mat.reshape(self.num_data, self.data_length)
        if not np.all((self.data_mat == 0) | (self.data_mat == 1)):
            raise ValueError("BernoulliMixture needs 0/1 data")
        self.iterations = 0
        self.seed = init_seed
        np.random.seed(self.seed)
        self.min_probability = np.float32(min_probability)
        self.not_data_mat = 1 - self.data_mat
        self.weights = 1. / num_mix * np.ones(num_mix, dtype=np.float32)
        self.hard_assignment = False
        self._ctx = ctx or api.default_ctx()
        self._corpus = api.Corpus(self.data_mat.astype(np.uint8), np.arange(self.num_data + 1, dtype=np.int64),
                                  D=self.data_length, ctx=self._ctx)
        self.init_affinities_templates(init_type, init_centers=init_centers)

    # ---- initialisation (:195-237) -------------------------------------------------------------------
    def init_affinities_templates(self, init_type, init_centers=None):
        if init_type in ('unif_rand', 'specific'):
            self.affinities = np.zeros((self.num_data, self.num_mix))
            self.work_templates = np.zeros((self.num_mix, self.data_length))
            base = self.num_mix * np.arange(self.num_data // self.num_mix)
            if init_type == 'specific':
                base = base[1]
            for mix_id in range(self.num_mix):
                self.affinities[base + mix_id, mix_id] = 1.
                self.work_templates[mix_id] = np.mean(self.data_mat[self.affinities[:, mix_id] == 1], axis=0)
                self.threshold_templates()
        elif init_type == 'preset' and init_centers is not None:
            self.num_mix = init_centers.shape[0]
            self.work_templates = np.clip(init_centers.reshape(self.num_mix, self.data_length),
                                          self.min_probability, 1 - self.min_probability)
        self.work_templates = np.clip(self.work_templates, self.min_probability, 1 - self.min_probability)
        self.log_templates = np.log(self.work_templates)
        self.log_invtemplates = np.log(1 - self.work_templates)

    def threshold_templates(self):
        self.work_templates = np.clip(self.work_templates, self.min_probability, 1 - self.min_probability)

    def set_templates(self):
        self.templates = self.work_templates.reshape((self.num_mix,) + tuple(self.data_shape))
        self.log_templates = np.log(self.templates)
        self.log_invtemplates = np.log(1 - self.templates)

    def reset_templates(self):
        """bernoulli_mixture.py:250-256: back to the flat [num_mix, data_length] tables (set_templates leaves them
        in data shape, which the E-step does not take)."""
        self.templates = self.work_templates.reshape(self.num_mix, self.data_length)
        self.log_templates = np.log(self.templates)
        self.log_invtemplates = np.log(1 - self.templates)

    def init_templates(self):
        """bernoulli_mixture.py:239-243."""
        self.work_templates = np.zeros((self.num_mix, self.data_length))
        self.templates = np.zeros((self.num_mix, self.data_length))

    def set_weights(self, new_weights):
        np.testing.assert_approx_equal(np.sum(new_weights), 1.)
        assert new_weights.shape == (self.num_mix,)
        self.weights = new_weights

    def get_bernoulli_mixture_named_tuple(self):
        return BernoulliMixtureSimple(log_templates=self.log_templates, log_invtemplates=self.log_invtemplates,
                                      weights=self.weights)

    # ---- EM (:108-184, :263-291) -----------------------------------------------------------------------
    def run_EM(self, tol, min_probability=0.05, debug_plot=False, hard_assignment=False, rand_seed=None,
               use_templates=None, max_iter=500, min_iter=10):
        self.hard_assignment = hard_assignment
        self.min_probability = np.float32(min_probability)
        new_loglikelihood = self._compute_loglikelihoods()
        loglikelihood = new_loglikelihood / (1 - tol) - 10
        self.iterations = 0
        while (np.abs((loglikelihood - new_loglikelihood) / loglikelihood) > tol and self.iterations < max_iter) \
                or self.iterations < min_iter:
            loglikelihood = new_loglikelihood
            self.M_step()
            new_loglikelihood = self._compute_loglikelihoods()
            self.iterations += 1
        self.set_templates()
        return loglikelihood

    def M_step(self):
        self.weights = np.mean(self.affinities, axis=0)
        self.work_templates = self._corpus.weighted_sums(self.affinities)   # np.dot(affinities.T, data_mat), :179
        self.work_templates /= self.num_data
        self.work_templates /= self.weights.reshape((self.num_mix, 1))
        self.threshold_templates()
        self.log_templates = np.log(self.work_templates).astype(np.float32)
        self.log_invtemplates = np.log(1 - self.work_templates).astype(np.float32)

    def _compute_loglikelihoods(self):
        template_logscores = self.get_template_loglikelihoods()
        loglikelihoods = template_logscores + np.tile(np.log(self.weights), (self.num_data, 1))
        max_vals = np.amax(loglikelihoods, axis=1)
        logmarginals_adj = np.sum(np.exp(loglikelihoods - np.tile(max_vals, (self.num_mix, 1)).transpose()), axis=1)
        loglikelihood = np.sum(np.log(logmarginals_adj)) + np.sum(max_vals)
        if self.hard_assignment:
            self.affinities[:] = 0
            self.affinities[np.arange(self.num_data), np.argmax(template_logscores, 1)] = 1
            return loglikelihood
        self.affinities = np.exp(loglikelihoods - np.tile(logmarginals_adj + max_vals,
                                                          (self.num_mix, 1)).transpose()).astype(np.float32)
        self.affinities /= np.tile(np.sum(self.affinities, axis=1), (self.num_mix, 1)).transpose().astype(np.float32)
        return loglikelihood

    def get_template_loglikelihoods(self):
        """[num_data, num_mix]: log P(x_n | component k), one launch over the resident data matrix."""
        lt = np.asarray(self.log_templates, np.float64).reshape(self.num_mix, self.data_length)
        lit = np.asarray(self.log_invtemplates, np.float64).reshape(self.num_mix, self.data_length)
        bank = api.Bank([((lt[k] - lit[k]).astype(np.float32)[None, :], float(lit[k].sum()))
                         for k in range(self.num_mix)], ctx=self._ctx)
        return self._corpus.score_all(bank, api.TSR_FAST | api.TSR_W24).T.astype(np.float64)

    def save(self, filename, save_affinities=False):
        entries = dict(templates=self.templates, weights=self.weights)
        if save_affinities:
            entries['affinities'] = self.affinities
        np.savez(filename, **entries)
