"""Diagonal-covariance Gaussian mixture for the cross-entropy pose search (reference pose/estimation.py:412-473 fits
`sklearn.mixture.GaussianMixture(covariance_type='diag', n_components=k, reg_covar=1e-5)` to <= 48 elite poses x 6 parameters
on the host every iteration and samples the next hypotheses from it).

scikit-learn's `fit` spends ~3 ms there (20 ms on the build container) -- parameter validation, a KMeans estimator object with
its own validation and thread-pool control, warnings machinery -- on a problem of 48 x 6 numbers, with the GPU idle: 10 % of the
released architecture's iteration.  `DiagGMM` is the same algorithm without the framework around it:

  * initialisation: hard k-means assignments (k-means++ seeding + Lloyd iterations; scikit-learn's default `init_params='kmeans'`)
    turned into one-hot responsibilities -- or responsibilities handed in (`fit(X, resp=...)`), which is how the tests pin the EM
    against scikit-learn: the same start gives the same mixture (weights / means / covariances to 1e-6);
  * EM: scikit-learn's equations for 'diag' (sklearn/mixture/_gaussian_mixture.py: _estimate_gaussian_parameters,
    _estimate_log_gaussian_prob, _compute_precision_cholesky), reg_covar added to the variances, convergence when the change of
    the mean log-likelihood falls below tol = 1e-3, at most 100 iterations;
  * `sample(n)`: scikit-learn's draw order on numpy's GLOBAL generator (multinomial over the weights, then standard normals per
    component), so a seeded search draws the same hypotheses from the same mixture as with scikit-learn.

Attributes mirror scikit-learn's (`weights_`, `means_`, `covariances_`, `precisions_cholesky_`, `converged_`, `n_iter_`), which
is all CrossEntropyPoseEstimator._combined_gmm touches."""
import numpy as np

_LOG_2PI = float(np.log(2.0 * np.pi))


def _kmeans_labels(X, k, rng, max_iter=300, tol=1e-4):
    """Hard assignments of a small data set: k-means++ seeding, Lloyd iterations until the centres move less than
    tol * mean variance (scikit-learn's stopping rule)."""
    n = X.shape[0]
    k = min(k, n)
    centres = np.empty((k, X.shape[1]))
    centres[0] = X[rng.randint(n)]
    d2 = ((X - centres[0]) ** 2).sum(1)
    for j in range(1, k):
        tot = d2.sum()
        idx = rng.randint(n) if not tot > 0.0 else int(np.searchsorted(np.cumsum(d2), rng.random_sample() * tot))
        centres[j] = X[min(idx, n - 1)]
        d2 = np.minimum(d2, ((X - centres[j]) ** 2).sum(1))
    thresh = tol * X.var(axis=0).mean()
    labels = np.zeros(n, dtype=np.int64)
    for _ in range(max_iter):
        dist = ((X[:, None, :] - centres[None, :, :]) ** 2).sum(2)
        labels = dist.argmin(1)
        new = centres.copy()
        for j in range(k):
            m = labels == j
            if m.any():
                new[j] = X[m].mean(0)
        shift = ((new - centres) ** 2).sum()
        centres = new
        if shift <= thresh:
            break
    return labels


class DiagGMM:
    def __init__(self, n_components, reg_covar=1e-5, tol=1e-3, max_iter=100):
        self.n_components, self.reg_covar, self.tol, self.max_iter = int(n_components), float(reg_covar), float(tol), int(max_iter)
        self.converged_, self.n_iter_ = False, 0

    # ---- scikit-learn's M and E steps for covariance_type='diag' ----
    def _m_step(self, X, resp):
        nk = resp.sum(axis=0) + 10.0 * np.finfo(resp.dtype).eps
        means = resp.T @ X / nk[:, None]
        avg_x2 = resp.T @ (X * X) / nk[:, None]
        self.covariances_ = avg_x2 - means ** 2 + self.reg_covar
        self.means_ = means
        w = nk / X.shape[0]
        self.weights_ = w / w.sum()
        self.precisions_cholesky_ = 1.0 / np.sqrt(self.covariances_)

    def _e_step(self, X):
        pc = self.precisions_cholesky_
        prec = pc ** 2
        log_det = np.log(pc).sum(axis=1)
        log_prob = (self.means_ ** 2 * prec).sum(axis=1) - 2.0 * X @ (self.means_ * prec).T + (X * X) @ prec.T
        weighted = -0.5 * (X.shape[1] * _LOG_2PI + log_prob) + log_det + np.log(self.weights_)
        mx = weighted.max(axis=1, keepdims=True)
        norm = mx[:, 0] + np.log(np.exp(weighted - mx).sum(axis=1))
        return float(norm.mean()), weighted - norm[:, None]

    def fit(self, X, resp=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        n = X.shape[0]
        if n < self.n_components:
            raise ValueError(f'Expected n_samples >= n_components but got n_components = {self.n_components}, n_samples = {n}')
        if resp is None:
            labels = _kmeans_labels(X, self.n_components, np.random.mtrand._rand)
            resp = np.zeros((n, self.n_components))
            resp[np.arange(n), labels] = 1.0
        self._m_step(X, np.asarray(resp, dtype=np.float64))
        lower = -np.inf
        self.converged_ = False
        for it in range(1, self.max_iter + 1):
            prev = lower
            lower, log_resp = self._e_step(X)
            self._m_step(X, np.exp(log_resp))
            self.n_iter_ = it
            if abs(lower - prev) < self.tol:
                self.converged_ = True
                break
        self.lower_bound_ = lower
        return self

    def sample(self, n_samples=1):
        """scikit-learn's GaussianMixture.sample for 'diag', drawing from numpy's global generator in the same order."""
        rng = np.random.mtrand._rand
        counts = rng.multinomial(n_samples, self.weights_)
        X = np.vstack([mean + rng.standard_normal(size=(int(c), mean.shape[0])) * np.sqrt(cov)
                       for mean, cov, c in zip(self.means_, self.covariances_, counts)])
        y = np.concatenate([np.full(int(c), j, dtype=int) for j, c in enumerate(counts)])
        return X, y
