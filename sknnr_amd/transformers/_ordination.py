"""Fit-time ordination algebra for the GNN (CCA) and MSN (CCorA) feature spaces.

Out of the accelerated hot path: a one-off O(N D^2)
computation at ``fit`` whose only product that the GPU ever sees is the projector
matrix.  Both solvers run once in ``__init__`` and keep their factors (the reference
recomputes QR/SVD on every property access).

Algorithms restated (pinned by ``tests/golden/moscow_*.npz`` ``tr_projector_``):

* constrained correspondence analysis as in vegan's ``ordConstrained`` --
  /root/reference/src/sknnr/transformers/_cca.py:22-45 (chi-square standardisation),
  :110-140 (weighted QR, least squares, SVD, rank), :193-203 (projector);
* canonical correlation analysis as in statsmodels' ``CanCorr`` with yaImpute's
  coefficient scaling and F test -- /root/reference/src/sknnr/transformers/_ccora.py:5-41,
  :56-67, :96-136.
"""

from __future__ import annotations

import math

import numpy as np
from scipy.stats import f as f_dist

__all__ = ["CCA", "CCorA"]


class CCA:
    """Canonical correspondence analysis of species matrix ``Y`` constrained by ``X``."""

    #: singular values below this are treated as zero (vegan's ``sqrt(.Machine$double.eps)``)
    ZERO = math.sqrt(2.220446e-16)

    def __init__(self, X, Y):
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        if np.any(Y.sum(axis=1) <= 0.0):
            raise ValueError("All row sums must be greater than 0")
        self.exclude_Y = Y.sum(axis=0) <= 0.0
        Y = Y[:, ~self.exclude_Y]

        # chi-square standardised species table (vegan initCA)
        P = Y / Y.sum()
        self.rw = P.sum(axis=1)
        self.cw = P.sum(axis=0)
        expected = np.outer(self.rw, self.cw)
        self.Ybar = (P - expected) / np.sqrt(expected)

        # row-weighted, centred environment
        self.env_center = np.average(X, axis=0, weights=self.rw)
        self.X_scale = (X - self.env_center) * np.sqrt(self.rw)[:, None]

        # weighted regression of Ybar on X through QR, then SVD of the fitted values
        self.Q, self.R = np.linalg.qr(self.X_scale)
        beta, _, ls_rank, _ = np.linalg.lstsq(self.R, self.Q.T @ self.Ybar, rcond=None)
        self.Y_fit = self.X_scale @ beta
        u, s, vt = np.linalg.svd(self.Y_fit, full_matrices=False)
        self.rank = int(min(ls_rank, np.sum(s > self.ZERO)))
        self._u = u[:, : self.rank]
        self._vt = vt[: self.rank]
        self.eigenvalues = np.square(s)[: self.rank]

        self.coefficients = np.linalg.lstsq(self.R, self.Q.T @ self._u, rcond=None)[0]
        self.axis_weights = np.diag(np.sqrt(self.eigenvalues / self.eigenvalues.sum()))

    @property
    def max_components(self) -> int:
        return self.rank

    def projector(self, n_components: int) -> np.ndarray:
        n = n_components
        return self.coefficients[:, :n] @ self.axis_weights[:n, :n]

    # site / species scores, for users of the reference's ordination object
    @property
    def site_lc_scores(self):
        return self._u / np.sqrt(self.rw)[:, None]

    @property
    def species_scores(self):
        return (self._vt.T / np.sqrt(self.cw)[:, None]) * np.sqrt(self.eigenvalues)


def _pvalues_of_canonical_correlations(p: int, q: int, n: int, cor: np.ndarray) -> np.ndarray:
    """Rao's F approximation for 'correlations i..s are all zero' (yaImpute ``ftest.cor``)."""
    s = min(p, q)
    k = np.arange(1, s + 1)
    wilks = np.array([np.prod(1.0 - np.square(cor[i:s])) for i in range(s)])
    r = (n - s - 1) - ((abs(p - q) + 1) / 2)
    a, b = p - k + 1, q - k + 1
    ndf = a * b
    u = (ndf - 2) / 4
    denom = np.square(a) + np.square(b) - 5
    t = np.zeros(s)
    pos = denom > 0
    t[pos] = np.sqrt((np.square(a[pos]) * np.square(b[pos]) - 4) / denom[pos])
    keep = t > 0
    t, wilks, u, ndf = t[keep], wilks[keep], u[keep], ndf[keep]
    root = np.power(wilks, 1.0 / t)
    ddf = r * t - 2 * u
    bad = (ddf < 1.0) | (ndf < 1)
    first_bad = int(np.argmax(bad)) if bad.any() else len(bad)
    ok = np.arange(len(bad)) < first_bad
    fstat = ((1.0 - root) / root) * (ddf / ndf)
    return np.array([1.0 - f_dist.cdf(fstat[i], ndf[i], ddf[i]) for i in np.nonzero(ok)[0]])


class CCorA:
    """Canonical correlation analysis between feature block ``X`` and target block ``y``."""

    TOLERANCE = 1e-8
    P_VAL = 0.05

    def __init__(self, X, y):
        X = np.array(X, dtype=np.float64)
        y = np.array(y, dtype=np.float64)
        self.k = min(X.shape[1], y.shape[1])
        self.X_norm = X - X.mean(axis=0)
        self.y_norm = y - y.mean(axis=0)

        ux, vx_ds = self._whiten(self.X_norm)
        uy, vy_ds = self._whiten(self.y_norm)
        left, s, right_t = np.linalg.svd(ux.T @ uy, full_matrices=False)
        self.cancorr = np.clip(s, 0.0, 1.0)

        # yaImpute scales every coefficient so that the first canonical variate has unit sd
        first = self.X_norm @ (vx_ds @ left[:, 0])
        self.cscal = 1.0 / np.std(first, ddof=1)
        self.x_coef = vx_ds @ left[:, : self.k] * self.cscal
        self.y_coef = vy_ds @ right_t.T[:, : self.k] * self.cscal

        self.f_test = _pvalues_of_canonical_correlations(
            self.y_coef.shape[0], self.x_coef.shape[0], self.y_norm.shape[0], self.cancorr
        )
        self.n_vec = max(1, len(self.f_test) - int(np.sum(self.f_test > self.P_VAL)))

    def _whiten(self, arr):
        """Thin SVD with near-null directions removed: returns (U, V' / s)."""
        u, s, vt = np.linalg.svd(arr, full_matrices=False)
        keep = s > self.TOLERANCE
        # the reference masks both axes of V' (rows, then columns) -- kept as is
        vt = vt[keep][:, keep]
        return u[:, keep], vt.T / s[keep]

    @property
    def max_components(self) -> int:
        return self.n_vec

    def projector(self, n_components: int) -> np.ndarray:
        n = n_components
        return self.x_coef[:, :n] @ np.diag(self.cancorr[:n])
