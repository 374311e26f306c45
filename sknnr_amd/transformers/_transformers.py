"""scikit-learn transformers that define the feature spaces of the estimators.

Same names, constructor parameters, fitted attributes and error behaviour as
/root/reference/src/sknnr/transformers/ (``_base.py``, ``_cca_transformer.py``,
``_ccora_transformer.py``, ``_mahalanobis_transformer.py``).  At query time every one of
them is the affine map ``((X - center) / scale) @ proj``; ``affine_params()`` hands that
map to the GPU engine, which applies it inside the kneighbors/predict launch
(``sknnr_index_set_affine``).  ``transform()`` itself is the plain numpy expression and
is not on the accelerated path.
"""

from __future__ import annotations

import numpy as np
from sklearn.base import BaseEstimator, OneToOneFeatureMixin, TransformerMixin
from sklearn.preprocessing import StandardScaler
from sklearn.utils.validation import FLOAT_DTYPES, check_is_fitted, validate_data

from ._ordination import CCA, CCorA

__all__ = [
    "StandardScalerWithDOF",
    "MahalanobisTransformer",
    "CCATransformer",
    "CCorATransformer",
    "ComponentReducerMixin",
]


class StandardScalerWithDOF(StandardScaler):
    """``StandardScaler`` whose ``scale_`` is the standard deviation with ``ddof`` degrees
    of freedom (REF transformers/_base.py:23-67)."""

    def __init__(self, ddof: int = 0):
        super().__init__()
        self.ddof = ddof

    def fit(self, X, y=None):
        fitted = super().fit(X, y)
        arr = validate_data(
            self, X=X, accept_sparse=False, dtype=FLOAT_DTYPES, ensure_all_finite="allow-nan",
            reset=False, ensure_min_samples=self.ddof + 1,
        )
        fitted.scale_ = np.std(arr, axis=0, ddof=self.ddof)
        return fitted

    def affine_params(self):
        check_is_fitted(self)
        return self.mean_, self.scale_, None


class ComponentReducerMixin:
    """Shared ``n_components`` handling of the ordination transformers
    (REF transformers/_base.py:70-101)."""

    def __init__(self, n_components: int | None = None):
        self.n_components = n_components

    def get_feature_names_out(self, input_features=None):
        check_is_fitted(self, "n_components_")
        stem = type(self.ordination_).__name__.lower()
        return np.asarray([f"{stem}{i}" for i in range(self.n_components_)], dtype=object)

    def set_n_components(self) -> None:
        limit = self.ordination_.max_components
        wanted = limit if self.n_components is None else self.n_components
        if not 0 <= wanted <= limit:
            raise ValueError(f"n_components={wanted} must be between 0 and {limit}")
        self.n_components_ = wanted


class CCATransformer(ComponentReducerMixin, TransformerMixin, BaseEstimator):
    """Canonical correspondence analysis feature space (GNN); REF
    transformers/_cca_transformer.py:22-97."""

    def fit(self, X, y):
        arr = validate_data(
            self, X=X, reset=True, dtype=FLOAT_DTYPES, ensure_all_finite=True,
            ensure_min_features=2, ensure_min_samples=1,
        )
        y = np.asarray(y)
        if y.ndim < 2:
            raise ValueError("`y` must be a 2D array.")
        self.ordination_ = CCA(arr, y)
        self.set_n_components()
        self.env_center_ = self.ordination_.env_center
        self.projector_ = self.ordination_.projector(self.n_components_)
        return self

    def transform(self, X, y=None):
        check_is_fitted(self)
        arr = validate_data(
            self, X=X, reset=False, dtype=FLOAT_DTYPES, ensure_all_finite=True,
            ensure_min_features=2, ensure_min_samples=1,
        )
        return (arr - self.env_center_) @ self.projector_

    def fit_transform(self, X, y):
        return self.fit(X, y).transform(X)

    def affine_params(self):
        check_is_fitted(self)
        return self.env_center_, None, self.projector_

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.target_tags.required = True
        tags.target_tags.positive_only = True
        return tags


class CCorATransformer(ComponentReducerMixin, TransformerMixin, BaseEstimator):
    """Canonical correlation analysis feature space (MSN); REF
    transformers/_ccora_transformer.py:21-79."""

    def fit(self, X, y):
        _, y_arr = validate_data(self, X=X, y=y, reset=True, multi_output=True)
        self.scaler_ = StandardScalerWithDOF(ddof=1).fit(X)
        if y_arr.ndim == 1:
            y_arr = y_arr.reshape(-1, 1)
        y_std = StandardScalerWithDOF(ddof=1).fit_transform(y_arr)
        self.ordination_ = CCorA(self.scaler_.transform(X), y_std)
        self.set_n_components()
        self.projector_ = self.ordination_.projector(self.n_components_)
        return self

    def transform(self, X, y=None):
        check_is_fitted(self)
        validate_data(self, X=X, reset=False, ensure_all_finite=True)
        return self.scaler_.transform(X) @ self.projector_

    def fit_transform(self, X, y):
        return self.fit(X, y).transform(X)

    def affine_params(self):
        check_is_fitted(self)
        return self.scaler_.mean_, self.scaler_.scale_, self.projector_

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.target_tags.required = True
        return tags


class MahalanobisTransformer(OneToOneFeatureMixin, TransformerMixin, BaseEstimator):
    """Standardise, then whiten with the inverse Cholesky factor of the covariance, so that
    Euclidean distance equals Mahalanobis distance; REF
    transformers/_mahalanobis_transformer.py:20-64."""

    def fit(self, X, y=None):
        validate_data(self, X=X, ensure_all_finite="allow-nan", reset=True, ensure_min_features=2)
        self.scaler_ = StandardScalerWithDOF(ddof=1).fit(X)
        cov = np.cov(self.scaler_.transform(X), rowvar=False)
        self.transform_ = np.linalg.inv(np.linalg.cholesky(cov).T)
        return self

    def transform(self, X, y=None):
        check_is_fitted(self)
        validate_data(self, X=X, ensure_all_finite="allow-nan", reset=False)
        return self.scaler_.transform(X) @ self.transform_

    def fit_transform(self, X, y=None):
        return self.fit(X, y).transform(X)

    def affine_params(self):
        check_is_fitted(self)
        return self.scaler_.mean_, self.scaler_.scale_, self.transform_

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.input_tags.allow_nan = True
        return tags
