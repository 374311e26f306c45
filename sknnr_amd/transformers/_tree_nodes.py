"""Tree-node feature spaces of the RFNN / GBNN estimators: one forest per target, every sample
described by the node it reaches in every tree.

Public surface (class names, constructor parameters, fitted attributes, feature names) follows
/root/reference/src/sknnr/transformers/_tree_node_transformer.py, _rfnode_transformer.py and
_gbnode_transformer.py.  Growing the forests and ``apply`` are scikit-learn's (host, outside the hot
path); what the GPU sees is the int64 node-id matrix these classes emit and
the per-tree weights, which feed the weighted-Hamming search (``SKNNR_FORMULA_HAMMING``).
"""

from __future__ import annotations

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.ensemble import (
    GradientBoostingClassifier,
    GradientBoostingRegressor,
    RandomForestClassifier,
    RandomForestRegressor,
)
from sklearn.utils.validation import check_array, check_is_fitted, validate_data

__all__ = ["TreeNodeTransformer", "RFNodeTransformer", "GBNodeTransformer"]


# ---------------------------------------------------------------------------------------------
# target columns: names and the estimator family each one needs (REF utils/__init__.py)
# ---------------------------------------------------------------------------------------------
def _is_missing(v) -> bool:
    return v is None or (isinstance(v, float) and v != v) or type(v).__name__ == "NAType"


def _columns_of(y):
    """``[(name, column values as an object array, declared dtype or None)]`` for a dataframe, series or
    array-like ``y``."""
    if hasattr(y, "columns") and hasattr(y, "dtypes"):  # dataframe-like (pandas, polars)
        declared = list(getattr(y.dtypes, "values", y.dtypes))
        data = np.asarray(y, dtype=object)
        return [(name, data[:, i], declared[i]) for i, name in enumerate(y.columns)]
    if hasattr(y, "name") and hasattr(y, "dtype") and not isinstance(y, np.ndarray):  # series-like
        return [("0" if y.name is None else y.name, np.asarray(y, dtype=object).reshape(-1), y.dtype)]
    data = np.asarray(y, dtype=object)
    if data.ndim == 1:
        data = data.reshape(-1, 1)
    return [(str(i), data[:, i], None) for i in range(data.shape[1])]


def _target_plan(y):
    """Per target: name, values promoted to the narrowest numpy dtype that holds them, and
    'regression' (numeric) or 'classification' (anything else, pandas categoricals included)."""
    plan = []
    columns = _columns_of(y)
    # target names key estimator_type_dict_: a name met twice is refused (1 and "1" are different keys)
    # REF utils/__init__.py:167-172
    seen, twice = set(), []
    for name, _, _ in columns:
        if name in seen and name not in twice:
            twice.append(name)
        seen.add(name)
    if twice:
        raise ValueError(f"Duplicate feature names found: {twice}.")
    for name, values, declared in columns:
        if any(_is_missing(v) for v in values):
            raise ValueError(f"Target {name} has NaN-like elements.")
        categorical = declared is not None and str(declared) == "category"
        narrow = np.asarray(values.tolist())
        if not categorical and np.issubdtype(narrow.dtype, np.str_):
            odd = {type(v) for v in values if not np.issubdtype(type(v), np.str_)}
            if odd:
                raise ValueError(
                    f"Target {name} has non-string types ({odd}) that cannot be safely converted to a "
                    f"string dtype ({narrow.dtype}).")
        if categorical:
            kind = "classification"
        else:
            kind = "regression" if np.issubdtype(narrow.dtype, np.number) else "classification"
        plan.append((name, narrow, kind))
    return plan


class TreeNodeTransformer(TransformerMixin, BaseEstimator):
    """Shared machinery: fit one tree ensemble per target, transform = node ids of every tree."""

    # subclasses: (regressor class, classifier class) and the two kwargs builders
    def _ensembles(self):
        raise NotImplementedError

    def _trees_per_iteration(self):
        raise NotImplementedError

    def _weights_per_tree(self, X, targets):
        raise NotImplementedError

    def fit(self, X, y):
        X_arr = validate_data(self, X=X, reset=True)
        if y is None:
            raise ValueError(f"{type(self).__name__} requires y to be passed, but the target y is None.")
        plan = _target_plan(y)
        (reg_cls, reg_kw), (clf_cls, clf_kw) = self._ensembles()
        self.estimator_type_dict_ = {name: kind for name, _, kind in plan}
        targets, forests = [], []
        for _, values, kind in plan:
            values = check_array(values, ensure_all_finite=True, dtype=None, ensure_2d=False, estimator=self)
            targets.append(values)
            forests.append((reg_cls(**reg_kw) if kind == "regression" else clf_cls(**clf_kw)).fit(X_arr, values))
        self.estimators_ = forests
        self.n_forests_ = len(forests)
        self.n_trees_per_iteration_ = self._trees_per_iteration()
        self.tree_weights_ = self._weights_per_tree(X_arr, targets)
        return self

    def transform(self, X):
        check_is_fitted(self)
        X_arr = validate_data(self, X=X, reset=False, ensure_min_features=1, ensure_min_samples=1)
        blocks = []
        for forest in self.estimators_:
            ids = forest.apply(X_arr)
            if ids.ndim == 3:  # multi-class boosting: (samples, stages, classes) -> class-major columns
                ids = ids.transpose(0, 2, 1).reshape(ids.shape[0], -1)
            blocks.append(ids)
        return np.hstack(blocks).astype("int64")

    def fit_transform(self, X, y):
        return self.fit(X, y).transform(X)

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.target_tags.required = True
        tags.transformer_tags.preserves_dtype = ["int64"]
        return tags


# constructor parameter -> ensemble keyword: same name unless mapped here
_RF_SHARED = ("n_estimators", "max_depth", "min_samples_split", "min_samples_leaf", "min_weight_fraction_leaf",
              "max_leaf_nodes", "min_impurity_decrease", "bootstrap", "oob_score", "n_jobs", "random_state",
              "verbose", "warm_start", "ccp_alpha", "max_samples", "monotonic_cst")
_RF_REG = {"criterion": "criterion_reg", "max_features": "max_features_reg"}
_RF_CLF = {"criterion": "criterion_clf", "max_features": "max_features_clf", "class_weight": "class_weight_clf"}

_GB_SHARED = ("learning_rate", "n_estimators", "subsample", "criterion", "min_samples_split", "min_samples_leaf",
              "min_weight_fraction_leaf", "max_depth", "min_impurity_decrease", "init", "random_state",
              "max_features", "verbose", "max_leaf_nodes", "warm_start", "validation_fraction",
              "n_iter_no_change", "tol", "ccp_alpha")
_GB_REG = {"loss": "loss_reg", "alpha": "alpha_reg"}
_GB_CLF = {"loss": "loss_clf"}


def _kwargs(obj, shared, mapped):
    out = {name: getattr(obj, name) for name in shared}
    out.update({kw: getattr(obj, attr) for kw, attr in mapped.items()})
    return out


class RFNodeTransformer(TreeNodeTransformer):
    """Node ids across one random forest per target (regressor for numeric targets, classifier
    otherwise); every tree weighs ``1 / n_estimators``."""

    def __init__(self, n_estimators=50, criterion_reg="squared_error", criterion_clf="gini", max_depth=None,
                 min_samples_split=2, min_samples_leaf=5, min_weight_fraction_leaf=0.0, max_features_reg=1.0,
                 max_features_clf="sqrt", max_leaf_nodes=None, min_impurity_decrease=0.0, bootstrap=True,
                 oob_score=False, n_jobs=None, random_state=None, verbose=0, warm_start=False,
                 class_weight_clf=None, ccp_alpha=0.0, max_samples=None, monotonic_cst=None):
        self.n_estimators = n_estimators
        self.criterion_reg = criterion_reg
        self.criterion_clf = criterion_clf
        self.max_depth = max_depth
        self.min_samples_split = min_samples_split
        self.min_samples_leaf = min_samples_leaf
        self.min_weight_fraction_leaf = min_weight_fraction_leaf
        self.max_features_reg = max_features_reg
        self.max_features_clf = max_features_clf
        self.max_leaf_nodes = max_leaf_nodes
        self.min_impurity_decrease = min_impurity_decrease
        self.bootstrap = bootstrap
        self.oob_score = oob_score
        self.n_jobs = n_jobs
        self.random_state = random_state
        self.verbose = verbose
        self.warm_start = warm_start
        self.class_weight_clf = class_weight_clf
        self.ccp_alpha = ccp_alpha
        self.max_samples = max_samples
        self.monotonic_cst = monotonic_cst

    def _ensembles(self):
        return ((RandomForestRegressor, _kwargs(self, _RF_SHARED, _RF_REG)),
                (RandomForestClassifier, _kwargs(self, _RF_SHARED, _RF_CLF)))

    def _trees_per_iteration(self):
        return [1] * self.n_forests_

    def _weights_per_tree(self, X, targets):
        return [np.full(self.n_estimators, 1.0 / self.n_estimators) for _ in range(self.n_forests_)]

    def get_feature_names_out(self, input_features=None):
        check_is_fitted(self, "estimators_")
        return np.asarray([f"rf{i}_tree{j}" for i, f in enumerate(self.estimators_) for j in range(f.n_estimators)],
                          dtype=object)


def _stage_gains(forest, X, target):
    """Share of the training-loss reduction owed to each boosting stage (the reference's
    ``train_improvement``): differences of [initial loss, train_score_], normalised; all ones when
    the loss never moves."""
    from sklearn._loss.loss import HalfBinomialLoss, HalfSquaredError

    if hasattr(forest, "classes_"):
        target = np.searchsorted(forest.classes_, target).astype("float64")
    scale = 2 if isinstance(forest._loss, (HalfSquaredError, HalfBinomialLoss)) else 1
    start = forest._loss(np.asarray(target, dtype=np.float64), forest._raw_predict_init(X)) * scale
    steps = np.diff(np.hstack([start, forest.train_score_]))
    if np.allclose(steps, 0.0):
        return np.ones_like(steps, dtype=np.float64)
    return steps / np.sum(steps)


class GBNodeTransformer(TreeNodeTransformer):
    """Node ids across one gradient-boosting ensemble per target; trees weigh by their stage's share of
    the training-loss reduction (``tree_weighting_method="train_improvement"``) or uniformly."""

    def __init__(self, loss_reg="squared_error", loss_clf="log_loss", learning_rate=0.1, n_estimators=100,
                 subsample=1.0, criterion="friedman_mse", min_samples_split=2, min_samples_leaf=1,
                 min_weight_fraction_leaf=0.0, max_depth=3, min_impurity_decrease=0.0, init=None,
                 random_state=None, max_features=None, alpha_reg=0.9, verbose=0, max_leaf_nodes=None,
                 warm_start=False, validation_fraction=0.1, n_iter_no_change=None, tol=0.0001, ccp_alpha=0.0,
                 tree_weighting_method="train_improvement"):
        self.loss_reg = loss_reg
        self.loss_clf = loss_clf
        self.learning_rate = learning_rate
        self.n_estimators = n_estimators
        self.subsample = subsample
        self.criterion = criterion
        self.min_samples_split = min_samples_split
        self.min_samples_leaf = min_samples_leaf
        self.min_weight_fraction_leaf = min_weight_fraction_leaf
        self.max_depth = max_depth
        self.min_impurity_decrease = min_impurity_decrease
        self.init = init
        self.random_state = random_state
        self.max_features = max_features
        self.alpha_reg = alpha_reg
        self.verbose = verbose
        self.max_leaf_nodes = max_leaf_nodes
        self.warm_start = warm_start
        self.validation_fraction = validation_fraction
        self.n_iter_no_change = n_iter_no_change
        self.tol = tol
        self.ccp_alpha = ccp_alpha
        self.tree_weighting_method = tree_weighting_method

    def _ensembles(self):
        return ((GradientBoostingRegressor, _kwargs(self, _GB_SHARED, _GB_REG)),
                (GradientBoostingClassifier, _kwargs(self, _GB_SHARED, _GB_CLF)))

    def _trees_per_iteration(self):
        return [f.n_trees_per_iteration_ for f in self.estimators_]

    def _weights_per_tree(self, X, targets):
        if self.tree_weighting_method == "uniform":
            counts = [f.n_estimators * f.n_trees_per_iteration_ for f in self.estimators_]
            return [np.full(n, 1.0 / n) for n in counts]
        if self.tree_weighting_method != "train_improvement":
            raise ValueError(f"Invalid tree_weighting_method: {self.tree_weighting_method}. "
                             "Must be 'train_improvement' or 'uniform'.")
        out = []
        for forest, target in zip(self.estimators_, targets):
            gains = _stage_gains(forest, X, target)
            gains = gains / gains.sum()
            out.append(np.tile(gains, forest.n_trees_per_iteration_))
        return out

    def get_feature_names_out(self, input_features=None):
        check_is_fitted(self, "estimators_")
        names = []
        for i, f in enumerate(self.estimators_):
            if f.n_trees_per_iteration_ == 1:
                names += [f"gb{i}_tree{k}" for k in range(f.n_estimators)]
            else:
                names += [f"gb{i}_cls{j}_tree{k}" for j in range(f.n_trees_per_iteration_) for k in range(f.n_estimators)]
        return np.asarray(names, dtype=object)
