"""The reference's transformer tests (REF tests/test_transformers.py:104-213) against this package's
transformers: same inputs, same assertions.  Host logic only (the transformers are fitted with numpy /
scikit-learn on the CPU in the reference too); no GPU needed."""

from __future__ import annotations

import pytest
from numpy.testing import assert_array_equal
from sklearn import config_context
from sklearn.exceptions import NotFittedError
from sklearn.preprocessing import StandardScaler
from sklearn.utils.estimator_checks import parametrize_with_checks

from sknnr_amd.datasets import load_moscow_stjoes
from sknnr_amd.transformers import (
    CCATransformer,
    CCorATransformer,
    GBNodeTransformer,
    MahalanobisTransformer,
    RFNodeTransformer,
    StandardScalerWithDOF,
)

TRANSFORMERS = [StandardScalerWithDOF, MahalanobisTransformer, CCATransformer, CCorATransformer, GBNodeTransformer,
                RFNodeTransformer]
ORDINATION = [CCATransformer, CCorATransformer]

# what the reference expects to fail, for the reason it gives: scikit-learn's checks hand CCA a 1-D y
CCA_NEEDS_2D_Y = [
    "check_estimators_dtypes", "check_dtype_object", "check_estimators_fit_returns_self", "check_pipeline_consistency",
    "check_estimators_overwrite_params", "check_fit_score_takes_y", "check_estimators_pickle",
    "check_transformer_data_not_an_array", "check_transformer_general", "check_transformer_preserve_dtypes",
    "check_methods_sample_order_invariance", "check_methods_subset_invariance", "check_dict_unchanged",
    "check_dont_overwrite_parameters", "check_fit_idempotent", "check_fit_check_is_fitted", "check_n_features_in",
    "check_fit2d_predict1d", "check_fit2d_1sample", "check_estimators_nan_inf", "check_requires_y_none",
    "check_readonly_memmap_input", "check_n_features_in_after_fitting", "check_f_contiguous_array_estimator",
    "check_positive_only_tag_during_fit",
]


def _expected_failures(transformer):
    if isinstance(transformer, CCATransformer):
        return {check: "CCA requires 2D y arrays." for check in CCA_NEEDS_2D_Y}
    return {}


@parametrize_with_checks([cls() for cls in TRANSFORMERS], expected_failed_checks=_expected_failures)
def test_sklearn_transformer_checks(estimator, check):
    check(estimator)


@pytest.mark.parametrize("transformer", TRANSFORMERS)
def test_transformers_get_feature_names_out(transformer):
    X, y = load_moscow_stjoes(return_X_y=True)
    fitted = transformer().fit(X=X, y=y)
    assert fitted.get_feature_names_out().shape == (fitted.transform(X=X).shape[1],)


def _configured(transformer, config_type, output_mode):
    ours, theirs = transformer(), StandardScaler()
    if config_type == "global":
        return ours, theirs, {"transform_output": output_mode}
    ours.set_output(transform=output_mode)
    theirs.set_output(transform=output_mode)
    return ours, theirs, {}


@pytest.mark.parametrize("config_type", ["global", "transformer"])
@pytest.mark.parametrize("output_mode", ["default", "pandas"])
@pytest.mark.parametrize("x_type", ["array", "dataframe"])
@pytest.mark.parametrize("transformer", TRANSFORMERS)
def test_transformer_output_type_consistency(config_type, output_mode, x_type, transformer):
    X, y = load_moscow_stjoes(return_X_y=True, as_frame=x_type == "dataframe")
    ours, theirs, cfg = _configured(transformer, config_type, output_mode)
    with config_context(**cfg):
        assert type(ours.fit_transform(X, y)) is type(theirs.fit_transform(X, y))


@pytest.mark.parametrize("config_type", ["global", "transformer"])
@pytest.mark.parametrize("output_mode", ["default", "pandas"])
@pytest.mark.parametrize("x_type", ["array", "dataframe"])
@pytest.mark.parametrize("transformer", TRANSFORMERS)
def test_transformer_feature_consistency(config_type, output_mode, x_type, transformer):
    X, y = load_moscow_stjoes(return_X_y=True, as_frame=x_type == "dataframe")
    ours, theirs, cfg = _configured(transformer, config_type, output_mode)
    with config_context(**cfg):
        if hasattr(theirs.fit(X, y), "feature_names_in_"):
            assert_array_equal(ours.fit(X, y).feature_names_in_, theirs.fit(X, y).feature_names_in_)
        else:
            assert not hasattr(ours.fit(X, y), "feature_names_in_")


@pytest.mark.parametrize("transformer", TRANSFORMERS)
def test_transformers_raise_notfitted_transform(transformer):
    X, _ = load_moscow_stjoes(return_X_y=True)
    with pytest.raises(NotFittedError):
        transformer().transform(X)


@pytest.mark.parametrize("transformer", ORDINATION)
@pytest.mark.parametrize("n_components", [None, 0, 5])
def test_transformers_n_components(transformer, n_components):
    X, y = load_moscow_stjoes(return_X_y=True)
    t = transformer(n_components=n_components).fit(X, y)
    if n_components is not None:
        assert t.n_components_ == n_components
    assert t.transform(X).shape[1] == t.n_components_


@pytest.mark.parametrize("transformer", ORDINATION)
@pytest.mark.parametrize("n_components", [-1, 1000])
def test_transformers_raise_out_of_range_n_components(transformer, n_components):
    X, y = load_moscow_stjoes(return_X_y=True)
    with pytest.raises(ValueError, match=r"n_components=-?\d+ must be between 0 and \d+"):
        transformer(n_components=n_components).fit(X, y)
