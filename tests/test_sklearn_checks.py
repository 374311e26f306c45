"""scikit-learn's estimator-check battery over the transformers (CPU) and the estimators (GPU: every
fit / kneighbors / predict of a check is a launch of the HIP library), with the expected-failure lists of
/root/reference/tests/test_transformers.py:58-105 and tests/test_estimators.py:64-134 (checks whose
synthetic data violates CCA's input constraints, and the transformed estimators' n_features_in_)."""

from __future__ import annotations

import pytest
from sklearn.utils.estimator_checks import parametrize_with_checks

import sknnr_amd
from sknnr_amd import transformers as T

_CCA_1D = ["check_estimators_dtypes", "check_dtype_object", "check_estimators_fit_returns_self",
           "check_pipeline_consistency", "check_estimators_overwrite_params", "check_fit_score_takes_y",
           "check_estimators_pickle", "check_methods_sample_order_invariance", "check_methods_subset_invariance",
           "check_dict_unchanged", "check_dont_overwrite_parameters", "check_fit_idempotent",
           "check_fit_check_is_fitted", "check_fit2d_predict1d", "check_fit2d_1sample", "check_estimators_nan_inf",
           "check_positive_only_tag_during_fit"]


def transformer_xfails(tr):
    if isinstance(tr, T.CCATransformer):
        names = _CCA_1D + ["check_transformer_data_not_an_array", "check_transformer_general",
                           "check_transformer_preserve_dtypes", "check_n_features_in", "check_requires_y_none",
                           "check_readonly_memmap_input", "check_n_features_in_after_fitting",
                           "check_f_contiguous_array_estimator"]
        return {n: "CCA requires 2D y arrays." for n in names}
    return {}


def estimator_xfails(est):
    out = {}
    if isinstance(est, sknnr_amd.GNNRegressor):
        names = _CCA_1D + ["check_regressors_train", "check_regressor_data_not_an_array",
                           "check_regressors_no_decision_function", "check_supervised_y_2d", "check_regressors_int"]
        out.update({n: "CCA requires 2D y arrays." for n in names})
        out.update({n: "Row sums must be greater than 0." for n in
                    ("check_regressor_multioutput", "check_readonly_memmap_input", "check_f_contiguous_array_estimator")})
    if isinstance(est, (sknnr_amd.MSNRegressor, sknnr_amd.GNNRegressor, sknnr_amd.RFNNRegressor, sknnr_amd.GBNNRegressor)):
        out.update({n: "Estimator stores transformed n_features_in_" for n in
                    ("check_n_features_in_after_fitting", "check_n_features_in")})
    return out


@parametrize_with_checks([T.StandardScalerWithDOF(), T.MahalanobisTransformer(), T.CCATransformer(), T.CCorATransformer(),
                          T.GBNodeTransformer(), T.RFNodeTransformer()], expected_failed_checks=transformer_xfails)
def test_sklearn_transformer_checks(estimator, check):
    check(estimator)


@pytest.mark.gpu
@pytest.mark.filterwarnings("ignore:divide by zero encountered")
@parametrize_with_checks([sknnr_amd.RawKNNRegressor(), sknnr_amd.EuclideanKNNRegressor(), sknnr_amd.MahalanobisKNNRegressor(),
                          sknnr_amd.MSNRegressor(), sknnr_amd.GNNRegressor(), sknnr_amd.RFNNRegressor(),
                          sknnr_amd.GBNNRegressor()], expected_failed_checks=estimator_xfails)
def test_sklearn_estimator_checks(estimator, check):
    check(estimator)
