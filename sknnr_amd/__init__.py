"""sknnr_amd -- MI355X-native backend for sknnr's kneighbors()/predict() hot path.

Same estimator surface as lemma-osu/sknnr (Euclidean family and the tree-node family); the arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/sknnr_hip.h``.
Importing the package is cheap and works without a GPU; fitting or querying an
estimator needs the built library and an MI355X (there is no CPU fallback).
"""

from ._base import RawKNNRegressor
from ._estimators import (
    EuclideanKNNRegressor,
    GNNRegressor,
    MahalanobisKNNRegressor,
    MSNRegressor,
)

from ._config import get_hamming_tie_policy, hamming_tie_policy, set_hamming_tie_policy
from ._forest_nn import GBNNRegressor, RFNNRegressor

__version__ = "0.2.0"

__all__ = [
    "RawKNNRegressor",
    "EuclideanKNNRegressor",
    "MahalanobisKNNRegressor",
    "MSNRegressor",
    "GNNRegressor",
    "RFNNRegressor",
    "GBNNRegressor",
]  # (= REF src/sknnr/__init__.py; the backend's own settings -- hamming_tie_policy & co -- are attributes, not exports)
