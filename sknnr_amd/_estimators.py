"""The four Euclidean-family estimators: one-method subclasses that pick the feature space.

    EuclideanKNNRegressor    /root/reference/src/sknnr/_euclidean.py:7-56
    MahalanobisKNNRegressor  /root/reference/src/sknnr/_mahalanobis.py:7-57
    GNNRegressor             /root/reference/src/sknnr/_gnn.py:7-70
    MSNRegressor             /root/reference/src/sknnr/_msn.py:7-69
"""

from __future__ import annotations

from ._base import OrdinationKNeighborsRegressor, TransformedKNeighborsRegressor, YFitMixin
from .transformers import (
    CCATransformer,
    CCorATransformer,
    MahalanobisTransformer,
    StandardScalerWithDOF,
)


class EuclideanKNNRegressor(TransformedKNeighborsRegressor):
    """kNN in standardised feature space (unit variance with N-1 degrees of freedom)."""

    def _get_transformer(self):
        return StandardScalerWithDOF(ddof=1)


class MahalanobisKNNRegressor(TransformedKNeighborsRegressor):
    """kNN under the Mahalanobis distance (standardise, then whiten)."""

    def _get_transformer(self):
        return MahalanobisTransformer()


class GNNRegressor(YFitMixin, OrdinationKNeighborsRegressor):
    """Gradient nearest neighbour (Ohmann & Gregory 2002): kNN in CCA ordination space."""

    def _get_transformer(self):
        return CCATransformer(self.n_components)


class MSNRegressor(YFitMixin, OrdinationKNeighborsRegressor):
    """Most similar neighbour (Moeur & Stage 1995): kNN in canonical-correlation space."""

    def _get_transformer(self):
        return CCorATransformer(self.n_components)
