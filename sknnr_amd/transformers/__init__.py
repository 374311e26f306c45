from ._ordination import CCA, CCorA
from ._transformers import (
    CCATransformer,
    CCorATransformer,
    ComponentReducerMixin,
    MahalanobisTransformer,
    StandardScalerWithDOF,
)

__all__ = [
    "StandardScalerWithDOF",
    "MahalanobisTransformer",
    "CCATransformer",
    "CCorATransformer",
    "ComponentReducerMixin",
    "CCA",
    "CCorA",
]
