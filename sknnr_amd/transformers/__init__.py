from ._ordination import CCA, CCorA
from ._transformers import (
    CCATransformer,
    CCorATransformer,
    ComponentReducerMixin,
    MahalanobisTransformer,
    StandardScalerWithDOF,
)

from ._tree_nodes import GBNodeTransformer, RFNodeTransformer, TreeNodeTransformer

__all__ = [
    "TreeNodeTransformer",
    "RFNodeTransformer",
    "GBNodeTransformer",
    "StandardScalerWithDOF",
    "MahalanobisTransformer",
    "CCATransformer",
    "CCorATransformer",
    "ComponentReducerMixin",
    "CCA",
    "CCorA",
]
