"""The two example datasets of the reference, shipped as arrays.

``load_moscow_stjoes`` (165 plots x 28 features / 35 targets) and ``load_swo_ecoplot``
(3005 x 18 / 25) return the same ``Dataset`` / ``(X, y)`` shapes as
/root/reference/src/sknnr/datasets/_base.py:250-377; the arrays were captured from the
reference's loaders by ``tests/golden/make_golden.py`` (column 0 of its CSVs is the
int64 plot id, kept here as ``index``).
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@dataclass
class Dataset:
    index: np.ndarray
    data: object
    target: object
    frame: object
    feature_names: list
    target_names: list

    def __repr__(self):
        return (f"Dataset(n={self.data.shape[0]}, features={len(self.feature_names)}, "
                f"targets={len(self.target_names)})")


def _load(name: str, return_X_y: bool, as_frame: bool):
    raw = np.load(os.path.join(_DATA, name + ".npz"), allow_pickle=False)
    index = raw["index"]
    data, target = raw["data"], raw["target"]
    features, targets = [str(s) for s in raw["feature_names"]], [str(s) for s in raw["target_names"]]
    frame = None
    if as_frame:
        try:
            import pandas as pd
        except ImportError as err:  # same behaviour as the reference's _import_pandas
            raise ImportError("pandas must be installed to use `as_frame=True`.") from err
        data = pd.DataFrame(data, columns=features).set_index(index)
        target = pd.DataFrame(target, columns=targets).set_index(index)
        frame = pd.concat([data, target], axis=1).set_index(index)
    if return_X_y:
        return data, target
    return Dataset(index=index, data=data, target=target, frame=frame, feature_names=features,
                   target_names=targets)


def load_moscow_stjoes(return_X_y: bool = False, as_frame: bool = False):
    """Moscow Mountain / St. Joes (Hudak 2010): 165 plots, 28 features, 35 targets."""
    return _load("moscow_stjoes", return_X_y, as_frame)


def load_swo_ecoplot(return_X_y: bool = False, as_frame: bool = False):
    """Southwest Oregon ecoplots: 3005 plots, 18 features, 25 targets."""
    return _load("swo_ecoplot", return_X_y, as_frame)


__all__ = ["Dataset", "load_moscow_stjoes", "load_swo_ecoplot"]
