import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def unpack_mask(bits, width):
    return np.unpackbits(bits, axis=-1, bitorder="little")[..., :width].astype(bool)


class ScalerFromArrays:
    """MinMaxScaler-like view of the affine parameters stored in a fixture."""

    def __init__(self, g, prefix):
        self.min_ = g[prefix + "min_"]
        self.scale_ = g[prefix + "scale_"]
        self.data_min_ = g[prefix + "data_min_"]
        self.data_max_ = g[prefix + "data_max_"]
        self.feature_range = (-1, 1)

    def inverse_transform(self, X):
        X = np.asarray(X)
        dt = np.float32 if X.dtype == np.float32 else np.float64
        X = np.array(X, dtype=dt, copy=True)
        X -= self.min_
        X /= self.scale_
        return X


def params_from_golden(g, prefix="w."):
    import torch
    import pinn_oracle as O
    n_hidden = sum(1 for k in g if k.startswith(prefix + "layers.layer_") and k.endswith(".weight"))
    return [torch.from_numpy(g[prefix + n].copy()) for n in O.param_names(n_hidden)]


@pytest.fixture(scope="session")
def golden():
    return load_golden
