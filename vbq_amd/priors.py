"""Priors that produce code-point tables (stub, filled in below)."""
import numpy as np


def pack_bmshj_params(matrices, biases, factors) -> np.ndarray:
    """Effective BMSHJ2018 parameters -> [C, 43] in the order include/vbq.h documents."""
    C = matrices[0].shape[0]
    parts = []
    for i in range(4):
        parts.append(np.asarray(matrices[i], np.float32).reshape(C, -1))
        parts.append(np.asarray(biases[i], np.float32).reshape(C, -1))
        if i < 3:
            parts.append(np.asarray(factors[i], np.float32).reshape(C, -1))
    out = np.concatenate(parts, axis=1)
    assert out.shape == (C, 43)
    return np.ascontiguousarray(out)
