"""Generates tests/golden/rounding_golden.npz by IMPORTING THE REFERENCE's
score/utils/matrix_utils.py (numpy + scipy only, importable in the build
container).  Run in the build container only; /root/reference does not exist
on the GPU box.  Inputs cover full-rank, shrunk (det in (0,1), what the
relaxation produces), reflected (det < 0), near-singular and all-zero blocks
in 2-D and 3-D."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from score.utils.matrix_utils import get_matrix_determinant, round_to_special_orthogonal  # noqa: E402

rng = np.random.default_rng(20221007)
out = {}
for d in (2, 3):
    mats = []
    for _ in range(40):  # random full rank
        mats.append(rng.normal(size=(d, d)))
    for _ in range(40):  # shrunk rotations, as the relaxation yields
        q, _r = np.linalg.qr(rng.normal(size=(d, d)))
        if np.linalg.det(q) < 0:
            q[:, -1] *= -1
        mats.append(q * rng.uniform(0.05, 1.0) + 1e-3 * rng.normal(size=(d, d)))
    for _ in range(20):  # reflections
        q, _r = np.linalg.qr(rng.normal(size=(d, d)))
        if np.linalg.det(q) > 0:
            q[:, -1] *= -1
        mats.append(q * rng.uniform(0.2, 1.5))
    mats.append(np.zeros((d, d)))
    mats.append(np.eye(d))
    mats.append(np.eye(d) * 1e-9)
    mats.append(np.diag([1.0] * (d - 1) + [1e-12]))
    M = np.stack(mats)
    R = np.stack([round_to_special_orthogonal(m) for m in M])
    det = np.array([get_matrix_determinant(m) for m in M])
    out[f"in_{d}d"] = M
    out[f"out_{d}d"] = R
    out[f"det_{d}d"] = det
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rounding_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
