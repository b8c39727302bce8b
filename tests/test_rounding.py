"""SO(d) rounding against golden vectors generated from the reference's own
score/utils/matrix_utils.py (tests/golden/make_rounding_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import score_oracle as so
from score_amd.rounding import get_matrix_determinant, round_to_special_orthogonal


@pytest.mark.parametrize("d", [2, 3])
def test_rounding_matches_reference_vectors(d):
    z = np.load(os.path.join(GOLDEN, "rounding_golden.npz"))
    M, R = z[f"in_{d}d"], z[f"out_{d}d"]
    full_rank = np.abs(z[f"det_{d}d"]) > 1e-6
    got = round_to_special_orthogonal(M)  # batched product path
    np.testing.assert_allclose(got[full_rank], R[full_rank], atol=1e-10)
    for i in range(len(M)):  # oracle restatement, one by one, incl. rank-deficient inputs
        Ro = so.round_to_special_orthogonal(M[i])
        np.testing.assert_allclose(Ro, R[i], atol=1e-10)
        np.testing.assert_allclose(round_to_special_orthogonal(M[i]), got[i], atol=1e-12)
        assert get_matrix_determinant(M[i]) == pytest.approx(z[f"det_{d}d"][i], abs=1e-12)
    # every output is a rotation, also for the rank-deficient inputs
    np.testing.assert_allclose(got @ np.swapaxes(got, 1, 2), np.tile(np.eye(d), (len(M), 1, 1)), atol=1e-9)
    np.testing.assert_allclose(np.linalg.det(got), 1.0, atol=1e-9)


def test_rounding_failure_is_a_value_error():
    with pytest.raises(ValueError, match="Could not round"):
        round_to_special_orthogonal(np.array([[np.nan, 0.0], [0.0, 1.0]]))
    with pytest.raises(AssertionError):
        round_to_special_orthogonal(np.zeros((2, 3)))


def _native_lib(path):
    from score_amd.solver import load_library

    return load_library(path)


@pytest.mark.parametrize("d", [2, 3])
def test_native_rounding_function_matches_reference_vectors(d):
    """csrc/score_round.hpp (the per-block function the HIP kernel runs per lane), here through the
    CPU twin's loop: same golden vectors from the reference's matrix_utils.py."""
    from conftest import TWIN_LIB

    lib = _native_lib(TWIN_LIB)
    z = np.load(os.path.join(GOLDEN, "rounding_golden.npz"))
    M, R = z[f"in_{d}d"], z[f"out_{d}d"]
    full_rank = np.abs(z[f"det_{d}d"]) > 1e-6
    got = round_to_special_orthogonal(M, lib=lib)
    np.testing.assert_allclose(got[full_rank], R[full_rank], atol=1e-10)
    np.testing.assert_allclose(got @ np.swapaxes(got, 1, 2), np.tile(np.eye(d), (len(M), 1, 1)), atol=1e-9)
    np.testing.assert_allclose(np.linalg.det(got), 1.0, atol=1e-9)
    # a larger random stack incl. near-rotations (what the solver hands over), reflections and scalings
    rng = np.random.default_rng(5 + d)
    big = rng.normal(size=(4000, d, d))
    q, _ = np.linalg.qr(rng.normal(size=(1000, d, d)))
    big[:1000] = q + 1e-3 * rng.normal(size=(1000, d, d))          # near-orthogonal, both determinant signs
    big[1000:1500] *= 10.0 ** rng.uniform(-6, 6, size=(500, 1, 1))  # badly scaled
    ref = round_to_special_orthogonal(big) if d == 2 else np.stack([so.round_to_special_orthogonal(m) for m in big])
    sv = np.linalg.svd(big, compute_uv=False)
    # the maximiser is well conditioned unless the two smallest singular values nearly cancel (det < 0)
    det = np.linalg.det(big)
    margin = np.where(det > 0, sv[:, -1] + sv[:, -2], sv[:, -2] - sv[:, -1]) / sv[:, 0]
    ok = margin > 1e-3
    assert ok.sum() > 3500
    np.testing.assert_allclose(round_to_special_orthogonal(big, lib=lib)[ok], ref[ok], atol=1e-9)


def test_native_rounding_degenerate_blocks_fall_back_to_the_svd():
    from conftest import TWIN_LIB

    lib = _native_lib(TWIN_LIB)
    M2 = np.array([[[0.0, 0.0], [0.0, 0.0]], [[1.0, 0.0], [0.0, -1.0]], [[2.0, -1.0], [1.0, 2.0]]])
    got = round_to_special_orthogonal(M2, lib=lib)
    np.testing.assert_allclose(got, round_to_special_orthogonal(M2), atol=1e-12)
    M3 = np.stack([np.zeros((3, 3)), np.diag([1.0, 1.0, 0.0]), np.diag([1.0, -1.0, -1.0]) * 3.0, -np.eye(3)])
    got = round_to_special_orthogonal(M3, lib=lib)
    np.testing.assert_allclose(got @ np.swapaxes(got, 1, 2), np.tile(np.eye(3), (4, 1, 1)), atol=1e-9)
    np.testing.assert_allclose(np.linalg.det(got), 1.0, atol=1e-9)
    np.testing.assert_allclose(got[2], np.diag([1.0, -1.0, -1.0]), atol=1e-12)  # unique: a rotation already
    for i in (0, 3):  # not unique: whatever the SVD formula returns
        np.testing.assert_allclose(got[i], so.round_to_special_orthogonal(M3[i]), atol=1e-9)
    with pytest.raises(ValueError, match="Could not round"):
        round_to_special_orthogonal(np.full((2, 3, 3), np.nan), lib=lib)
    assert round_to_special_orthogonal(np.zeros((0, 3, 3)), lib=lib).shape == (0, 3, 3)
