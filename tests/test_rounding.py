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
