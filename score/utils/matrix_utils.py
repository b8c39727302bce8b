"""The two helpers of the reference's score/utils/matrix_utils.py that sit on the
solve path (:46-79)."""
from score_amd.rounding import get_matrix_determinant, round_to_special_orthogonal  # noqa: F401
