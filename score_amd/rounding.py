"""SO(d) rounding of the relaxed rotation blocks.

Counterpart of ``round_to_special_orthogonal`` (score/utils/matrix_utils.py:
59-79) as used by ``VariableCollection.get_variable_values``
(score/utils/gurobi_utils.py:115-125): R = U V^T from the SVD, with the last
singular direction flipped when det(U V^T) < 0, followed by the reference's
validity check (matrix_utils.py:293-318, rtol = atol = 1e-3).  Batched over
all poses instead of one Python call per pose.  For d = 2 the maximiser of
tr(R^T M) over SO(2) -- which is what U diag(1, det(U V^T)) V^T computes -- has
the closed form R(theta), theta = atan2(M10 - M01, M00 + M11); only the inputs
for which that maximiser is not unique (M00 + M11 = M10 - M01 = 0: zero and
scaled-reflection matrices, where the reference returns whatever its SVD
picks) go through the SVD.

``lib`` given (the loaded C-ABI library, ``score_amd.solver.load_library``): the stack is rounded by
``score_round_to_so`` -- on the GPU for the HIP library, one block per lane (closed form for d = 2,
Horn's quaternion eigenvector for d = 3; csrc/score_round.hpp) -- and only the blocks it flags as
degenerate go through the SVD here.  20 000 3-D poses: 65 ms of batched NumPy SVD otherwise.
"""
from __future__ import annotations

import numpy as np


def get_matrix_determinant(mat: np.ndarray) -> float:
    """matrix_utils.py:46-56."""
    mat = np.asarray(mat)
    assert mat.shape[0] == mat.shape[1], "matrix must be square"
    return float(np.linalg.det(mat))


def check_rotation_matrix(R: np.ndarray, assert_test: bool = True) -> None:
    """matrix_utils.py:293-318."""
    d = R.shape[-1]
    RRt = R @ np.swapaxes(R, -1, -2)
    if not np.allclose(RRt, np.eye(d), rtol=1e-3, atol=1e-3):
        if assert_test:
            raise ValueError(f"R is not orthogonal {RRt}")
    det = np.linalg.det(R)
    if np.any(np.abs(det - 1) >= 1e-3):
        if assert_test:
            raise ValueError(f"R det incorrect {det}")


def _svd_round(M: np.ndarray) -> np.ndarray:
    U, _, Vh = np.linalg.svd(M)
    R = U @ Vh
    neg = np.linalg.det(R) < 0
    if np.any(neg):
        Uf = U.copy()
        Uf[neg, :, -1] *= -1.0  # U diag(1,..,1,-1) V^T
        R = np.where(neg[:, None, None], Uf @ Vh, R)
    return R


def _closed_form_round_2d(M: np.ndarray) -> np.ndarray:
    cx = M[:, 0, 0] + M[:, 1, 1]
    sx = M[:, 1, 0] - M[:, 0, 1]
    h = np.hypot(cx, sx)
    scale = np.abs(M).sum(axis=(1, 2))
    degenerate = ~(h > 1e-9 * scale)  # also catches M = 0
    hs = np.where(degenerate, 1.0, h)
    c, s_ = cx / hs, sx / hs
    R = np.empty_like(M)
    R[:, 0, 0] = c; R[:, 0, 1] = -s_
    R[:, 1, 0] = s_; R[:, 1, 1] = c
    if np.any(degenerate):
        R[degenerate] = _svd_round(M[degenerate])
    return R


def _native_round(M: np.ndarray, lib, device: int) -> np.ndarray:
    import ctypes as C

    M = np.ascontiguousarray(M, dtype=np.float64)
    R = np.empty_like(M)
    flags = np.empty(len(M), dtype=np.int32)
    f64p = C.POINTER(C.c_double)
    rc = lib.score_round_to_so(M.shape[-1], len(M), M.ctypes.data_as(f64p), R.ctypes.data_as(f64p),
                               flags.ctypes.data_as(C.POINTER(C.c_int32)), int(device))
    if rc != 0:
        raise RuntimeError(f"score_round_to_so failed: {lib.score_last_error().decode()}")
    # non-degenerate blocks are rotations by construction (cos/sin pair, unit quaternion) for finite
    # input, which the caller has checked; only the SVD fallback needs the reference's validity check
    degenerate = flags != 0
    if np.any(degenerate):
        R[degenerate] = _svd_round(M[degenerate])
        check_rotation_matrix(R[degenerate], assert_test=True)
    if not np.all(np.isfinite(R)):
        raise ValueError("non-finite rotation")
    return R


def round_to_special_orthogonal(mat: np.ndarray, lib=None, device: int = 0) -> np.ndarray:
    """Round one (d, d) matrix or a (N, d, d) stack onto SO(d)."""
    mat = np.asarray(mat, dtype=np.float64)
    single = mat.ndim == 2
    M = mat[None] if single else mat
    if M.shape[-1] != M.shape[-2]:
        raise AssertionError("matrix must be square")
    try:
        if not np.all(np.isfinite(M)):
            raise ValueError("non-finite entries")
        if lib is not None and not single and M.shape[-1] in (2, 3) and len(M):
            R = _native_round(M, lib, device)
        else:
            R = _closed_form_round_2d(M) if (M.shape[-1] == 2 and not single) else _svd_round(M)
            check_rotation_matrix(R, assert_test=True)
    except (ValueError, np.linalg.LinAlgError):
        raise ValueError(f"Could not round matrix to special orthogonal form: {mat}")
    return R[0] if single else R


def finish_device_poses(T: np.ndarray, relaxed: np.ndarray, flags: np.ndarray) -> np.ndarray:
    """The host's share of ``score_read_estimates``: pose blocks the device flagged as degenerate (their SO(d) projection is
    not unique: rank-deficient / reflection-like) take the reference's SVD formula (matrix_utils.py:59-79) and its validity
    check; a non-finite estimate raises the reference's ``ValueError``."""
    d = T.shape[-1] - 1
    try:
        if not np.isfinite(T).all() or not np.isfinite(relaxed).all():
            raise ValueError("non-finite entries")
        degenerate = flags != 0
        if degenerate.any():
            R = _svd_round(np.ascontiguousarray(relaxed[degenerate][:, :, :d]))
            check_rotation_matrix(R, assert_test=True)
            T[degenerate, :d, :d] = R
    except (ValueError, np.linalg.LinAlgError):
        raise ValueError(f"Could not round matrix to special orthogonal form: {relaxed[:, :, :d]}")
    return T
