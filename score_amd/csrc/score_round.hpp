// score_round.hpp -- SO(d) rounding of the relaxed rotation blocks, one block per lane / loop iteration.
//
// Counterpart of `round_to_special_orthogonal` (score/utils/matrix_utils.py:59-79) as used by
// `VariableCollection.get_variable_values` (score/utils/gurobi_utils.py:115-125): R = U V' from the SVD
// of M with the last singular direction flipped when det(U V') < 0 -- i.e. the maximiser of tr(R'M)
// over SO(d).  That maximiser has SVD-free forms:
//   d = 2: R(theta), theta = atan2(M10 - M01, M00 + M11);
//   d = 3: the unit quaternion that is the dominant eigenvector of Horn's symmetric 4 x 4 matrix
//          N(M) (cyclic Jacobi, fp64).
// Where the maximiser is not unique (rank-deficient / reflection-like inputs: the reference returns
// whatever its SVD picks) the block is flagged `degenerate` and left to the caller's SVD fallback.
// The same function body is compiled for the device (score_hip.hip: k_round_so) and for the host
// (the CPU twin's loop), so the CPU tests pin the arithmetic to the reference's golden vectors.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SCORE_HD __host__ __device__ __forceinline__
#else
#define SCORE_HD inline
#endif

namespace score {

SCORE_HD void round_so2(const double* M, double* R, int32_t* degenerate) {
    const double cx = M[0] + M[3], sx = M[2] - M[1];
    const double h = std::sqrt(cx * cx + sx * sx);
    const double scale = std::fabs(M[0]) + std::fabs(M[1]) + std::fabs(M[2]) + std::fabs(M[3]);
    const bool bad = !(h > 1e-9 * scale);  // also catches M = 0 and NaN
    const double c = bad ? 1.0 : cx / h, s = bad ? 0.0 : sx / h;
    R[0] = c; R[1] = -s; R[2] = s; R[3] = c;
    *degenerate = bad ? 1 : 0;
}

SCORE_HD void round_so3(const double* M, double* R, int32_t* degenerate) {
    // Horn (1987): tr(R'M) = q' N q for the unit quaternion q = (w, x, y, z) of R
    const double Sxx = M[0], Sxy = M[1], Sxz = M[2], Syx = M[3], Syy = M[4], Syz = M[5], Szx = M[6], Szy = M[7], Szz = M[8];
    // (R maps like M: we maximise sum_ij R_ij M_ij)
    double A[4][4] = {
        {Sxx + Syy + Szz, Szy - Syz, Sxz - Szx, Syx - Sxy},
        {Szy - Syz, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
        {Sxz - Szx, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
        {Syx - Sxy, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) scale += std::fabs(M[i]);
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (!(off > 1e-34 * scale * scale)) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {  // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {  // A <- J' A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > A[best][best]) best = i;
    double second = -1e300;
    for (int i = 0; i < 4; ++i)
        if (i != best && A[i][i] > second) second = A[i][i];
    // the maximiser is unique iff the top eigenvalue is simple
    const bool bad = !(A[best][best] - second > 1e-9 * scale) || !(scale == scale);
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    const double nq = std::sqrt(w * w + x * x + y * y + z * z);
    if (bad || !(nq > 0.0)) { w = 1.0; x = y = z = 0.0; } else { w /= nq; x /= nq; y /= nq; z /= nq; }
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
    *degenerate = bad ? 1 : 0;
}

}  // namespace score
