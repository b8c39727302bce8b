// score_gn.hpp -- local refinement after SCORE: Gauss-Newton with
// Levenberg-Marquardt damping on SE(d)^N x R^(d L), d = 2 or 3 (the reference's model is dimension-generic,
// gurobi_utils.py:37-50), shared by the HIP backend and the CPU twin.
//
// 3-D: the state holds every pose as [R (3 x 3, row-major) | t]; a step lives in the tangent space,
// (omega, v) per pose with the retraction R <- R Exp(omega), t <- t + v, so the unknowns of pose p are the six
// columns 6 (p - 1) .. +5 = [omega | v].  The preconditioner's chains are 3 x 3 as in 2-D: per robot one chain
// over the omega blocks and one over the v blocks (their strong couplings run pose to pose within the same
// kind; the omega-v cross blocks are left to the PCG: 84 against 32 iterations with exact 6 x 6 chain blocks
// on a 4 x 300-pose graph -- and no 6 x 6 variant of the chain kernels to carry).
//
// The reference's README (README.md:63-67) hands the SCORE estimate to a local nonlinear least-squares
// solver (GTSAM) for the maximum-likelihood estimate; this is that step over exactly the factors SCORE
// reads (relative poses with the chordal rotation cost gurobi_utils.py:504-526, ranges :449-501,
// landmark priors :433-446):
//
//   F = sum_rel  kappa |t_j - t_i - R(th_i) t_ij|^2 + tau |R(th_j) - R(th_i) R_ij|_F^2
//     + sum_rng  w (|p_a - p_b| - d)^2  +  sum_prior  w |l - l0|^2
//
// Unknowns u = [th_p, x_p, y_p for poses p = 1.. | landmarks (x, y)]; pose 0 (first pose of chain 0) is
// held where SCORE put it.  Per measurement a small dense block J_e'J_e and gradient J_e'r_e is
// computed (one lane / loop iteration each: `gn_rel_block`, `gn_range_block`, `gn_prior_block`, compiled
// for both sides), every entry of H = J'J and g = J'r then sums its contributions in a fixed order
// (contribution lists built once on the host) -- no atomics, same sums on both backends.  The damped
// normal equations go through the handle's linear mode (score_linear_solve's core: k_factor on the pose
// chains, k_prec_pre + k_spmv PCG).  The LM loop (`gn_levenberg_marquardt`) mirrors
// score_amd/refine.py::_lm_loop, which the tests run beside it with SciPy's sparse LU.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../../include/score_hip.h"
#include "score_host.hpp"

#if defined(__HIPCC__)
#define SCORE_GN_HD __host__ __device__ __forceinline__
#else
#define SCORE_GN_HD inline
#endif

namespace score {

// pose p of the state: (theta, x, y); pose 0 is the pin
SCORE_GN_HD void gn_pose(const double* u, const double* pin, int64_t p, double& th, double& x, double& y) {
    if (p == 0) { th = pin[0]; x = pin[1]; y = pin[2]; return; }
    const double* q = u + 3 * (p - 1);
    th = q[0]; x = q[1]; y = q[2];
}
// point of a range endpoint (variable id: pose or landmark)
SCORE_GN_HD void gn_point(const double* u, const double* pin, int64_t Np, int64_t v, double& x, double& y) {
    if (v < Np) { double th; gn_pose(u, pin, v, th, x, y); return; }
    const double* q = u + 3 * (Np - 1) + 2 * (v - Np);
    x = q[0]; y = q[1];
}

// relative-pose measurement i -> j.  Local unknowns [th_i, x_i, y_i, th_j, x_j, y_j].
// Returns the cost r'r; with H / g non-null also J'J (6 x 6, row-major) and J'r (6).
SCORE_GN_HD double gn_rel_block(double thi, double xi, double yi, double thj, double xj, double yj, const double* tm,
                                const double* Rm, double kappa, double tau, double* H, double* g) {
    const double sk = sqrt(kappa), st = sqrt(tau);
    const double ci = cos(thi), si = sin(thi), cj = cos(thj), sj = sin(thj);
    double r[6];
    r[0] = sk * (xj - xi - (ci * tm[0] - si * tm[1]));
    r[1] = sk * (yj - yi - (si * tm[0] + ci * tm[1]));
    // R_j - R_i Rm, row-major
    r[2] = st * (cj - (ci * Rm[0] - si * Rm[2]));
    r[3] = st * (-sj - (ci * Rm[1] - si * Rm[3]));
    r[4] = st * (sj - (si * Rm[0] + ci * Rm[2]));
    r[5] = st * (cj - (si * Rm[1] + ci * Rm[3]));
    double cost = 0.0;
    for (int k = 0; k < 6; ++k) cost += r[k] * r[k];
    if (!H) return cost;
    double J[6][6];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) J[a][b] = 0.0;
    // d(R_i tm)/dth_i
    const double dRt0 = -si * tm[0] - ci * tm[1], dRt1 = ci * tm[0] - si * tm[1];
    J[0][0] = -sk * dRt0; J[0][1] = -sk; J[0][4] = sk;
    J[1][0] = -sk * dRt1; J[1][2] = -sk; J[1][5] = sk;
    // -(dR_i Rm), dR_i = [[-s, -c], [c, -s]]
    J[2][0] = -st * (-si * Rm[0] - ci * Rm[2]);
    J[3][0] = -st * (-si * Rm[1] - ci * Rm[3]);
    J[4][0] = -st * (ci * Rm[0] - si * Rm[2]);
    J[5][0] = -st * (ci * Rm[1] - si * Rm[3]);
    J[2][3] = st * -sj; J[3][3] = st * -cj; J[4][3] = st * cj; J[5][3] = st * -sj;
    for (int a = 0; a < 6; ++a) {
        double ga = 0.0;
        for (int k = 0; k < 6; ++k) ga += J[k][a] * r[k];
        g[a] = ga;
        for (int b = 0; b < 6; ++b) {
            double h = 0.0;
            for (int k = 0; k < 6; ++k) h += J[k][a] * J[k][b];
            H[a * 6 + b] = h;
        }
    }
    return cost;
}

// range measurement between points a and b.  Local unknowns [xa, ya, xb, yb]; H 4 x 4, g 4.
SCORE_GN_HD double gn_range_block(double xa, double ya, double xb, double yb, double dist, double prec, double* H, double* g) {
    const double sw = sqrt(prec);
    const double dx = xa - xb, dy = ya - yb;
    const double rho = sqrt(dx * dx + dy * dy);
    const double r = sw * (rho - dist);
    if (H) {
        const bool tiny = !(rho > 1e-12);
        const double g0 = tiny ? 0.0 : dx / rho, g1 = tiny ? 0.0 : dy / rho;
        const double J[4] = {sw * g0, sw * g1, -sw * g0, -sw * g1};
        for (int a = 0; a < 4; ++a) {
            g[a] = J[a] * r;
            for (int b = 0; b < 4; ++b) H[a * 4 + b] = J[a] * J[b];
        }
    }
    return r * r;
}

// landmark prior.  Local unknowns [lx, ly]; H holds the two diagonal entries, g 2.
SCORE_GN_HD double gn_prior_block(double lx, double ly, const double* t0, double prec, double* H, double* g) {
    const double sw = sqrt(prec);
    const double r0 = sw * (lx - t0[0]), r1 = sw * (ly - t0[1]);
    if (H) { H[0] = prec; H[1] = prec; g[0] = sw * r0; g[1] = sw * r1; }
    return r0 * r0 + r1 * r1;
}

// ---------------------------------------------------------------------------
// SE(3): state = [R_p (9, row-major), t_p (3)] for every pose p = 0..Np-1 (pose 0 stays where it is), then the
// landmarks (3 each).  Local unknowns of a pose: omega (R <- R Exp(omega)), v (t <- t + v).
// ---------------------------------------------------------------------------
SCORE_GN_HD void gn_so3_exp(const double* w, double* E) {  // Rodrigues; E row-major
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double A, B;
    if (th < 1e-8) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; }
    else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; }
    const double K[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double kk = 0.0;
            for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
            E[i * 3 + j] = (i == j ? 1.0 : 0.0) + A * K[i * 3 + j] + B * kk;
        }
}
// retraction of one pose: out = [R Exp(omega) | t + v]   (step == nullptr: copy)
SCORE_GN_HD void gn_pose3_retract(const double* X, const double* step, double* out) {
    if (!step) { for (int k = 0; k < 12; ++k) out[k] = X[k]; return; }
    double E[9];
    gn_so3_exp(step, E);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += X[i * 3 + k] * E[k * 3 + j];
            out[i * 3 + j] = a;
        }
    for (int k = 0; k < 3; ++k) out[9 + k] = X[9 + k] + step[3 + k];
}
// point of a range endpoint in the 3-D state
SCORE_GN_HD const double* gn_point3(const double* X, int64_t Np, int64_t v) {
    return v < Np ? X + 12 * v + 9 : X + 12 * Np + 3 * (v - Np);
}

// relative-pose measurement i -> j in 3-D.  Local unknowns [om_i, v_i, om_j, v_j] (12).  Residuals (12):
// sqrt(kappa) (t_j - t_i - R_i tm), sqrt(tau) (R_j - R_i Rm) row-major.  With H / g: J'J (12 x 12) and J'r.
SCORE_GN_HD double gn_rel_block3(const double* Xi, const double* Xj, const double* tm, const double* Rm, double kappa, double tau,
                                 double* H, double* g) {
    const double sk = sqrt(kappa), st = sqrt(tau);
    const double* Ri = Xi; const double* ti = Xi + 9;
    const double* Rj = Xj; const double* tj = Xj + 9;
    double r[12];
    for (int a = 0; a < 3; ++a) {
        double rt = 0.0;
        for (int k = 0; k < 3; ++k) rt += Ri[a * 3 + k] * tm[k];
        r[a] = sk * (tj[a] - ti[a] - rt);
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double rr = 0.0;
            for (int k = 0; k < 3; ++k) rr += Ri[a * 3 + k] * Rm[k * 3 + b];
            r[3 + a * 3 + b] = st * (Rj[a * 3 + b] - rr);
        }
    double cost = 0.0;
    for (int k = 0; k < 12; ++k) cost += r[k] * r[k];
    if (!H) return cost;
    // J (12 x 12): columns [om_i 0..2 | v_i 3..5 | om_j 6..8 | v_j 9..11]
    double J[12][12];
    for (int a = 0; a < 12; ++a)
        for (int b = 0; b < 12; ++b) J[a][b] = 0.0;
    // translation rows: d/dv_j = I, d/dv_i = -I, d/dom_i = R_i [tm]x
    const double tx[9] = {0.0, -tm[2], tm[1], tm[2], 0.0, -tm[0], -tm[1], tm[0], 0.0};
    for (int a = 0; a < 3; ++a) {
        J[a][9 + a] = sk;
        J[a][3 + a] = -sk;
        for (int c = 0; c < 3; ++c) {
            double m = 0.0;
            for (int k = 0; k < 3; ++k) m += Ri[a * 3 + k] * tx[k * 3 + c];
            J[a][c] = sk * m;
        }
    }
    // rotation rows: d/dom_j[c] = R_j [e_c]x ; d/dom_i[c] = -R_i [e_c]x Rm
    for (int c = 0; c < 3; ++c) {
        double Ec[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        // [e_c]x : (e_c x) entries
        if (c == 0) { Ec[5] = -1.0; Ec[7] = 1.0; }
        else if (c == 1) { Ec[2] = 1.0; Ec[6] = -1.0; }
        else { Ec[1] = -1.0; Ec[3] = 1.0; }
        double RjE[9], RiE[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double x = 0.0, y = 0.0;
                for (int k = 0; k < 3; ++k) { x += Rj[a * 3 + k] * Ec[k * 3 + b]; y += Ri[a * 3 + k] * Ec[k * 3 + b]; }
                RjE[a * 3 + b] = x; RiE[a * 3 + b] = y;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double y = 0.0;
                for (int k = 0; k < 3; ++k) y += RiE[a * 3 + k] * Rm[k * 3 + b];
                J[3 + a * 3 + b][6 + c] = st * RjE[a * 3 + b];
                J[3 + a * 3 + b][c] = -st * y;
            }
    }
    for (int a = 0; a < 12; ++a) {
        double ga = 0.0;
        for (int k = 0; k < 12; ++k) ga += J[k][a] * r[k];
        g[a] = ga;
        for (int b = 0; b < 12; ++b) {
            double h = 0.0;
            for (int k = 0; k < 12; ++k) h += J[k][a] * J[k][b];
            H[a * 12 + b] = h;
        }
    }
    return cost;
}

// range between points a and b in 3-D.  Local unknowns [pa (3), pb (3)]; H 6 x 6, g 6.
SCORE_GN_HD double gn_range_block3(const double* pa, const double* pb, double dist, double prec, double* H, double* g) {
    const double sw = sqrt(prec);
    const double d[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    const double rho = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double r = sw * (rho - dist);
    if (H) {
        const bool tiny = !(rho > 1e-12);
        double J[6];
        for (int k = 0; k < 3; ++k) { J[k] = tiny ? 0.0 : sw * d[k] / rho; J[3 + k] = -J[k]; }
        for (int a = 0; a < 6; ++a) {
            g[a] = J[a] * r;
            for (int b = 0; b < 6; ++b) H[a * 6 + b] = J[a] * J[b];
        }
    }
    return r * r;
}

// landmark prior in 3-D.  H holds the three diagonal entries, g 3.
SCORE_GN_HD double gn_prior_block3(const double* l, const double* t0, double prec, double* H, double* g) {
    const double sw = sqrt(prec);
    double c = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double r = sw * (l[k] - t0[k]);
        c += r * r;
        if (H) { H[k] = prec; g[k] = sw * r; }
    }
    return c;
}

// ---------------------------------------------------------------------------
// host side: the graph, the pattern of J'J, the contribution lists
// ---------------------------------------------------------------------------
struct GnProblem {
    int dim = 2;
    int64_t Np = 0, Nl = 0, n = 0;
    int dp() const { return dim == 2 ? 3 : 6; }          // unknowns per pose
    int rel_nb() const { return 2 * dp(); }              // local unknowns of a relative-pose block
    int rng_nb() const { return 2 * dim; }
    int64_t state_size() const { return dim == 2 ? n : 12 * Np + 3 * Nl; }  // 3-D: [R | t] of every pose, landmarks
    std::vector<int32_t> chain_len;
    std::vector<int32_t> rel_i, rel_j, rng_a, rng_b, pri_l;
    std::vector<double> rel_t, rel_R, rel_kappa, rel_tau, rng_dist, rng_prec, pri_t, pri_prec;
    double pin[3] = {0, 0, 0};
    int64_t n_rel() const { return (int64_t)rel_i.size(); }
    int64_t n_rng() const { return (int64_t)rng_a.size(); }
    int64_t n_pri() const { return (int64_t)pri_l.size(); }
    // block storage: rel blocks (36 + 6 each), then range blocks (16 + 4), then prior blocks (2 + 2)
    int64_t hblk_size() const { return (int64_t)rel_nb() * rel_nb() * n_rel() + (int64_t)rng_nb() * rng_nb() * n_rng() + dim * n_pri(); }
    int64_t gblk_size() const { return (int64_t)rel_nb() * n_rel() + (int64_t)rng_nb() * n_rng() + dim * n_pri(); }
    int64_t n_meas() const { return n_rel() + n_rng() + n_pri(); }
    // pattern of H (CSR, sorted, with diagonal) and, per entry / per unknown, the block slots it sums
    std::vector<int32_t> hptr, hcol, hc_ptr, hc_slot, gc_ptr, gc_slot, diag_pos;
    std::vector<int32_t> chain_ptr, node_first_col;

    int64_t pose_col(int64_t p) const { return p == 0 ? -1 : dp() * (p - 1); }
    // first column of the translation of a range endpoint (2-D: after theta; 3-D: after omega)
    int64_t point_col(int64_t v) const { return v < Np ? (v == 0 ? -1 : dp() * (v - 1) + (dp() - dim)) : dp() * (Np - 1) + dim * (v - Np); }
};

inline void gn_build(const score_graph& g, GnProblem& P) {
    BuildScope scope;
    if (g.dim != 2 && g.dim != 3) throw std::runtime_error("score_refine: dim must be 2 or 3");
    if (g.n_chains <= 0 || !g.chain_len) throw std::runtime_error("score_refine: no pose chains");
    P.dim = g.dim;
    const int d = g.dim, dp = P.dp(), nbr = P.rel_nb(), nbg = P.rng_nb();
    P.chain_len.assign(g.chain_len, g.chain_len + g.n_chains);
    P.Np = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        if (g.chain_len[c] < 0) throw std::runtime_error("score_refine: negative chain length");
        P.Np += g.chain_len[c];
    }
    if (P.Np == 0 || g.chain_len[0] == 0) throw std::runtime_error("score_refine: no pose variables");
    P.Nl = g.n_landmarks;
    P.n = (int64_t)dp * (P.Np - 1) + (int64_t)d * P.Nl;
    if (P.n >= ((int64_t)1 << 31) / 64) throw std::runtime_error("score_refine: too many unknowns");
    P.rel_i.assign(g.rel_base, g.rel_base + g.n_rel);
    P.rel_j.assign(g.rel_to, g.rel_to + g.n_rel);
    P.rel_t.assign(g.rel_t, g.rel_t + (int64_t)d * g.n_rel);
    P.rel_R.assign(g.rel_R, g.rel_R + (int64_t)d * d * g.n_rel);
    P.rel_kappa.assign(g.rel_kappa, g.rel_kappa + g.n_rel);
    P.rel_tau.assign(g.rel_tau, g.rel_tau + g.n_rel);
    P.rng_a.assign(g.rng_a, g.rng_a + g.n_rng);
    P.rng_b.assign(g.rng_b, g.rng_b + g.n_rng);
    P.rng_dist.assign(g.rng_dist, g.rng_dist + g.n_rng);
    P.rng_prec.assign(g.rng_prec, g.rng_prec + g.n_rng);
    P.pri_l.assign(g.lprior_lm, g.lprior_lm + g.n_lprior);
    P.pri_t.assign(g.lprior_t, g.lprior_t + (int64_t)d * g.n_lprior);
    P.pri_prec.assign(g.lprior_prec, g.lprior_prec + g.n_lprior);
    for (int64_t e = 0; e < g.n_rel; ++e)
        if (P.rel_i[e] < 0 || P.rel_i[e] >= P.Np || P.rel_j[e] < 0 || P.rel_j[e] >= P.Np)
            throw std::runtime_error("score_refine: relative-pose endpoint out of range");
    for (int64_t r = 0; r < g.n_rng; ++r)
        if (P.rng_a[r] < 0 || P.rng_a[r] >= P.Np + P.Nl || P.rng_b[r] < 0 || P.rng_b[r] >= P.Np + P.Nl)
            throw std::runtime_error("score_refine: range endpoint out of range");
    for (int64_t e = 0; e < g.n_lprior; ++e)
        if (P.pri_l[e] < 0 || P.pri_l[e] >= P.Nl) throw std::runtime_error("score_refine: landmark prior out of range");

    // Every block entry that lands on two unknowns is a (row, col, slot) triplet.  The measurement loops run
    // twice over the same code -- a counting pass sizes every row, a filling pass writes (col, slot) pairs in
    // measurement order -- then the rows are ordered by column and merged in parallel (stable: an entry sums
    // its slots in measurement order on every backend).  Gradient lists: the same without columns.
    const int64_t hb_rng = (int64_t)nbr * nbr * g.n_rel, gb_rng = (int64_t)nbr * g.n_rel;
    const int64_t hb_pri = hb_rng + (int64_t)nbg * nbg * g.n_rng, gb_pri = gb_rng + (int64_t)nbg * g.n_rng;
    std::vector<int32_t> hcnt((size_t)P.n + 1, 0), gcnt((size_t)P.n + 1, 0), hfill, gfill, tcol, tslot;
    bool filling = false;
    auto addH = [&](int64_t row, int64_t col, int64_t slot) {
        if (!filling) { ++hcnt[(size_t)row + 1]; return; }
        const int32_t pos = hfill[(size_t)row]++;
        tcol[(size_t)pos] = (int32_t)col;
        tslot[(size_t)pos] = (int32_t)slot;
    };
    auto addG = [&](int64_t row, int64_t slot) {
        if (!filling) { ++gcnt[(size_t)row + 1]; return; }
        P.gc_slot[(size_t)gfill[(size_t)row]++] = (int32_t)slot;
    };
    auto measurements = [&]() {
        for (int64_t e = 0; e < g.n_rel; ++e) {
            const int64_t ci = P.pose_col(P.rel_i[e]), cj = P.pose_col(P.rel_j[e]);
            int64_t L[12];
            for (int a = 0; a < dp; ++a) { L[a] = ci < 0 ? -1 : ci + a; L[dp + a] = cj < 0 ? -1 : cj + a; }
            for (int a = 0; a < nbr; ++a) {
                if (L[a] < 0) continue;
                addG(L[a], (int64_t)nbr * e + a);
                for (int b = 0; b < nbr; ++b)
                    if (L[b] >= 0) addH(L[a], L[b], (int64_t)nbr * nbr * e + a * nbr + b);
            }
        }
        for (int64_t r = 0; r < g.n_rng; ++r) {
            const int64_t ca = P.point_col(P.rng_a[r]), cb = P.point_col(P.rng_b[r]);
            int64_t L[6];
            for (int a = 0; a < d; ++a) { L[a] = ca < 0 ? -1 : ca + a; L[d + a] = cb < 0 ? -1 : cb + a; }
            for (int a = 0; a < nbg; ++a) {
                if (L[a] < 0) continue;
                addG(L[a], gb_rng + (int64_t)nbg * r + a);
                for (int b = 0; b < nbg; ++b)
                    if (L[b] >= 0) addH(L[a], L[b], hb_rng + (int64_t)nbg * nbg * r + a * nbg + b);
            }
        }
        for (int64_t e = 0; e < g.n_lprior; ++e) {
            const int64_t c = (int64_t)dp * (P.Np - 1) + (int64_t)d * (int64_t)P.pri_l[e];
            for (int a = 0; a < d; ++a) {
                addG(c + a, gb_pri + (int64_t)d * e + a);
                addH(c + a, c + a, hb_pri + (int64_t)d * e + a);
            }
        }
        for (int64_t i = 0; i < P.n; ++i) addH(i, i, -1);  // the diagonal always exists
    };
    measurements();  // counting pass
    for (int64_t i = 0; i < P.n; ++i) { hcnt[(size_t)i + 1] += hcnt[(size_t)i]; gcnt[(size_t)i + 1] += gcnt[(size_t)i]; }
    tcol.resize((size_t)hcnt[(size_t)P.n]); tslot.resize((size_t)hcnt[(size_t)P.n]);
    P.gc_ptr = gcnt;
    P.gc_slot.assign((size_t)gcnt[(size_t)P.n], 0);
    hfill.assign(hcnt.begin(), hcnt.end() - 1);
    gfill.assign(gcnt.begin(), gcnt.end() - 1);
    filling = true;
    measurements();  // filling pass
    // rows -> pattern + contribution lists, part by part
    struct Part { std::vector<int32_t> col, ent_len, slot, row_len; int64_t i0 = 0; };
    const int T = parallel_parts(P.n, 4096);
    std::vector<Part> parts((size_t)std::max(1, T));
    P.diag_pos.assign((size_t)P.n, -1);
    std::vector<int32_t> diag_rel((size_t)P.n, -1);  // position of the diagonal within its part
    auto row_weight = [&](int64_t i) {
        const double w = (double)(hcnt[(size_t)i + 1] - hcnt[(size_t)i]);
        return w > 64.0 ? 4.0 * w : w;
    };
    parallel_ranges_balanced(P.n, 4096, row_weight, [&](int t, int64_t i0, int64_t i1) {
        Part W;
        W.i0 = i0;
        W.col.reserve((size_t)(hcnt[(size_t)i1] - hcnt[(size_t)i0]));
        W.slot.reserve((size_t)(hcnt[(size_t)i1] - hcnt[(size_t)i0]));
        struct Ent { int32_t col, slot; };
        std::vector<Ent> L;
        for (int64_t i = i0; i < i1; ++i) {
            L.clear();
            for (int32_t k = hcnt[(size_t)i]; k < hcnt[(size_t)i + 1]; ++k) L.push_back(Ent{tcol[(size_t)k], tslot[(size_t)k]});
            if (L.size() > 64) {
                std::stable_sort(L.begin(), L.end(), [](const Ent& x, const Ent& y) { return x.col < y.col; });
            } else {
                for (size_t x = 1; x < L.size(); ++x) {
                    const Ent e = L[x];
                    size_t y = x;
                    while (y > 0 && L[y - 1].col > e.col) { L[y] = L[y - 1]; --y; }
                    L[y] = e;
                }
            }
            int32_t nent = 0;
            size_t k = 0;
            while (k < L.size()) {
                const int32_t col = L[k].col;
                int32_t ns = 0;
                if (col == (int32_t)i) diag_rel[(size_t)i] = (int32_t)W.col.size();
                for (; k < L.size() && L[k].col == col; ++k)
                    if (L[k].slot >= 0) { W.slot.push_back(L[k].slot); ++ns; }
                W.col.push_back(col);
                W.ent_len.push_back(ns);
                ++nent;
            }
            W.row_len.push_back(nent);
        }
        parts[(size_t)t] = std::move(W);
    }, T);
    {
        std::vector<size_t> eoff(parts.size() + 1, 0), soff(parts.size() + 1, 0);
        for (size_t k = 0; k < parts.size(); ++k) {
            eoff[k + 1] = eoff[k] + parts[k].col.size();
            soff[k + 1] = soff[k] + parts[k].slot.size();
        }
        P.hptr.assign((size_t)P.n + 1, 0);
        P.hcol.resize(eoff.back());
        P.hc_ptr.assign(eoff.back() + 1, 0);
        P.hc_slot.resize(soff.back());
        parallel_ranges((int64_t)parts.size(), 1, [&](int, int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                const Part& W = parts[(size_t)k];
                std::copy(W.col.begin(), W.col.end(), P.hcol.begin() + (std::ptrdiff_t)eoff[(size_t)k]);
                std::copy(W.slot.begin(), W.slot.end(), P.hc_slot.begin() + (std::ptrdiff_t)soff[(size_t)k]);
                int32_t acc = (int32_t)eoff[(size_t)k];
                for (size_t r = 0; r < W.row_len.size(); ++r) {
                    const int64_t i = W.i0 + (int64_t)r;
                    P.diag_pos[(size_t)i] = (int32_t)eoff[(size_t)k] + diag_rel[(size_t)i];
                    acc += W.row_len[r];
                    P.hptr[(size_t)i + 1] = acc;
                }
                int32_t sacc = (int32_t)soff[(size_t)k];
                for (size_t e = 0; e < W.ent_len.size(); ++e) { sacc += W.ent_len[e]; P.hc_ptr[eoff[(size_t)k] + e + 1] = sacc; }
            }
        });
    }
    // chain hint (3 x 3 blocks).  2-D: one chain per robot, node = pose (theta, x, y).  3-D: two chains per robot,
    // the omega blocks and the v blocks of its poses (columns 6 (p - 1) and 6 (p - 1) + 3).  The pinned pose is not a node.
    P.chain_ptr.assign(1, 0);
    P.node_first_col.clear();
    const int kinds = d == 2 ? 1 : 2;
    int64_t p0 = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        for (int kind = 0; kind < kinds; ++kind) {
            int32_t nodes = 0;
            for (int32_t i = 0; i < g.chain_len[c]; ++i) {
                const int64_t p = p0 + i;
                if (p == 0) continue;
                P.node_first_col.push_back((int32_t)(dp * (p - 1) + 3 * kind));
                ++nodes;
            }
            if (nodes > 0) P.chain_ptr.push_back(P.chain_ptr.back() + nodes);
        }
        p0 += g.chain_len[c];
    }
}

struct GnInfo {
    double cost_initial = 0, cost_final = 0, grad_inf = 0;
    int32_t iterations = 0, linear_solves = 0, pcg_iters = 0;
};

// Backend concept:  double eval(const double* u_or_null_for_current, bool with_blocks)  -> cost at the trial
// (or current) point;  assemble() -> |g|_inf (H values and g from the blocks of the current point);
// bool solve(lambda, rel_tol, &pcg_iters) -> step ready;  trial() makes u + step the trial point;
// accept() makes the trial point current.
template <class Backend>
inline void gn_levenberg_marquardt(Backend& be, int max_iters, double tol, double pcg_rel_tol, GnInfo& info) {
    double f = be.eval_current(true);
    info.cost_initial = f;
    double lam = 1e-6;
    double gnorm = INFINITY;
    int it = 0;
    for (it = 1; it <= max_iters; ++it) {
        gnorm = be.assemble();
        if (gnorm <= tol * std::max(1.0, f)) break;
        bool accepted = false;
        double fn = f;
        for (int k = 0; k < 12; ++k) {
            int used = 0;
            const bool ok = be.solve(lam, pcg_rel_tol, &used);
            info.linear_solves += 1;
            info.pcg_iters += used;
            if (!ok) { lam *= 10.0; continue; }
            fn = be.eval_trial();
            if (fn < f) { accepted = true; break; }
            lam *= 10.0;
        }
        if (!accepted) break;
        const double dec = f - fn;
        be.accept();
        f = fn;
        lam = std::max(lam * 0.1, 1e-12);
        (void)be.eval_current(true);
        if (dec <= 1e-14 * std::max(1.0, f)) break;
    }
    info.iterations = std::min(it, max_iters);
    info.cost_final = f;
    info.grad_inf = gnorm;
}

}  // namespace score
