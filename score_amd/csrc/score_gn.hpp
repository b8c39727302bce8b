// score_gn.hpp -- local refinement after SCORE: Gauss-Newton with
// Levenberg-Marquardt damping on SE(2)^N x R^(2 L), shared by the HIP backend and the CPU twin.
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
// host side: the graph, the pattern of J'J, the contribution lists
// ---------------------------------------------------------------------------
struct GnProblem {
    int64_t Np = 0, Nl = 0, n = 0;
    std::vector<int32_t> chain_len;
    std::vector<int32_t> rel_i, rel_j, rng_a, rng_b, pri_l;
    std::vector<double> rel_t, rel_R, rel_kappa, rel_tau, rng_dist, rng_prec, pri_t, pri_prec;
    double pin[3] = {0, 0, 0};
    int64_t n_rel() const { return (int64_t)rel_i.size(); }
    int64_t n_rng() const { return (int64_t)rng_a.size(); }
    int64_t n_pri() const { return (int64_t)pri_l.size(); }
    // block storage: rel blocks (36 + 6 each), then range blocks (16 + 4), then prior blocks (2 + 2)
    int64_t hblk_size() const { return 36 * n_rel() + 16 * n_rng() + 2 * n_pri(); }
    int64_t gblk_size() const { return 6 * n_rel() + 4 * n_rng() + 2 * n_pri(); }
    int64_t n_meas() const { return n_rel() + n_rng() + n_pri(); }
    // pattern of H (CSR, sorted, with diagonal) and, per entry / per unknown, the block slots it sums
    std::vector<int32_t> hptr, hcol, hc_ptr, hc_slot, gc_ptr, gc_slot, diag_pos;
    std::vector<int32_t> chain_ptr, node_first_col;

    int64_t pose_col(int64_t p) const { return p == 0 ? -1 : 3 * (p - 1); }
    int64_t point_col(int64_t v) const { return v < Np ? (v == 0 ? -1 : 3 * (v - 1) + 1) : 3 * (Np - 1) + 2 * (v - Np); }
};

inline void gn_build(const score_graph& g, GnProblem& P) {
    if (g.dim != 2) throw std::runtime_error("score_refine: 2-D graphs only");
    if (g.n_chains <= 0 || !g.chain_len) throw std::runtime_error("score_refine: no pose chains");
    P.chain_len.assign(g.chain_len, g.chain_len + g.n_chains);
    P.Np = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        if (g.chain_len[c] < 0) throw std::runtime_error("score_refine: negative chain length");
        P.Np += g.chain_len[c];
    }
    if (P.Np == 0 || g.chain_len[0] == 0) throw std::runtime_error("score_refine: no pose variables");
    P.Nl = g.n_landmarks;
    P.n = 3 * (P.Np - 1) + 2 * P.Nl;
    if (P.n >= ((int64_t)1 << 31) / 64) throw std::runtime_error("score_refine: too many unknowns");
    P.rel_i.assign(g.rel_base, g.rel_base + g.n_rel);
    P.rel_j.assign(g.rel_to, g.rel_to + g.n_rel);
    P.rel_t.assign(g.rel_t, g.rel_t + 2 * g.n_rel);
    P.rel_R.assign(g.rel_R, g.rel_R + 4 * g.n_rel);
    P.rel_kappa.assign(g.rel_kappa, g.rel_kappa + g.n_rel);
    P.rel_tau.assign(g.rel_tau, g.rel_tau + g.n_rel);
    P.rng_a.assign(g.rng_a, g.rng_a + g.n_rng);
    P.rng_b.assign(g.rng_b, g.rng_b + g.n_rng);
    P.rng_dist.assign(g.rng_dist, g.rng_dist + g.n_rng);
    P.rng_prec.assign(g.rng_prec, g.rng_prec + g.n_rng);
    P.pri_l.assign(g.lprior_lm, g.lprior_lm + g.n_lprior);
    P.pri_t.assign(g.lprior_t, g.lprior_t + 2 * g.n_lprior);
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
    const int64_t hb_rng = 36 * g.n_rel, gb_rng = 6 * g.n_rel;
    const int64_t hb_pri = hb_rng + 16 * g.n_rng, gb_pri = gb_rng + 4 * g.n_rng;
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
            int64_t L[6];
            for (int a = 0; a < 3; ++a) { L[a] = ci < 0 ? -1 : ci + a; L[3 + a] = cj < 0 ? -1 : cj + a; }
            for (int a = 0; a < 6; ++a) {
                if (L[a] < 0) continue;
                addG(L[a], 6 * e + a);
                for (int b = 0; b < 6; ++b)
                    if (L[b] >= 0) addH(L[a], L[b], 36 * e + a * 6 + b);
            }
        }
        for (int64_t r = 0; r < g.n_rng; ++r) {
            const int64_t ca = P.point_col(P.rng_a[r]), cb = P.point_col(P.rng_b[r]);
            const int64_t L[4] = {ca < 0 ? -1 : ca, ca < 0 ? -1 : ca + 1, cb < 0 ? -1 : cb, cb < 0 ? -1 : cb + 1};
            for (int a = 0; a < 4; ++a) {
                if (L[a] < 0) continue;
                addG(L[a], gb_rng + 4 * r + a);
                for (int b = 0; b < 4; ++b)
                    if (L[b] >= 0) addH(L[a], L[b], hb_rng + 16 * r + a * 4 + b);
            }
        }
        for (int64_t e = 0; e < g.n_lprior; ++e) {
            const int64_t c = 3 * (P.Np - 1) + 2 * (int64_t)P.pri_l[e];
            for (int a = 0; a < 2; ++a) {
                addG(c + a, gb_pri + 2 * e + a);
                addH(c + a, c + a, hb_pri + 2 * e + a);
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
    });
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
    // chain hint: one chain per robot, node = pose (theta, x, y); the pinned pose is not a node
    P.chain_ptr.assign(1, 0);
    P.node_first_col.clear();
    int64_t p = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        int32_t nodes = 0;
        for (int32_t i = 0; i < g.chain_len[c]; ++i, ++p) {
            if (p == 0) continue;
            P.node_first_col.push_back((int32_t)(3 * (p - 1)));
            ++nodes;
        }
        if (nodes > 0) P.chain_ptr.push_back(P.chain_ptr.back() + nodes);
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
