// score_headform.hpp -- the reference's DEFAULT relaxation ("QCQP", /root/reference/score/utils/gurobi_utils.py:139-144)
// brought into the form the semismooth-Newton polish works on.
//
// The QCQP relaxation gives every range measurement a direction vector r_ij in the unit ball (gurobi_utils.py:296-310 the
// variable, :341-344 the constraint ||r_ij||^2 <= 1, :488-496 the cost w ||t_i - t_j - d~ r_ij||^2).  As a conic program that is a
// second-order cone with a CONSTANT head:
//
//     head row   : no entry of A,  b = beta > 0                       (s_head = beta)
//     tail row a : one entry -alpha on a column rho_a,  b = 0         (s_tail = alpha r)
//     column rho_a is private to that row of A; in P it has a diagonal g > 0, the same for every a, and couplings to
//     columns that are not range directions only:  1/2 g |r|^2 + r'(P_ru u + q_r),   |r| <= R = beta / alpha.
//
// For fixed u the minimum over r is closed form: with v(u) = -(P_ru u + q_r),
//
//     min_r = -|v|^2 / (2g) + 1/(2g) max(0, |v| - gR)^2 ,     r* = v / max(g, |v| / R) ,
//
// i.e. the Schur complement of the r block in P plus exactly the term a PRIVATE HEAD variable h leaves behind
// (score_polish_host.hpp): cost 1/2 c h^2 - c theta h with c = 1/g, theta = gR, cone (h, v(u)) in SOC.  A cone may be scaled by
// any lambda > 0; with lambda_k = 1 / max |P_ru| its tail rows get unit coefficients -- for SCORE (h, v) / (2 w d~) = (d_ij,
// t_i - t_j) with cost w (d_ij - d~)^2: the SOCP relaxation's own rows and numbers (unscaled, ADMM needs ten times the
// iterations on them).  headform_reduce
// rewrites a program of this kind into that head form -- the "SOCP" relaxation with rescaled cone rows (gurobi_utils.py:289-294,
// :345-352, :486-487) -- which the solver then treats like any other: ADMM warm-up, Newton polish, chain preconditioner, row
// replication.  headform_expand maps a solution back: u as it is, r = r*(u), the cone rows' slacks (beta, alpha r) and duals
// y_tail = (g r - v) / alpha, y_head = |y_tail| (stationarity of the r columns, complementarity).  Objective values agree
// (the constant g R^2 / 2 - |q_r|^2 / (2g) per cone moves into c0).
//
// Programs that do not have the structure are left alone (headform_reduce returns false); SCORE_QCQP_PLAIN=1 switches the
// rewrite off (the plain ADMM loop on the program as given: rounds 1-4).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/score_hip.h"
#include "score_host.hpp"

namespace score {

struct HeadForm {
    int32_t n0 = 0, m0 = 0;       // the program as given
    int32_t n1 = 0;               // unknowns of the head form (rows: unchanged)
    int32_t T = 0;                // tail rows per cone
    int32_t z = 0;
    std::vector<int32_t> u_cols;  // head-form column -> given column (the first n1 - n_cones columns)
    // per cone
    std::vector<int32_t> row0;    // head row
    std::vector<int32_t> rcol;    // T given columns per cone
    std::vector<double> g, R, alpha, beta, lambda;
    // the head form itself (storage behind `view`)
    std::vector<int32_t> P_ptr, P_col, A_ptr, A_col, soc_dims, chain_ptr, node_first_col;
    std::vector<double> P_val, q, A_val, b;
    score_problem view{};
    size_t n_cones() const { return row0.size(); }
};

inline bool headform_enabled() { return std::getenv("SCORE_QCQP_PLAIN") == nullptr; }

// false: the program is not of the constant-head kind (out is left in an unspecified state)
inline bool headform_no(int where) {
    if (trace_on("headform")) std::fprintf(stderr, "[score setup] head form declined at check %d\n", where);
    return false;
}
inline bool headform_reduce(const score_problem& p, HeadForm& F) {
    F = HeadForm();
    const int32_t n = p.n, m = p.m;
    if (p.n_soc <= 0 || n <= 0 || m <= 0) return headform_no(1);
    F.n0 = n; F.m0 = m; F.z = p.z;
    const int32_t T = p.soc_dims[0] - 1;
    if (T < 1 || T > 3) return headform_no(2);
    F.T = T;
    // column counts of A: a direction column appears in exactly one row
    std::vector<int32_t> acount((size_t)n, 0);
    for (int64_t k = 0; k < p.A_rowptr[m]; ++k) {
        const int32_t c = p.A_col[k];
        if (c < 0 || c >= n) return headform_no(3);
        ++acount[(size_t)c];
    }
    std::vector<int32_t> cone_of((size_t)n, -1);  // direction columns: their cone
    const size_t nc = (size_t)p.n_soc;
    F.row0.resize(nc); F.rcol.resize(nc * T); F.g.resize(nc); F.R.resize(nc); F.alpha.resize(nc); F.beta.resize(nc); F.lambda.assign(nc, 1.0);
    int32_t row = p.z;
    for (size_t k = 0; k < nc; ++k) {
        if (p.soc_dims[k] != T + 1) return headform_no(4);
        if (p.A_rowptr[row + 1] != p.A_rowptr[row]) return headform_no(5);  // constant head
        const double beta = p.b[row];
        if (!(beta > 0.0) || !std::isfinite(beta)) return headform_no(6);
        double alpha = 0.0;
        for (int a = 0; a < T; ++a) {
            const int32_t r = row + 1 + a;
            if (p.A_rowptr[r + 1] - p.A_rowptr[r] != 1 || p.b[r] != 0.0) return headform_no(7);
            const int32_t c = p.A_col[p.A_rowptr[r]];
            const double av = p.A_val[p.A_rowptr[r]];
            if (!(av < 0.0) || acount[(size_t)c] != 1 || cone_of[(size_t)c] >= 0) return headform_no(8);
            if (a == 0) alpha = -av; else if (-av != alpha) return headform_no(9);
            cone_of[(size_t)c] = (int32_t)k;
            F.rcol[k * T + a] = c;
        }
        F.row0[k] = row; F.alpha[k] = alpha; F.beta[k] = beta; F.R[k] = beta / alpha;
        row += T + 1;
    }
    if (row != m) return headform_no(10);
    // the direction columns' rows of P: a positive diagonal (the same for the cone's T columns), no coupling between directions.
    // g = 0 with nothing else in those rows and in q (a measured distance of zero: the direction drops out of the cost): the
    // cone stays as an idle one -- head cost 1/2 h^2, empty tail rows
    for (size_t k = 0; k < nc; ++k) {
        double g = 0.0, big = 0.0, qbig = 0.0;
        for (int a = 0; a < T; ++a) {
            const int32_t c = F.rcol[k * T + a];
            double diag = 0.0;
            for (int32_t e = p.P_rowptr[c]; e < p.P_rowptr[c + 1]; ++e) {
                const int32_t j = p.P_col[e];
                if (j == c) diag = p.P_val[e];
                else if (cone_of[(size_t)j] >= 0 && p.P_val[e] != 0.0) return headform_no(11);
                else big = std::max(big, std::fabs(p.P_val[e]));
            }
            if (!(diag >= 0.0) || !std::isfinite(diag)) return headform_no(12);
            if (a == 0) g = diag; else if (diag != g) return headform_no(13);
            qbig = std::max(qbig, std::fabs(p.q[c]));
        }
        if (g == 0.0 && (big != 0.0 || qbig != 0.0)) return headform_no(16);  // (unbounded below or not convex)
        F.g[k] = g;
        if (big > 0.0 && std::isfinite(1.0 / big)) F.lambda[k] = 1.0 / big;
    }
    // zero-cone rows never touch a direction column (acount == 1 and the tail row holds it)
    // ---- columns of the head form: the others in their order, then one head per cone ----
    std::vector<int32_t> cmap((size_t)n, -1);
    F.u_cols.reserve((size_t)n);
    for (int32_t c = 0; c < n; ++c)
        if (cone_of[(size_t)c] < 0) { cmap[(size_t)c] = (int32_t)F.u_cols.size(); F.u_cols.push_back(c); }
    const int32_t nu = (int32_t)F.u_cols.size();
    F.n1 = nu + (int32_t)nc;
    // ---- P~ = P_uu - P_ur G^-1 P_ru, q~ = q_u - P_ur G^-1 q_r; head rows: c = 1/g on the diagonal, q_h = -R ----
    struct Ent { int32_t col; double val, mag; bool schur; };
    F.P_ptr.assign((size_t)F.n1 + 1, 0);
    F.q.assign((size_t)F.n1, 0.0);
    double c0 = p.c0;
    {   // rows of P~ in parts (one per host thread), written in place after a count
        const int parts = parallel_parts(nu, 8192);
        std::vector<std::vector<int32_t>> pc((size_t)parts);
        std::vector<std::vector<double>> pv((size_t)parts);
        std::vector<int64_t> part_row0((size_t)parts + 1, 0);
        parallel_ranges(nu, 8192, [&](int t, int64_t i0, int64_t i1) {
            if (i1 <= i0) return;
            std::vector<Ent> L;
            std::vector<int32_t> lc;
            std::vector<double> lv;
            lc.reserve((size_t)(p.P_rowptr[F.u_cols[(size_t)(i1 - 1)] + 1] - p.P_rowptr[F.u_cols[(size_t)i0]]));
            lv.reserve(lc.capacity());
            for (int64_t iu = i0; iu < i1; ++iu) {
                const int32_t i = F.u_cols[(size_t)iu];
                L.clear();
                double qi = p.q[i];
                for (int32_t e = p.P_rowptr[i]; e < p.P_rowptr[i + 1]; ++e) {
                    const int32_t j = p.P_col[e];
                    const double v = p.P_val[e];
                    const int32_t k = cone_of[(size_t)j];
                    if (k < 0) { L.push_back(Ent{cmap[(size_t)j], v, std::fabs(v), false}); continue; }
                    if (F.g[(size_t)k] == 0.0) continue;  // (idle cone: the entry is zero)
                    const double f = v / F.g[(size_t)k];  // P[i, rho] / g
                    qi -= f * p.q[j];
                    for (int32_t e2 = p.P_rowptr[j]; e2 < p.P_rowptr[j + 1]; ++e2) {
                        const int32_t j2 = p.P_col[e2];
                        if (cone_of[(size_t)j2] >= 0) continue;  // (the diagonal of rho)
                        const double tt = f * p.P_val[e2];
                        L.push_back(Ent{cmap[(size_t)j2], -tt, std::fabs(tt), true});
                    }
                }
                F.q[(size_t)iu] = qi;
                // stable order by column, then merge; a sum the Schur terms have cancelled to rounding is not an entry
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
                size_t x = 0;
                int32_t len = 0;
                while (x < L.size()) {
                    const int32_t c = L[x].col;
                    double sum = 0.0, mag = 0.0;
                    bool schur = false;
                    for (; x < L.size() && L[x].col == c; ++x) { sum += L[x].val; mag = std::max(mag, L[x].mag); schur |= L[x].schur; }
                    if (schur && std::fabs(sum) <= 1e-12 * mag) continue;
                    lc.push_back(c); lv.push_back(sum);
                    ++len;
                }
                F.P_ptr[(size_t)iu + 1] = len;
            }
            part_row0[(size_t)t] = i0;
            pc[(size_t)t] = std::move(lc);
            pv[(size_t)t] = std::move(lv);
        }, parts);
        for (int32_t iu = 0; iu < nu; ++iu) F.P_ptr[(size_t)iu + 1] += F.P_ptr[(size_t)iu];
        F.P_col.resize((size_t)F.P_ptr[(size_t)nu] + nc);
        F.P_val.resize((size_t)F.P_ptr[(size_t)nu] + nc);
        parallel_ranges(parts, 1, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                if (pc[(size_t)t].empty()) continue;
                const size_t o = (size_t)F.P_ptr[(size_t)part_row0[(size_t)t]];
                std::copy(pc[(size_t)t].begin(), pc[(size_t)t].end(), F.P_col.begin() + (std::ptrdiff_t)o);
                std::copy(pv[(size_t)t].begin(), pv[(size_t)t].end(), F.P_val.begin() + (std::ptrdiff_t)o);
            }
        });
    }
    for (size_t k = 0; k < nc; ++k) {
        const int32_t h = nu + (int32_t)k;
        const size_t at = (size_t)F.P_ptr[(size_t)h];
        F.P_col[at] = h; F.P_val[at] = F.g[k] > 0.0 ? 1.0 / (F.g[k] * F.lambda[k] * F.lambda[k]) : 1.0;
        F.P_ptr[(size_t)h + 1] = (int32_t)at + 1;
        if (F.g[k] == 0.0) continue;
        F.q[(size_t)h] = -F.R[k] / F.lambda[k];
        c0 += 0.5 * F.g[k] * F.R[k] * F.R[k];
        for (int a = 0; a < T; ++a) {
            const double qr = p.q[F.rcol[k * T + a]];
            c0 -= 0.5 * qr * qr / F.g[k];
        }
    }
    // ---- A: zero-cone rows with renamed columns; cone k: (head: -1 on h_k, b = 0), tail a: lambda P[rho_a, u], b = -lambda q[rho_a] ----
    F.A_ptr.assign((size_t)m + 1, 0);
    F.b.assign((size_t)m, 0.0);
    for (int32_t r = 0; r < p.z; ++r) {
        for (int32_t e = p.A_rowptr[r]; e < p.A_rowptr[r + 1]; ++e) { F.A_col.push_back(cmap[(size_t)p.A_col[e]]); F.A_val.push_back(p.A_val[e]); }
        F.A_ptr[(size_t)r + 1] = (int32_t)F.A_col.size();
        F.b[(size_t)r] = p.b[r];
    }
    for (size_t k = 0; k < nc; ++k) {
        const int32_t r0 = F.row0[k];
        F.A_col.push_back(nu + (int32_t)k); F.A_val.push_back(-1.0);
        F.A_ptr[(size_t)r0 + 1] = (int32_t)F.A_col.size();
        for (int a = 0; a < T; ++a) {
            const int32_t rho = F.rcol[k * T + a];
            for (int32_t e = p.P_rowptr[rho]; e < p.P_rowptr[rho + 1] && F.g[k] > 0.0; ++e) {
                const int32_t j = p.P_col[e];
                if (cone_of[(size_t)j] >= 0) continue;
                F.A_col.push_back(cmap[(size_t)j]); F.A_val.push_back(F.lambda[k] * p.P_val[e]);
            }
            F.A_ptr[(size_t)(r0 + 1 + a) + 1] = (int32_t)F.A_col.size();
            F.b[(size_t)(r0 + 1 + a)] = -F.lambda[k] * p.q[rho];
        }
    }
    F.soc_dims.assign(p.soc_dims, p.soc_dims + p.n_soc);
    // ---- hints: chains through the column map; the replication hint survives when every replica loses the same columns ----
    int32_t rep_d = 0, rep_n = 0;
    if (p.n_chains > 0) {
        F.chain_ptr.assign(p.chain_ptr, p.chain_ptr + p.n_chains + 1);
        const int32_t nodes = p.chain_ptr[p.n_chains];
        F.node_first_col.resize((size_t)nodes);
        for (int32_t j = 0; j < nodes; ++j) {
            const int32_t c = p.node_first_col[j];
            if (c < 0 || c + p.block_size > n) return headform_no(14);
            for (int e = 0; e < p.block_size; ++e)
                if (cmap[(size_t)(c + e)] != cmap[(size_t)c] + e) return headform_no(15);  // a chain block must stay a block
            F.node_first_col[(size_t)j] = cmap[(size_t)c];
        }
    }
    if (p.rep_d > 1 && p.rep_n > 0 && (int64_t)p.rep_d * p.rep_n <= n) {
        bool ok = true;
        int32_t per = -1;
        for (int k = 0; k < p.rep_d && ok; ++k) {
            int32_t cnt = 0;
            for (int32_t c = k * p.rep_n; c < (k + 1) * p.rep_n; ++c) cnt += cone_of[(size_t)c] >= 0;
            if (per < 0) per = cnt; else ok = (cnt == per);
        }
        if (ok) { rep_d = p.rep_d; rep_n = p.rep_n - per; }
    }
    score_problem& v = F.view;
    std::memset(&v, 0, sizeof(v));
    v.n = F.n1; v.m = m;
    v.P_rowptr = F.P_ptr.data(); v.P_col = F.P_col.data(); v.P_val = F.P_val.data();
    v.q = F.q.data(); v.c0 = c0;
    v.A_rowptr = F.A_ptr.data(); v.A_col = F.A_col.data(); v.A_val = F.A_val.data(); v.b = F.b.data();
    v.z = p.z; v.n_soc = p.n_soc; v.soc_dims = F.soc_dims.data();
    v.block_size = p.block_size; v.n_chains = p.n_chains;
    v.chain_ptr = p.n_chains > 0 ? F.chain_ptr.data() : nullptr;
    v.node_first_col = p.n_chains > 0 ? F.node_first_col.data() : nullptr;
    v.rep_d = rep_d; v.rep_n = rep_n;
    return true;
}

// (the view's pointers follow the vectors when a HeadForm is moved)
inline void headform_rebind(HeadForm& F) {
    score_problem& v = F.view;
    v.P_rowptr = F.P_ptr.data(); v.P_col = F.P_col.data(); v.P_val = F.P_val.data(); v.q = F.q.data();
    v.A_rowptr = F.A_ptr.data(); v.A_col = F.A_col.data(); v.A_val = F.A_val.data(); v.b = F.b.data();
    v.soc_dims = F.soc_dims.data();
    v.chain_ptr = v.n_chains > 0 ? F.chain_ptr.data() : nullptr;
    v.node_first_col = v.n_chains > 0 ? F.node_first_col.data() : nullptr;
}

// Solution of the head form (xr: n1; yr, sr: m, may be null with their outputs) -> solution of the program as given.
inline void headform_expand(const HeadForm& F, const double* xr, const double* yr, const double* sr, double* x, double* y, double* s) {
    const int32_t nu = (int32_t)F.u_cols.size();
    const int T = F.T;
    if (x)
        for (int32_t iu = 0; iu < nu; ++iu) x[F.u_cols[(size_t)iu]] = xr[iu];
    if (y && yr) for (int32_t r = 0; r < F.z; ++r) y[r] = yr[r];
    if (s && sr) for (int32_t r = 0; r < F.z; ++r) s[r] = sr[r];
    for (size_t k = 0; k < F.n_cones(); ++k) {
        const int32_t r0 = F.row0[k];
        double v[3] = {0, 0, 0}, nv2 = 0.0;
        for (int a = 0; a < T; ++a) {
            const int32_t r = r0 + 1 + a;
            double acc = F.b[(size_t)r];
            for (int32_t e = F.A_ptr[(size_t)r]; e < F.A_ptr[(size_t)r + 1]; ++e) acc -= F.A_val[(size_t)e] * xr[F.A_col[(size_t)e]];
            acc /= F.lambda[k];
            v[a] = acc; nv2 += acc * acc;
        }
        const double nv = std::sqrt(nv2), g = F.g[k];
        const double den = g > 0.0 ? std::max(g, nv / F.R[k]) : 1.0;  // (idle cone: v = 0, r = 0)
        double yt2 = 0.0;
        for (int a = 0; a < T; ++a) {
            const double r = v[a] / den;
            if (x) x[F.rcol[k * T + a]] = r;
            if (s) s[r0 + 1 + a] = F.alpha[k] * r;
            const double yt = (g * r - v[a]) / F.alpha[k];
            if (y) y[r0 + 1 + a] = yt;
            yt2 += yt * yt;
        }
        if (s) s[r0] = F.beta[k];
        if (y) y[r0] = std::sqrt(yt2);
    }
}

// The same record for a QCQP factor graph (score_create_from_graphs, relaxation = 1) without building either program on the
// host: the head form of the graph's QCQP program IS its SOCP program (lambda_k = 1 / (2 w d~): cone rows t_a - t_b, head cost
// w (d_ij - d~)^2 -- see the head of this file), whose columns are the QCQP program's minus the direction columns, heads at
// the end (score_assemble.hpp: replica by replica the pose entries, the landmarks, [the direction components]; then the
// distances).  The handle builds the SOCP program on the device; this record maps its solution to the QCQP program's x, y, s.
inline void headform_from_graph(const score_graph& gr, HeadForm& F) {
    F = HeadForm();
    const int d = gr.dim, D1 = d + 1;
    int64_t Np = 0;
    for (int c = 0; c < gr.n_chains; ++c) Np += gr.chain_len[c];
    const int64_t Nl = gr.n_landmarks, Nr = gr.n_rng;
    const int64_t nrep_s = (Np - 1) * D1 + Nl, nrep_q = nrep_s + Nr, lm_base = (Np - 1) * D1;
    F.n0 = (int32_t)(d * nrep_q); F.n1 = (int32_t)(d * nrep_s + Nr); F.m0 = (int32_t)(Nr * D1); F.T = d; F.z = 0;
    F.u_cols.resize((size_t)(d * nrep_s));
    for (int k = 0; k < d; ++k)
        for (int64_t j = 0; j < nrep_s; ++j) F.u_cols[(size_t)(k * nrep_s + j)] = (int32_t)(k * nrep_q + j);
    const size_t nc = (size_t)Nr;
    F.row0.resize(nc); F.rcol.resize(nc * d); F.g.resize(nc); F.R.assign(nc, 1.0); F.alpha.assign(nc, 1.0); F.beta.assign(nc, 1.0);
    F.lambda.assign(nc, 1.0);
    F.A_ptr.assign((size_t)F.m0 + 1, 0);
    F.b.assign((size_t)F.m0, 0.0);
    F.A_col.reserve(nc * d * 2); F.A_val.reserve(nc * d * 2);
    auto tcol = [&](int64_t v, int k) -> int64_t {  // translation component k of a variable id in the SOCP program (-1: pinned)
        if (v < Np) return v == 0 ? -1 : k * nrep_s + (v - 1) * D1 + d;
        return k * nrep_s + lm_base + (v - Np);
    };
    for (size_t r = 0; r < nc; ++r) {
        const double w = gr.rng_prec[r], dist = gr.rng_dist[r];
        const double cross = 2.0 * w * dist;         // |P_ru| entries of the QCQP program (score_assemble.hpp: 2 w cf[a] cf[b])
        const double g = 2.0 * w * -dist * -dist;    // its diagonal, with the assembler's own rounding
        F.row0[r] = (int32_t)(r * D1);
        F.A_ptr[(size_t)(r * D1) + 1] = (int32_t)F.A_col.size();  // (the head row's entry is not needed for the way back)
        F.g[r] = (g > 0.0 && cross > 0.0) ? g : 0.0;
        if (F.g[r] > 0.0) F.lambda[r] = 1.0 / cross;
        for (int k = 0; k < d; ++k) {
            F.rcol[r * d + k] = (int32_t)(k * nrep_q + nrep_s + (int64_t)r);
            if (F.g[r] > 0.0) {
                const int64_t ca = tcol(gr.rng_a[r], k), cb = tcol(gr.rng_b[r], k);
                if (ca >= 0) { F.A_col.push_back((int32_t)ca); F.A_val.push_back(-1.0); }
                if (cb >= 0) { F.A_col.push_back((int32_t)cb); F.A_val.push_back(1.0); }
            }
            F.A_ptr[(size_t)(r * D1 + 1 + k) + 1] = (int32_t)F.A_col.size();
        }
    }
}

}  // namespace score
