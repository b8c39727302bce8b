// score_assemble.hpp -- native model construction: FactorGraphData (flat arrays) -> conic QP.
//
// The C++ counterpart of score_amd/assemble.py, i.e. of the reference's
// `initialize_model` (score/utils/gurobi_utils.py:173-187): variables (:221-310), the
// pinned first pose (:181-183, :316-333; eliminated from the unknowns), second-order cones
// (:336-352), relative-pose costs (:380-430, :504-526), range costs (:449-501) and landmark
// priors (:433-446), written straight into the standard form the solver consumes
//
//     minimise 1/2 x'Px + q'x + c0   s.t.  A x + s = b,  s in SOC(d+1)^Nr
//
// without going through a residual Jacobian: every measurement adds its (d+1)x(d+1) blocks to P
// analytically.  Python model construction was the end-to-end bottleneck once the solve takes
// milliseconds (SURVEY.md hard part 8: ~45 ms per 20-robot graph, and serialised by the GIL over
// a batch); this runs in a few milliseconds and releases the GIL.
//
// Column layout (identical to assemble.py, so ScoreModel.expand()/reduce() apply unchanged):
//   solver space = replica by replica, one replica per matrix row k = 0..d-1 of the poses [R | t]:
//     replica k = for every pose chain the (d+1) entries [R(k,:) t(k)] of every pose except the pinned one
//                 (each block-tridiagonal preconditioner chain is contiguous), then coordinate k of every
//                 landmark, then (QCQP) component k of every range vector r_ij;
//   then the tail: (SOCP) the range variables d_ij.
//   The model only couples the entries of ONE replica at a time, except through the cones
//   (gurobi_utils.py:504-526, :345-352): P = I_d (x) P_row (+ the tail's diagonal), every cone has a head
//   row on the tail and d identical rows, one per replica -- the structure score_problem::rep_d / rep_n
//   announce, which lets the solver stream K_row once for all d right-hand sides.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <stdexcept>
#include <string>
#include <vector>

#include "score_host.hpp"
#include "score_round.hpp"

namespace score {

struct AssembledQP {
    int32_t n = 0, m = 0, dim = 0, relaxation = 0;
    std::vector<int32_t> P_ptr, P_col, A_ptr, A_col, soc_dims, chain_ptr, node_first_col;
    std::vector<double> P_val, q, A_val, b;
    double c0 = 0.0;
    int32_t block_size = 0;
    int32_t rep_n = 0;  // unknowns per replica (replicas = dim)

    void view(score_problem* p) const {
        std::memset(p, 0, sizeof(*p));
        p->n = n; p->m = m;
        p->P_rowptr = P_ptr.data(); p->P_col = P_col.data(); p->P_val = P_val.data();
        p->q = q.data(); p->c0 = c0;
        p->A_rowptr = A_ptr.data(); p->A_col = A_col.data(); p->A_val = A_val.data(); p->b = b.data();
        p->z = 0; p->n_soc = (int32_t)soc_dims.size(); p->soc_dims = soc_dims.data();
        p->block_size = block_size; p->n_chains = (int32_t)chain_ptr.size() - 1;
        p->chain_ptr = chain_ptr.data(); p->node_first_col = node_first_col.data();
        p->rep_d = dim; p->rep_n = rep_n;
    }
};

namespace detail {
struct Trip { int32_t col; double val; };
}

inline void assemble_graph(const score_graph& g, AssembledQP& out) {
    BuildScope scope;
    const int d = g.dim;
    if (d != 2 && d != 3) throw std::runtime_error("score_graph: dim must be 2 or 3");
    if (g.relaxation != 0 && g.relaxation != 1) throw std::runtime_error("score_graph: relaxation must be 0 (SOCP) or 1 (QCQP)");
    if (g.n_chains <= 0 || !g.chain_len) throw std::runtime_error("score_graph: no pose chains");
    const int D1 = d + 1;
    int64_t Np = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        if (g.chain_len[c] < 0) throw std::runtime_error("score_graph: negative chain length");
        Np += g.chain_len[c];
    }
    if (Np == 0 || g.chain_len[0] == 0) throw std::runtime_error("factor graph has no pose variables");
    const int64_t Nl = g.n_landmarks, Nr = g.n_rng;
    const int64_t n_rep = (Np - 1) * D1 + Nl + (g.relaxation == 0 ? 0 : Nr);  // unknowns per replica
    const int64_t lm_base = (Np - 1) * D1, rq_base = lm_base + Nl;          // within a replica
    const int64_t rng_base = (int64_t)d * n_rep;                            // tail (SOCP range variables)
    const int64_t n = rng_base + (g.relaxation == 0 ? Nr : 0);
    if (n >= ((int64_t)1 << 31)) throw std::runtime_error("score_graph: too many unknowns");
    // solver column of entry (k, j) of pose p: k * n_rep + pose_col[p] + j ; pose_col = -1 for the pinned pose
    // (pose 0 of chain 0).  Within a chain the poses (minus the pin) are consecutive nodes.
    std::vector<int64_t> pose_col(Np);
    out = AssembledQP();
    out.dim = d; out.relaxation = g.relaxation;
    out.rep_n = (int32_t)n_rep;
    out.chain_ptr.assign(1, 0);
    {
        int64_t p = 0, col = 0;
        std::vector<int64_t> chain_col, chain_free;
        for (int c = 0; c < g.n_chains; ++c) {
            const int64_t L = g.chain_len[c];
            const int64_t Lfree = L - (c == 0 ? 1 : 0);
            for (int64_t i = 0; i < L; ++i, ++p) {
                if (c == 0 && i == 0) { pose_col[p] = -1; continue; }
                pose_col[p] = col + (i - (c == 0 ? 1 : 0)) * D1;
            }
            chain_col.push_back(col);
            chain_free.push_back(Lfree);
            col += Lfree * D1;
        }
        for (int k = 0; k < d; ++k)  // chains replica by replica
            for (int c = 0; c < g.n_chains; ++c) {
                if (chain_free[(size_t)c] <= 0) continue;
                for (int64_t node = 0; node < chain_free[(size_t)c]; ++node)
                    out.node_first_col.push_back((int32_t)(k * n_rep + chain_col[(size_t)c] + node * D1));
                out.chain_ptr.push_back(out.chain_ptr.back() + (int32_t)chain_free[(size_t)c]);
            }
    }
    out.block_size = D1;
    auto pcol = [&](int64_t p, int k, int j) -> int64_t { return pose_col[p] < 0 ? -1 : k * n_rep + pose_col[p] + j; };
    // translation column k of a variable id (pose or landmark); -1 for the pinned pose (value 0)
    auto tcol = [&](int64_t v, int k) -> int64_t {
        if (v < 0 || v >= Np + Nl) throw std::runtime_error("score_graph: range endpoint out of range");
        if (v < Np) return pcol(v, k, d);
        return k * n_rep + lm_base + (v - Np);
    };
    auto rqcol = [&](int64_t r, int k) -> int64_t { return k * n_rep + rq_base + r; };  // QCQP range vector component
    // ---- P, q, c0.  The measurement loops run twice over the same code: a counting pass sizes every
    //      row, a filling pass writes (column, value) pairs into one flat array; rows are then sorted
    //      and merged in parallel (rows are short: a few (d+1)-blocks each). ----
    PhaseTimer pt(trace_on("assemble"));  // phase marks on stderr
    pt.mark("assemble: layout");
    out.q.assign((size_t)n, 0.0);
    double c0 = 0.0;
    std::vector<int32_t> cnt((size_t)n + 1, 0);
    std::vector<int32_t> tcols;
    std::vector<double> tvals;
    std::vector<int32_t> fill;
    bool filling = false;
    // P = I_d (x) P_row (+ the tail's diagonal): the d replicas of a row hold the same values at shifted columns, so only
    // replica 0's rows (r < n_rep) and the tail's are gathered, sorted and merged; the others are copies (below).  q and
    // c0 are not replicated (the pinned pose is [I | 0]: its row k is e_k) and come from every replica's terms.
    auto addP = [&](int64_t r, int64_t c, double v) {
        if (r >= n_rep && r < rng_base) return;
        if (!filling) { ++cnt[(size_t)r + 1]; return; }
        const int32_t pos = fill[(size_t)r]++;
        tcols[(size_t)pos] = (int32_t)c;
        tvals[(size_t)pos] = v;
    };
    auto measurements = [&]() {
        // relative-pose terms.  Per matrix row k (the d rows of a pose are decoupled and share their blocks):
        //   residual = u_j - G u_i,  u = (R[k,0..d-1], t[k]),  G[c][l] = Rm[l][c] (c < d), G[d][l] = tm[l], G[d][d] = 1,
        //   weights W = diag(tau, ..., tau, kappa).   cost = res' W res.
        for (int64_t e = 0; e < g.n_rel; ++e) {
            const int64_t i = g.rel_base[e], j = g.rel_to[e];
            if (i < 0 || i >= Np || j < 0 || j >= Np) throw std::runtime_error("score_graph: relative-pose endpoint out of range");
            const double* tm = g.rel_t + e * d;
            const double* Rm = g.rel_R + e * d * d;
            const double kap = g.rel_kappa[e], tau = g.rel_tau[e];
            double G[4][4] = {{0}}, W[4];
            for (int c = 0; c < d; ++c) {
                for (int l = 0; l < d; ++l) G[c][l] = Rm[l * d + c];
                W[c] = tau;
            }
            for (int l = 0; l < d; ++l) G[d][l] = tm[l];
            G[d][d] = 1.0;
            W[d] = kap;
            double GtWG[4][4], GtW[4][4];
            for (int a = 0; a < D1; ++a)
                for (int b2 = 0; b2 < D1; ++b2) {
                    GtW[a][b2] = G[b2][a] * W[b2];
                    double s_ = 0;
                    for (int c = 0; c < D1; ++c) s_ += G[c][a] * W[c] * G[c][b2];
                    GtWG[a][b2] = s_;
                }
            for (int k = 0; k < d; ++k) {
                const int64_t ci = pcol(i, k, 0), cj = pcol(j, k, 0);
                double ui[4] = {0, 0, 0, 0}, uj[4] = {0, 0, 0, 0};  // pinned pose: [I | 0]
                ui[k] = 1.0; uj[k] = 1.0;
                if (cj >= 0) {
                    for (int a = 0; a < D1; ++a) addP(cj + a, cj + a, 2.0 * W[a]);
                    if (ci >= 0) {
                        for (int a = 0; a < D1; ++a)
                            for (int b2 = 0; b2 < D1; ++b2) {
                                const double v = -2.0 * GtW[a][b2];  // block (i, j) = -2 G'W ; (j, i) its transpose
                                if (G[b2][a] != 0.0) { addP(ci + a, cj + b2, v); addP(cj + b2, ci + a, v); }
                            }
                    } else if (filling) {  // u_i fixed: cost (u_j - G u_i)' W (u_j - G u_i)
                        for (int a = 0; a < D1; ++a) {
                            double gu = 0;
                            for (int l = 0; l < D1; ++l) gu += G[a][l] * ui[l];
                            out.q[(size_t)(cj + a)] += -2.0 * W[a] * gu;
                            c0 += W[a] * gu * gu;
                        }
                    }
                }
                if (ci >= 0) {
                    for (int a = 0; a < D1; ++a)
                        for (int b2 = 0; b2 < D1; ++b2) addP(ci + a, ci + b2, 2.0 * GtWG[a][b2]);
                    if (cj < 0 && filling) {  // u_j fixed
                        for (int a = 0; a < D1; ++a) {
                            double s_ = 0;
                            for (int c = 0; c < D1; ++c) s_ += GtW[a][c] * uj[c];
                            out.q[(size_t)(ci + a)] += -2.0 * s_;
                        }
                        for (int c = 0; c < D1; ++c) c0 += W[c] * uj[c] * uj[c];
                    }
                }
                if (ci < 0 && cj < 0 && filling) {
                    for (int a = 0; a < D1; ++a) {
                        double gu = 0;
                        for (int l = 0; l < D1; ++l) gu += G[a][l] * ui[l];
                        c0 += W[a] * (uj[a] - gu) * (uj[a] - gu);
                    }
                }
            }
        }
        // range costs
        for (int64_t r = 0; r < Nr; ++r) {
            const double w = g.rng_prec[r], dist = g.rng_dist[r];
            if (g.relaxation == 0) {  // w (d_ij - dist)^2    (:487)
                const int64_t c = rng_base + r;
                addP(c, c, 2.0 * w);
                if (filling) {
                    out.q[(size_t)c] += -2.0 * w * dist;
                    c0 += w * dist * dist;
                }
            } else {  // w || t_a - t_b - dist r ||^2    (:489-496)
                for (int k = 0; k < d; ++k) {
                    const int64_t cs[3] = {tcol(g.rng_a[r], k), tcol(g.rng_b[r], k), rqcol(r, k)};
                    const double cf[3] = {1.0, -1.0, -dist};
                    for (int a = 0; a < 3; ++a)
                        for (int b2 = 0; b2 < 3; ++b2)
                            if (cs[a] >= 0 && cs[b2] >= 0) addP(cs[a], cs[b2], 2.0 * w * cf[a] * cf[b2]);
                    // (a pinned translation is zero: no linear or constant term)
                }
            }
        }
        // landmark priors  w (l[k] - tv[k])^2    (:433-446)
        for (int64_t e = 0; e < g.n_lprior; ++e) {
            const int64_t l = g.lprior_lm[e];
            if (l < 0 || l >= Nl) throw std::runtime_error("score_graph: landmark prior out of range");
            const double w = g.lprior_prec[e];
            for (int k = 0; k < d; ++k) {
                const int64_t c = k * n_rep + lm_base + l;
                const double tv = g.lprior_t[e * d + k];
                addP(c, c, 2.0 * w);
                if (filling) {
                    out.q[(size_t)c] += -2.0 * w * tv;
                    c0 += w * tv * tv;
                }
            }
        }
    };
    measurements();  // counting pass
    pt.mark("assemble: counting pass");
    for (int64_t i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
    tcols.resize((size_t)cnt[(size_t)n]);
    tvals.resize((size_t)cnt[(size_t)n]);
    fill.assign(cnt.begin(), cnt.end() - 1);
    filling = true;
    measurements();  // filling pass
    pt.mark("assemble: filling pass");
    out.c0 = c0;
    // sort + merge every stored row (stable: equal columns are summed in the order they were met)
    out.n = (int32_t)n;
    out.P_ptr.assign((size_t)n + 1, 0);
    {
        const int64_t n_tail = n - rng_base, n_act = n_rep + n_tail;  // active rows: replica 0, then the tail
        auto act_row = [&](int64_t a) { return a < n_rep ? a : a - n_rep + rng_base; };
        const int T = parallel_parts(n_act, 8192);
        std::vector<std::vector<int32_t>> pc(T);
        std::vector<std::vector<double>> pv(T);
        std::vector<int32_t> len((size_t)n_act, 0);
        parallel_ranges(n_act, 8192, [&](int t, int64_t a0, int64_t a1) {
            // thread-local buffers, handed over at the end: the headers of pc[t] / pv[t] share cache lines
            // with their neighbours', and every push_back writes the header (measured: the 16-thread phase
            // was slower than one thread)
            std::vector<int32_t> lc;
            std::vector<double> lv;
            size_t ub = 0;
            for (int64_t a = a0; a < a1; ++a) ub += (size_t)(cnt[(size_t)act_row(a) + 1] - cnt[(size_t)act_row(a)]);
            lc.reserve(ub);
            lv.reserve(ub);
            std::vector<detail::Trip> L;
            for (int64_t a = a0; a < a1; ++a) {
                const int64_t i = act_row(a);
                L.clear();
                for (int32_t k = cnt[(size_t)i]; k < cnt[(size_t)i + 1]; ++k) L.push_back(detail::Trip{tcols[(size_t)k], tvals[(size_t)k]});
                // stable order by column: insertion sort for the usual short rows
                if (L.size() > 64) {
                    std::stable_sort(L.begin(), L.end(), [](const detail::Trip& x, const detail::Trip& y) { return x.col < y.col; });
                } else {
                    for (size_t x = 1; x < L.size(); ++x) {
                        const detail::Trip e = L[x];
                        size_t y = x;
                        while (y > 0 && L[y - 1].col > e.col) { L[y] = L[y - 1]; --y; }
                        L[y] = e;
                    }
                }
                int32_t c_ = 0;
                size_t x = 0;
                while (x < L.size()) {
                    const int32_t c = L[x].col;
                    double s_ = 0;
                    for (; x < L.size() && L[x].col == c; ++x) s_ += L[x].val;
                    lc.push_back(c); lv.push_back(s_);
                    ++c_;
                }
                len[(size_t)a] = c_;
            }
            pc[t] = std::move(lc);
            pv[t] = std::move(lv);
        }, T);
        pt.mark("assemble:   rows (parallel)");
        // row pointers: replica k's rows repeat replica 0's lengths
        for (int k = 0; k < d; ++k)
            for (int64_t a = 0; a < n_rep; ++a) out.P_ptr[(size_t)(k * n_rep + a) + 1] = len[(size_t)a];
        for (int64_t a = n_rep; a < n_act; ++a) out.P_ptr[(size_t)act_row(a) + 1] = len[(size_t)a];
        for (int64_t i = 0; i < n; ++i) out.P_ptr[(size_t)i + 1] += out.P_ptr[(size_t)i];
        out.P_col.resize((size_t)out.P_ptr[(size_t)n]);
        out.P_val.resize((size_t)out.P_ptr[(size_t)n]);
        // the parts, in order, are replica 0's entries followed by the tail's: every part is copied into place once per
        // replica (columns shifted by the replica's offset); a part may straddle the end of replica 0
        std::vector<int64_t> part_a0((size_t)T + 1);
        for (int t = 0; t <= T; ++t) part_a0[(size_t)t] = n_act * t / T;  // the ranges of parallel_ranges are contiguous and ordered
        const int64_t e_rep = out.P_ptr[(size_t)n_rep];                    // entries of one replica
        parallel_ranges(T, 1, [&](int, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                const std::vector<int32_t>& lc = pc[(size_t)t];
                const std::vector<double>& lv = pv[(size_t)t];
                if (lc.empty()) continue;
                const int64_t a0 = part_a0[(size_t)t];
                const int64_t o0 = a0 < n_rep ? out.P_ptr[(size_t)a0] : e_rep + (out.P_ptr[(size_t)act_row(a0)] - out.P_ptr[(size_t)rng_base]);  // offset within [replica 0 | tail]
                const int64_t in_rep = std::max<int64_t>(0, std::min<int64_t>((int64_t)lc.size(), e_rep - o0));  // entries of this part that belong to replica 0
                for (int k = 0; k < d; ++k) {
                    const int32_t shift = (int32_t)(k * n_rep);
                    int32_t* dc = &out.P_col[(size_t)(k * e_rep + o0)];
                    if (in_rep > 0) {
                        for (int64_t e = 0; e < in_rep; ++e) dc[e] = lc[(size_t)e] + shift;
                        std::memcpy(&out.P_val[(size_t)(k * e_rep + o0)], lv.data(), (size_t)in_rep * sizeof(double));
                    }
                }
                if ((int64_t)lc.size() > in_rep) {  // tail entries: behind all replicas
                    const int64_t ot = (int64_t)d * e_rep + (o0 + in_rep - e_rep);
                    std::memcpy(&out.P_col[(size_t)ot], lc.data() + in_rep, (size_t)((int64_t)lc.size() - in_rep) * sizeof(int32_t));
                    std::memcpy(&out.P_val[(size_t)ot], lv.data() + in_rep, (size_t)((int64_t)lv.size() - in_rep) * sizeof(double));
                }
            }
        });
        pt.mark("assemble:   replicate + concatenate");
    }
    pt.mark("assemble: sort + merge rows");
    // ---- cones (:336-352): s = b - A x ----
    const int64_t m = Nr * D1;
    out.m = (int32_t)m;
    out.b.assign((size_t)m, 0.0);
    out.A_ptr.assign(1, 0);
    out.soc_dims.assign((size_t)Nr, D1);
    for (int64_t r = 0; r < Nr; ++r) {
        if (g.relaxation == 0) {
            // (d_ij, t_a - t_b) in SOC: A = -[e_d ; e_ta - e_tb], b = 0
            out.A_col.push_back((int32_t)(rng_base + r)); out.A_val.push_back(-1.0);
            out.A_ptr.push_back((int32_t)out.A_col.size());
            for (int k = 0; k < d; ++k) {
                int64_t ca = tcol(g.rng_a[r], k), cb = tcol(g.rng_b[r], k);
                double va = -1.0, vb = 1.0;
                if (ca >= 0 && cb >= 0 && ca == cb) throw std::runtime_error("score_graph: range between a variable and itself");
                if (ca >= 0 && cb >= 0 && cb < ca) { std::swap(ca, cb); std::swap(va, vb); }
                if (ca >= 0) { out.A_col.push_back((int32_t)ca); out.A_val.push_back(va); }
                if (cb >= 0) { out.A_col.push_back((int32_t)cb); out.A_val.push_back(vb); }
                out.A_ptr.push_back((int32_t)out.A_col.size());
            }
        } else {
            // (1, r_ij) in SOC: A = -[0 ; I], b = (1, 0..0)
            out.b[(size_t)(r * D1)] = 1.0;
            out.A_ptr.push_back((int32_t)out.A_col.size());
            for (int k = 0; k < d; ++k) {
                out.A_col.push_back((int32_t)rqcol(r, k)); out.A_val.push_back(-1.0);
                out.A_ptr.push_back((int32_t)out.A_col.size());
            }
        }
    }
    pt.mark("assemble: cones");
}

// The SKELETON of the program assemble_graph builds: sizes, cone dimensions, the chain and replication hints and the
// objective constant c0 -- everything a handle's host side needs when the matrices themselves are built on the device from the
// graph's arrays (score_create_from_graphs; score_setup_device.hpp, k_ga_*).  Same checks, same messages as assemble_graph;
// c0 is accumulated in assemble_graph's order (measurement by measurement), so it is the same double.
inline void graph_skeleton(const score_graph& g, AssembledQP& out) {
    const int d = g.dim;
    if (d != 2 && d != 3) throw std::runtime_error("score_graph: dim must be 2 or 3");
    if (g.relaxation != 0 && g.relaxation != 1) throw std::runtime_error("score_graph: relaxation must be 0 (SOCP) or 1 (QCQP)");
    if (g.n_chains <= 0 || !g.chain_len) throw std::runtime_error("score_graph: no pose chains");
    const int D1 = d + 1;
    int64_t Np = 0;
    for (int c = 0; c < g.n_chains; ++c) {
        if (g.chain_len[c] < 0) throw std::runtime_error("score_graph: negative chain length");
        Np += g.chain_len[c];
    }
    if (Np == 0 || g.chain_len[0] == 0) throw std::runtime_error("factor graph has no pose variables");
    const int64_t Nl = g.n_landmarks, Nr = g.n_rng;
    const int64_t n_rep = (Np - 1) * D1 + Nl + (g.relaxation == 0 ? 0 : Nr);
    const int64_t rng_base = (int64_t)d * n_rep;
    const int64_t n = rng_base + (g.relaxation == 0 ? Nr : 0);
    if (n >= ((int64_t)1 << 31)) throw std::runtime_error("score_graph: too many unknowns");
    out = AssembledQP();
    out.dim = d; out.relaxation = g.relaxation;
    out.n = (int32_t)n; out.m = (int32_t)(Nr * D1);
    out.rep_n = (int32_t)n_rep;
    out.block_size = D1;
    out.soc_dims.assign((size_t)Nr, D1);
    out.chain_ptr.assign(1, 0);
    {
        std::vector<int64_t> chain_col, chain_free;
        int64_t col = 0;
        for (int c = 0; c < g.n_chains; ++c) {
            const int64_t Lfree = (int64_t)g.chain_len[c] - (c == 0 ? 1 : 0);
            chain_col.push_back(col);
            chain_free.push_back(Lfree);
            col += Lfree * D1;
        }
        for (int k = 0; k < d; ++k)
            for (int c = 0; c < g.n_chains; ++c) {
                if (chain_free[(size_t)c] <= 0) continue;
                for (int64_t node = 0; node < chain_free[(size_t)c]; ++node)
                    out.node_first_col.push_back((int32_t)(k * n_rep + chain_col[(size_t)c] + node * D1));
                out.chain_ptr.push_back(out.chain_ptr.back() + (int32_t)chain_free[(size_t)c]);
            }
    }
    double c0 = 0.0;
    for (int64_t e = 0; e < g.n_rel; ++e) {
        const int64_t i = g.rel_base[e], j = g.rel_to[e];
        if (i < 0 || i >= Np || j < 0 || j >= Np) throw std::runtime_error("score_graph: relative-pose endpoint out of range");
        if (i != 0 && j != 0) continue;  // (only measurements at the pinned pose leave a constant)
        const double* tm = g.rel_t + e * d;
        const double* Rm = g.rel_R + e * d * d;
        const double kap = g.rel_kappa[e], tau = g.rel_tau[e];
        double G[4][4] = {{0}}, W[4];
        for (int c = 0; c < d; ++c) {
            for (int l = 0; l < d; ++l) G[c][l] = Rm[l * d + c];
            W[c] = tau;
        }
        for (int l = 0; l < d; ++l) G[d][l] = tm[l];
        G[d][d] = 1.0;
        W[d] = kap;
        for (int k = 0; k < d; ++k) {
            double ui[4] = {0, 0, 0, 0}, uj[4] = {0, 0, 0, 0};
            ui[k] = 1.0; uj[k] = 1.0;
            if (j != 0 && i == 0) {
                for (int a = 0; a < D1; ++a) {
                    double gu = 0;
                    for (int l = 0; l < D1; ++l) gu += G[a][l] * ui[l];
                    c0 += W[a] * gu * gu;
                }
            }
            if (i != 0 && j == 0)
                for (int c = 0; c < D1; ++c) c0 += W[c] * uj[c] * uj[c];
            if (i == 0 && j == 0)
                for (int a = 0; a < D1; ++a) {
                    double gu = 0;
                    for (int l = 0; l < D1; ++l) gu += G[a][l] * ui[l];
                    c0 += W[a] * (uj[a] - gu) * (uj[a] - gu);
                }
        }
    }
    for (int64_t r = 0; r < Nr; ++r) {
        const int64_t va = g.rng_a[r], vb = g.rng_b[r];
        if (va < 0 || va >= Np + Nl || vb < 0 || vb >= Np + Nl) throw std::runtime_error("score_graph: range endpoint out of range");
        if (g.relaxation == 0) {
            if (va == vb && va != 0) throw std::runtime_error("score_graph: range between a variable and itself");
            const double w = g.rng_prec[r], dist = g.rng_dist[r];
            c0 += w * dist * dist;
        }
    }
    for (int64_t e = 0; e < g.n_lprior; ++e) {
        const int64_t l = g.lprior_lm[e];
        if (l < 0 || l >= Nl) throw std::runtime_error("score_graph: landmark prior out of range");
        const double w = g.lprior_prec[e];
        for (int k = 0; k < d; ++k) {
            const double tv = g.lprior_t[e * d + k];
            c0 += w * tv * tv;
        }
    }
    out.c0 = c0;
}

// score/solve_score.py:28-32 (_check_factor_graph): is every variable touched by a measurement or a prior?  (Indices out of
// range count as "not connected": the model construction reports them.)
inline bool graph_connected(const score_graph& g) {
    int64_t Np = 0;
    for (int c = 0; c < g.n_chains; ++c) Np += std::max(0, g.chain_len[c]);
    const int64_t Nv = Np + std::max(0, g.n_landmarks);
    std::vector<char> touched((size_t)Nv, 0);
    auto touch = [&](int64_t v) { if (v >= 0 && v < Nv) touched[(size_t)v] = 1; };
    for (int64_t e = 0; e < g.n_rel; ++e) { touch(g.rel_base[e]); touch(g.rel_to[e]); }
    for (int64_t r = 0; r < g.n_rng; ++r) { touch(g.rng_a[r]); touch(g.rng_b[r]); }
    for (int64_t e = 0; e < g.n_lprior; ++e) touch(Np + g.lprior_lm[e]);
    for (char tch : touched)
        if (!tch) return false;
    return true;
}

// What a handle made from factor graphs remembers of them for score_read_estimates: per problem its offsets and counts, and the
// range endpoints / measured distances (the QCQP directions of a program solved through the SOCP are r = D / max(|D|, dist)).
struct EstProb {
    int32_t xoff, Np, Nl, Nr, n_rep;
    int32_t pose_off, lm_off, rng_off;   // first pose / landmark / range of the problem in the concatenated outputs
};
struct EstLayout {
    int32_t d = 0, relaxation = 0;
    std::vector<EstProb> probs;
    std::vector<int32_t> rng_a, rng_b;   // concatenated, problem-local variable ids
    std::vector<double> rng_dist;
    int64_t n_pose = 0, n_lm = 0, n_rng = 0;
    bool dirs_always = false;            // a QCQP graph solved in head form: ranges are always the directions
    bool valid() const { return !probs.empty(); }
};
// relaxation_as >= 0: the columns are those of that relaxation's program whatever the graphs say (score_headform.hpp: a QCQP
// graph's head form has the SOCP program's columns)
inline void est_layout_from_graphs(const score_graph* graphs, int count, const std::vector<int64_t>& xoff, EstLayout& L, int relaxation_as = -1) {
    L = EstLayout();
    const int relax = relaxation_as >= 0 ? relaxation_as : graphs[0].relaxation;
    L.d = graphs[0].dim; L.relaxation = relax;
    L.dirs_always = relaxation_as == 0 && graphs[0].relaxation == 1;
    const int D1 = L.d + 1;
    for (int p = 0; p < count; ++p) {
        const score_graph& g = graphs[p];
        int64_t Np = 0;
        for (int c = 0; c < g.n_chains; ++c) Np += g.chain_len[c];
        EstProb e{};
        e.xoff = (int32_t)xoff[(size_t)p]; e.Np = (int32_t)Np; e.Nl = g.n_landmarks; e.Nr = (int32_t)g.n_rng;
        e.n_rep = (int32_t)((Np - 1) * D1 + g.n_landmarks + (relax == 0 ? 0 : g.n_rng));
        e.pose_off = (int32_t)L.n_pose; e.lm_off = (int32_t)L.n_lm; e.rng_off = (int32_t)L.n_rng;
        L.n_pose += Np; L.n_lm += g.n_landmarks; L.n_rng += g.n_rng;
        L.probs.push_back(e);
        L.rng_a.insert(L.rng_a.end(), g.rng_a, g.rng_a + g.n_rng);
        L.rng_b.insert(L.rng_b.end(), g.rng_b, g.rng_b + g.n_rng);
        L.rng_dist.insert(L.rng_dist.end(), g.rng_dist, g.rng_dist + g.n_rng);
    }
}
// The estimate of every problem of the handle in the reference's own shapes, from the solver-space solution x (unscaled):
// replaces VariableCollection.get_variable_values (/root/reference/score/utils/gurobi_utils.py:114-136) -- poses as homogeneous
// (d+1) x (d+1) matrices with the rotation block rounded onto SO(d) (score_round.hpp: the maximiser of tr(R'M), what
// round_to_special_orthogonal computes; degenerate[i] = 1 where it is not unique: identity then, the caller decides), the
// relaxation's own blocks [R | t], landmarks, range variables (qcqp_dirs on an SOCP program: the optimal QCQP directions).
// The host specification of k_read_estimates (and the CPU twin's path).
inline void read_estimates_host(const EstLayout& L, int qcqp_dirs, const double* x, double* poses, double* relaxed, double* lms,
                                double* rng, int32_t* degenerate) {
    const int d = L.d, D1 = d + 1;
    for (const EstProb& P : L.probs) {
        auto xv = [&](int64_t local) { return x[(size_t)(P.xoff + local)]; };
        for (int64_t lp = 0; lp < P.Np; ++lp) {
            const int64_t i = P.pose_off + lp;
            double blk[3][4], m[9], r[9];
            for (int k = 0; k < d; ++k)
                for (int j = 0; j < D1; ++j) blk[k][j] = lp == 0 ? (j == k ? 1.0 : 0.0) : xv((int64_t)k * P.n_rep + (lp - 1) * D1 + j);
            for (int k = 0; k < d; ++k)
                for (int j = 0; j < d; ++j) m[k * d + j] = blk[k][j];
            int32_t bad = 0;
            if (d == 2) round_so2(m, r, &bad); else round_so3(m, r, &bad);
            if (degenerate) degenerate[i] = bad;
            if (relaxed)
                for (int k = 0; k < d; ++k)
                    for (int j = 0; j < D1; ++j) relaxed[(size_t)i * d * D1 + k * D1 + j] = blk[k][j];
            if (poses) {
                double* T = poses + (size_t)i * D1 * D1;
                for (int k = 0; k < d; ++k) {
                    for (int j = 0; j < d; ++j) T[k * D1 + j] = r[k * d + j];
                    T[k * D1 + d] = blk[k][d];
                }
                for (int j = 0; j < d; ++j) T[d * D1 + j] = 0.0;
                T[d * D1 + d] = 1.0;
            }
        }
        const int64_t lm0 = (int64_t)(P.Np - 1) * D1;
        if (lms)
            for (int64_t l = 0; l < P.Nl; ++l)
                for (int k = 0; k < d; ++k) lms[(size_t)(P.lm_off + l) * d + k] = xv((int64_t)k * P.n_rep + lm0 + l);
        if (!rng) continue;
        auto tvar = [&](int64_t v, int k) {
            if (v < P.Np) return v == 0 ? 0.0 : xv((int64_t)k * P.n_rep + (v - 1) * D1 + d);
            return xv((int64_t)k * P.n_rep + lm0 + (v - P.Np));
        };
        for (int64_t r_ = 0; r_ < P.Nr; ++r_) {
            const int64_t i = P.rng_off + r_;
            if (L.relaxation != 0) {
                for (int k = 0; k < d; ++k) rng[(size_t)i * d + k] = xv((int64_t)k * P.n_rep + lm0 + P.Nl + r_);
            } else if (!qcqp_dirs) {
                rng[(size_t)i] = xv((int64_t)d * P.n_rep + r_);
            } else {
                double dl[3], nn = 0.0;
                for (int k = 0; k < d; ++k) { dl[k] = tvar(L.rng_a[(size_t)i], k) - tvar(L.rng_b[(size_t)i], k); nn += dl[k] * dl[k]; }
                const double den = std::max(std::sqrt(nn), L.rng_dist[(size_t)i]);
                // (a range measured as exactly 0: the cost w |t_i - t_j - 0 r|^2 does not depend on r -- every path returns r = 0 there,
                //  as headform_expand does for its idle cone)
                const bool idle = !(L.rng_dist[(size_t)i] > 0.0);
                for (int k = 0; k < d; ++k) rng[(size_t)i * d + k] = (den > 0.0 && !idle) ? dl[k] / den : 0.0;
            }
        }
    }
}

// `count` graphs, one graph per part of one parallel region (score_assemble_batch).  out[i] is filled for every i or an
// exception names the first graph that failed (the others' results are dropped by the caller).
inline void assemble_graphs(const score_graph* graphs, int count, AssembledQP* const* out) {
    BuildScope scope;
    std::vector<std::string> err((size_t)count);
    // (a graph takes 1-2 ms: parts of one graph each, dealt round-robin by an atomic counter so that a large graph
    //  does not hold up the part it shares with others)
    std::atomic<int> next{0};
    const int T = (int)std::min<int64_t>(region_width(), count);
    parallel_ranges(T, 1, [&](int, int64_t, int64_t) {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= count) break;
            try {
                assemble_graph(graphs[i], *out[i]);
            } catch (const std::exception& e) {
                err[(size_t)i] = e.what()[0] ? e.what() : "error";
            }
        }
    }, T);
    for (int i = 0; i < count; ++i)
        if (!err[(size_t)i].empty()) throw std::runtime_error("graph " + std::to_string(i) + ": " + err[(size_t)i]);
}

}  // namespace score
