// score_polish_host.hpp -- host-side setup of the semismooth-Newton polish.
//
// Structure exploited (detected from the conic data, no extra ABI): every cone is
// a second-order cone whose HEAD variable h is private -- it appears in exactly
// one row of A (its own head row, coefficient a < 0, b_head = 0) and only on the
// diagonal of P.  That is the SCORE "SOCP" relaxation (score/utils/gurobi_utils.py
// :289-294 distance variable, :345-352 cone, :486-487 cost).  Minimising over the
// head variables in closed form leaves the unconstrained, C^1, piecewise-quadratic
//
//     F(u) = 1/2 u'P u + q'u + sum_k 1/2 c_k max(0, |t_k(u)| - theta_k)^2 ,
//     t_k(u) = b_tail,k - A_tail,k u ,  c_k = P_hh / a^2 ,  theta_k = |a| (-q_h / P_hh)
//
// (everything in the equilibrated variables).  Its generalised Hessian is
// P + A_tail' B A_tail with one (dim-1)x(dim-1) block per ACTIVE cone, i.e. a
// sparse SPD matrix on a fixed superset pattern "H".  The Newton systems are
// solved by the same PCG + chain-preconditioner kernels as the ADMM KKT systems.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "score_host.hpp"
#include "score_band.hpp"

namespace score {

constexpr int kPolishMaxTail = 3;  // tail dimension d of SOC(d + 1), d in {2, 3}
// added to the diagonal of the non-head rows: keeps H positive definite along flat directions
// (landmarks all of whose cones are slack, gauge modes of robots no active cone ties down)
constexpr double kPolishDiagReg = 1e-9;
constexpr int kLongContrib = 64;  // entries of H with more contributions get a workgroup of their own (k_hassemble)

struct PolishData {
    bool available = false;
    int T = 0;  // tail dimension (uniform over the batch)
    // per cone (same order as HostSystem::cone_row)
    std::vector<int32_t> head_col;
    std::vector<double> a_abs, ck, theta, xstar;
    std::vector<int32_t> is_head;  // per column
    // Newton matrix pattern, P on that pattern, contribution lists
    Csr Hm;
    std::vector<double> Pon;
    std::vector<int32_t> cptr, ccone, cab;
    std::vector<double> ccoef;
    RowBlocks rbH;
    // chain / Jacobi positions in Hm.val
    std::vector<int32_t> pos_diag, pos_sub, diag_pos;
    BandLayout band;  // band view of Hm (score_band.hpp): every chain of every replica is a run of plain rows
    std::vector<int32_t> long_ent, long_prob;  // entries with more than kLongContrib contributions, and their problems
};

// Structure detection (see the head of this file): fills the per-cone data and Q.T; false when the program is not of that
// form.  cone_of_row (optional): the cone of every row of A.
inline bool polish_structure(const HostSystem& H, PolishData& Q, std::vector<int32_t>* cone_of_row_out = nullptr) {
    const int64_t n = H.n_tot, m = H.m_tot;
    const size_t ncones = H.cone_row.size();
    if (ncones == 0 || m == 0) return false;
    int T = H.cone_dim[0] - 1;
    if (T < 1 || T > kPolishMaxTail) return false;
    Q.head_col.assign(ncones, -1);
    Q.a_abs.assign(ncones, 0.0); Q.ck.assign(ncones, 0.0); Q.theta.assign(ncones, 0.0); Q.xstar.assign(ncones, 0.0);
    Q.is_head.assign(n, 0);
    std::vector<int32_t> cone_of_row(m, -1);
    for (size_t k = 0; k < ncones; ++k) {
        if (H.cone_type[k] != 1 || H.cone_dim[k] - 1 != T) return false;
        const int r0 = H.cone_row[k];
        for (int a = 0; a <= T; ++a) cone_of_row[r0 + a] = (int32_t)k;
        if (H.A.ptr[r0 + 1] - H.A.ptr[r0] != 1) return false;
        const int32_t h = H.A.col[H.A.ptr[r0]];
        const double a = H.A.val[H.A.ptr[r0]];
        if (!(a < 0.0) || H.b[r0] != 0.0) return false;
        // column h: a single entry in A (this one) and a positive diagonal-only row in P
        if (H.G2.ptr[h + 1] - H.g2_split[h] != 1) return false;
        if (H.g2_split[h] - H.G2.ptr[h] != 1 || H.G2.col[H.G2.ptr[h]] != h) return false;
        const double phh = H.G2.val[H.G2.ptr[h]];
        if (!(phh > 0.0)) return false;
        Q.head_col[k] = h;
        Q.is_head[h] = 1;
        Q.a_abs[k] = -a;
        Q.ck[k] = phh / (a * a);
        Q.xstar[k] = -H.q[h] / phh;
        Q.theta[k] = -a * Q.xstar[k];
    }
    Q.T = T;
    if (cone_of_row_out) *cone_of_row_out = std::move(cone_of_row);
    return true;
}

inline void build_polish(const HostSystem& H, PolishData& Q, bool verbose = false, bool band_view = false) {
    Q = PolishData();
    BuildScope scope;  // (built on a thread of its own next to the handle's uploads: shares the thread budget)
    PhaseTimer pt(verbose);
    const int64_t n = H.n_tot;
    std::vector<int32_t> cone_of_row;
    if (!polish_structure(H, Q, &cone_of_row)) return;
    const int T = Q.T;
    pt.mark("    polish: structure check");
    // ---- Newton matrix pattern + contribution lists ----
    struct Contrib { int32_t j, cone, ab; double coef; };
    struct Part {  // rows [i0, i1) of H, built by one host thread
        std::vector<int32_t> col, row_len, ent_len, ccone, cab;
        std::vector<double> pon, ccoef;
    };
    const int n_parts = parallel_parts(n, 8192);
    std::vector<Part> parts((size_t)n_parts);
    auto h_row_weight = [&](int64_t i) {  // contributions gathered for row i (sorted: long rows cost n log n)
        const double g = (double)(H.g2_split[i] - H.G2.ptr[i]) + 2.0 * T * (double)(H.G2.ptr[i + 1] - H.g2_split[i]);
        return g > 64.0 ? 4.0 * g : g;
    };
    parallel_ranges_balanced(n, 8192, h_row_weight, [&](int t, int64_t i0, int64_t i1) {
        Part W;  // thread-local, handed over at the end (neighbouring parts' vector headers share cache lines)
        std::vector<Contrib> rowc;
        {   // reserve (virtual) room for the worst case so that the buffers never re-allocate
            size_t ub_ent = 0, ub_con = 0;
            for (int64_t i = i0; i < i1; ++i) {
                ub_ent += (size_t)(H.g2_split[i] - H.G2.ptr[i]) + 1;
                ub_con += (size_t)(H.G2.ptr[i + 1] - H.g2_split[i]) * T * 2;
            }
            ub_ent += ub_con;
            W.col.reserve(ub_ent); W.pon.reserve(ub_ent); W.ent_len.reserve(ub_ent);
            W.ccone.reserve(ub_con); W.cab.reserve(ub_con); W.ccoef.reserve(ub_con);
            W.row_len.reserve(i1 - i0);
        }
        for (int64_t i = i0; i < i1; ++i) {
            rowc.clear();
            if (Q.is_head[i]) {  // decoupled: unit diagonal, the Newton step leaves it alone
                W.col.push_back((int32_t)i);
                W.pon.push_back(1.0);
                W.ent_len.push_back(0);
                W.row_len.push_back(1);
                continue;
            }
            for (int k = H.G2.ptr[i]; k < H.g2_split[i]; ++k)  // P row (cone = -1 marks a P entry)
                rowc.push_back(Contrib{H.G2.col[k], -1, 0, H.G2.val[k]});
            bool has_diag = false;
            for (const auto& c : rowc) has_diag = has_diag || (c.j == i);
            if (!has_diag) rowc.push_back(Contrib{(int32_t)i, -1, 0, 0.0});
            for (int t2 = H.g2_split[i]; t2 < H.G2.ptr[i + 1]; ++t2) {  // rows of A that contain column i
                const int r = H.G2.col[t2] - (int32_t)n;
                const double vi = H.G2.val[t2];
                const int cone = cone_of_row[r];
                const int r0 = H.cone_row[cone];
                const int a = r - r0 - 1;  // tail index of row r
                if (a < 0) continue;       // (head rows only hold the head column)
                for (int b = 0; b < T; ++b) {
                    const int rb = r0 + 1 + b;
                    for (int kk = H.A.ptr[rb]; kk < H.A.ptr[rb + 1]; ++kk)
                        rowc.push_back(Contrib{H.A.col[kk], cone, a * T + b, vi * H.A.val[kk]});
                }
            }
            // stable order by column: insertion sort for the usual short rows (std::stable_sort allocates its buffer per call)
            if (rowc.size() > 64) {
                std::stable_sort(rowc.begin(), rowc.end(), [](const Contrib& x, const Contrib& y) { return x.j < y.j; });
            } else {
                for (size_t x = 1; x < rowc.size(); ++x) {
                    const Contrib c = rowc[x];
                    size_t y = x;
                    while (y > 0 && rowc[y - 1].j > c.j) { rowc[y] = rowc[y - 1]; --y; }
                    rowc[y] = c;
                }
            }
            size_t e = 0;
            int32_t nent = 0;
            while (e < rowc.size()) {
                const int32_t j = rowc[e].j;
                double pval = 0.0;
                int32_t nc = 0;
                for (; e < rowc.size() && rowc[e].j == j; ++e) {
                    if (rowc[e].cone < 0) pval += rowc[e].coef;
                    else { W.ccone.push_back(rowc[e].cone); W.cab.push_back(rowc[e].ab); W.ccoef.push_back(rowc[e].coef); ++nc; }
                }
                W.col.push_back(j);
                W.pon.push_back(j == i ? pval + kPolishDiagReg : pval);
                W.ent_len.push_back(nc);
                ++nent;
            }
            W.row_len.push_back(nent);
        }
        parts[t] = std::move(W);
    }, n_parts);
    pt.mark("    polish: rows");
    Q.Hm.nrows = Q.Hm.ncols = n;
    Q.Hm.ptr.assign(1, 0);
    Q.cptr.assign(1, 0);
    std::vector<std::vector<int32_t>> longs(parts.size());  // entries with more than kLongContrib contributions, part by part
    {   // sizes, then every part is copied into place by its own thread
        std::vector<size_t> eoff(parts.size() + 1, 0), coff(parts.size() + 1, 0), roff(parts.size() + 1, 0);
        for (size_t k = 0; k < parts.size(); ++k) {
            eoff[k + 1] = eoff[k] + parts[k].col.size();
            coff[k + 1] = coff[k] + parts[k].ccone.size();
            roff[k + 1] = roff[k] + parts[k].row_len.size();
        }
        Q.Hm.col.resize(eoff.back()); Q.Pon.resize(eoff.back());
        Q.ccone.resize(coff.back()); Q.cab.resize(coff.back()); Q.ccoef.resize(coff.back());
        Q.Hm.ptr.resize(n + 1); Q.cptr.resize(eoff.back() + 1);
        parallel_ranges((int64_t)parts.size(), 1, [&](int, int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                const Part& W = parts[k];
                for (size_t e = 0; e < W.ent_len.size(); ++e)
                    if (W.ent_len[e] > kLongContrib) longs[(size_t)k].push_back((int32_t)(eoff[k] + e));
                std::copy(W.col.begin(), W.col.end(), Q.Hm.col.begin() + eoff[k]);
                std::copy(W.pon.begin(), W.pon.end(), Q.Pon.begin() + eoff[k]);
                std::copy(W.ccone.begin(), W.ccone.end(), Q.ccone.begin() + coff[k]);
                std::copy(W.cab.begin(), W.cab.end(), Q.cab.begin() + coff[k]);
                std::copy(W.ccoef.begin(), W.ccoef.end(), Q.ccoef.begin() + coff[k]);
                int32_t acc = (int32_t)eoff[k];
                for (size_t r = 0; r < W.row_len.size(); ++r) { acc += W.row_len[r]; Q.Hm.ptr[roff[k] + r + 1] = acc; }
                int32_t cacc = (int32_t)coff[k];
                for (size_t e = 0; e < W.ent_len.size(); ++e) { cacc += W.ent_len[e]; Q.cptr[eoff[k] + e + 1] = cacc; }
            }
        });
    }
    {   // the long entries in order, with the problem each belongs to
        int pr = 0;
        for (const auto& lp : longs)
            for (int32_t e : lp) {
                while (pr + 1 < H.count && (int64_t)e >= (int64_t)Q.Hm.ptr[(size_t)H.xoff[pr + 1]]) ++pr;
                Q.long_ent.push_back(e);
                Q.long_prob.push_back(pr);
            }
    }
    pt.mark("    polish: concatenate");
    Q.rbH = make_rowblocks(Q.Hm, plain_segments(H.xoff), H.count);
    pt.mark("    polish: row blocks");
    // ---- chain block / Jacobi positions in H ----
    const int bs = H.bs;
    const int b2 = bs * bs;
    Q.pos_diag.assign(H.node_col.size() * b2, -1);
    Q.pos_sub.assign(H.node_col.size() * b2, -1);
    std::vector<int32_t> prev_col(H.node_col.size(), -1);  // column of the chain predecessor
    for (const auto& ch : H.chains)
        for (int i = 1; i < ch.N; ++i) prev_col[ch.node_begin + i] = H.node_col[ch.node_begin + i - 1];
    parallel_ranges((int64_t)H.node_col.size(), 2048, [&](int, int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g) {
            const int32_t col = H.node_col[g];
            size_t o = (size_t)g * b2;
            for (int a = 0; a < bs; ++a)
                for (int b = 0; b < bs; ++b, ++o) {
                    Q.pos_diag[o] = find_in_row(Q.Hm, col + a, col + b);
                    if (prev_col[g] >= 0) Q.pos_sub[o] = find_in_row(Q.Hm, col + a, prev_col[g] + b);
                }
        }
    });
    Q.diag_pos.clear();
    for (int32_t c : H.diag_cols) Q.diag_pos.push_back(find_in_row(Q.Hm, c, c));
    pt.mark("    polish: positions");
    if (!H.chainsH.empty() && band_view) {
        const std::vector<char> all(H.chainsH.size(), 1);
        Q.band = build_band_layout(Q.Hm, plain_segments(H.xoff), band_runs(H.chainsH, all, bs, 1, H.rep_n, false), bs, H.count);
        pt.mark("    polish: band view");
    }
    Q.available = true;
}

}  // namespace score
