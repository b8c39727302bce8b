// ORACLE / TEST INFRASTRUCTURE -- CPU twin of the HIP solver (libscore_cpu.so).
//
// Same C ABI (include/score_hip.h), same host-side setup and ADMM driver
// (score_amd/csrc/score_host.hpp, score_driver.hpp), but every device kernel
// is restated as a plain loop.  It exists so that (1) the algorithm, the
// equilibration, the KKT assembly and the multi-level chain factorisation can
// be tested in a container without a GPU, (2) the GPU kernels can be compared
// vector-by-vector against an independent execution of the same iteration,
// and (3) bench.py has a CPU baseline ("port") to time beside the GPU.
// The product path (score_amd/) never loads this library: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// The reference has no counterpart: its solve is Gurobi's barrier method
// (score/solve_score.py:76), which is closed source and absent here.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <mutex>
#endif

#include "../../score_amd/csrc/score_driver.hpp"
#include "../../score_amd/csrc/score_assemble.hpp"
#include "../../score_amd/csrc/score_gn.hpp"
#include "../../score_amd/csrc/score_round.hpp"
#include "../../score_amd/csrc/score_generate.hpp"
#include "../../score_amd/csrc/score_link.hpp"  // (host part only: which loop closures the product's Newton preconditioner would carry -- the plan, for the CPU tests)
#include "../../score_amd/csrc/score_polish_host.hpp"  // (layout checks only: band_check_h; the twin has no polish)

namespace {

using namespace score;

thread_local std::string g_err;

struct CpuBackend {
    static constexpr bool kFactorOnHost = true;
    // By default the twin applies the full K the problem defines, whatever its row-replication hint says (and so checks
    // the product's replicated kernels against an execution that knows nothing of the structure).  With
    // SCORE_TWIN_REPLICATION set it runs the replicated host structures instead (K and G1 hold replica 0's rows, a
    // replica's chain uses its owner's factors): the CPU-side test of that setup code.
    static bool allow_rep() { return std::getenv("SCORE_TWIN_REPLICATION") != nullptr; }
    RuizOffload* ruiz_offload(const score_settings&) { return nullptr; }  // the twin equilibrates with the host loop
    // y[row (+ q rs)] = fn(M row . v[col + q rs_in]) over the rows M holds for problem pi
    template <class F>
    void for_rows(const Csr& M, int pi, int rs_in_fixed, F&& fn) const {
        const HostSystem& h = *H;
        const int64_t x0 = h.xoff[pi], x1 = h.xoff[pi + 1];
        if (h.rep <= 1) {
#pragma omp parallel for schedule(static)
            for (int64_t i = x0; i < x1; ++i) fn(i, i, 0);
            return;
        }
        const int64_t nr = h.rep_n[(size_t)pi];
#pragma omp parallel for schedule(static)
        for (int64_t i = x0; i < x0 + nr; ++i)
            for (int q = 0; q < h.rep; ++q) fn(i, i + q * nr, (int64_t)q * (rs_in_fixed ? rs_in_fixed : nr));
#pragma omp parallel for schedule(static)
        for (int64_t i = x0 + (int64_t)h.rep * nr; i < x1; ++i) fn(i, i, 0);
    }
    const HostSystem* H = nullptr;
    score_settings st{};
    std::vector<double> xtu, xy, s, r, z, p, w, kx;  // xtu = [xt | u], xy = [x | y], kx = K xt
    std::vector<double> scr;
    std::vector<int> done;
    std::vector<double> cg_red;  // last measured sqrt(r'z_final / r'z_init) per problem
    int cg_iters = 2;

    void set_cg_iters(int k) { cg_iters = k; }
    bool polish(const HostSystem&, const score_settings&, const std::vector<int>&, int*, int*, const std::vector<double>*) { return false; }  // HIP backend only
    bool polish_available() const { return false; }
    void set_newton_limit(int) {}
    void cg_reduction(std::vector<double>& out) { out = cg_red; }
    std::vector<double> prec_part, dot_part;  // partial sums, added in a fixed order (the same bits whatever the thread team)

    bool device_setup_ok(const HostSystem&, const score_problem*, const score_settings&) const { return false; }  // (the twin IS the host setup)
    bool device_setup_ok_graphs(const HostSystem&, const score_graph*, const score_settings&) const { return false; }
    int64_t L_pairs() const { return link_plan.items.empty() ? 0 : (int64_t)link_plan.pair_cols.size(); }
    LinkPlan link_plan;  // (the product's plan of loop closures inside its Newton preconditioner, score_link.hpp: host logic only here)
    void init(const HostSystem& h, const score_settings& s_, const score_problem* probs = nullptr, const score_graph* graphs = nullptr) {
        H = &h;
        st = s_;
        {
            std::vector<int32_t> pairs;
            if (graphs) find_link_pairs_graphs(h, graphs, pairs);
            else if (probs) find_link_pairs_P(h, probs, pairs);
            make_link_plan(h, pairs, link_plan);
        }
        xtu.assign(h.n_tot + h.m_tot, 0.0);
        xy.assign(h.n_tot + h.m_tot, 0.0);
        s.assign(h.m_tot, 0.0);
        r.assign(h.n_tot, 0.0);
        z.assign(h.n_tot, 0.0);
        p.assign(h.n_tot, 0.0);
        w.assign(h.n_tot, 0.0);
        kx.assign(h.n_tot, 0.0);
        scr.assign((size_t)std::max<int64_t>(1, h.scratch_nodes) * std::max(1, h.bs), 0.0);
        done.assign(h.count, 0);
        cg_red.assign(h.count, 0.0);
        cg_iters = st.cg_iters;
        reset();
    }
    // ---- the loop closures inside the preconditioner (score_link.hpp), as host loops: the specification of the HIP backend's
    //      link kernels ON THE ADMM SET (K).  z = y - Z t, y = T^-1 r (the chains), Z = T^-1 U, t = (I + G Z_U)^-1 G y_U, group by group.
    std::vector<double> link_Zr;                 // rounds x n_tot
    std::vector<std::vector<double>> link_Q;     // per group: n_u x n_u, row-major
    bool link_on = false;
    void chains_only(int pi) {  // z = T^-1 r on the chains of problem pi (Jacobi columns: r * dinv)
        double unused = 0.0;
        const bool keep = link_on;
        link_on = false;
        precond(pi, unused);
        link_on = keep;
    }
    void link_refresh() {
        const HostSystem& h = *H;
        const LinkPlan& L = link_plan;
        link_on = false;
        if (L.empty() || std::getenv("SCORE_NO_LINKS") != nullptr) return;
        const size_t n = (size_t)h.n_tot;
        link_Zr.assign((size_t)L.rounds * n, 0.0);
        std::vector<double> keep_r = r, keep_z = z;
        for (int rd = 0; rd < L.rounds; ++rd) {
            std::fill(r.begin(), r.end(), 0.0);
            for (size_t u = 0; u < L.ucol.size(); ++u)
                if (L.uround[u] == rd) r[(size_t)L.ucol[u]] = 1.0;
            for (int pi = 0; pi < h.count; ++pi) chains_only(pi);
            std::copy(z.begin(), z.begin() + (std::ptrdiff_t)n, link_Zr.begin() + (std::ptrdiff_t)((size_t)rd * n));
        }
        r = keep_r; z = keep_z;
        link_Q.assign(L.probs.size(), {});
        for (size_t g = 0; g < L.probs.size(); ++g) {
            const LinkProb& P = L.probs[g];
            const int m = P.n_u;
            std::vector<double> G((size_t)m * m, 0.0), S((size_t)m * m, 0.0);
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) {
                    if (!L.mask[(size_t)P.q_off + (size_t)a * m + b]) continue;
                    int32_t sa = 0, sb = 0;
                    const int32_t ra = link_owner_col(h, P.prob, L.ucol[(size_t)P.u_begin + a], &sa), cb = link_owner_col(h, P.prob, L.ucol[(size_t)P.u_begin + b], &sb);
                    if (sa != sb) continue;  // (a loop closure couples row k with row k: the same replica)
                    const int pos = find_in_row(h.K, ra, cb);
                    if (pos >= 0) G[(size_t)a * m + b] = h.K.val[(size_t)pos];
                }
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) {
                    double acc = a == b ? 1.0 : 0.0;
                    for (int c = 0; c < m; ++c)
                        if (G[(size_t)a * m + c] != 0.0 && L.usuper[(size_t)P.u_begin + c] == L.usuper[(size_t)P.u_begin + b])
                            acc += G[(size_t)a * m + c] * link_Zr[(size_t)L.uround[(size_t)P.u_begin + b] * n + (size_t)L.ucol[(size_t)P.u_begin + c]];
                    S[(size_t)a * m + b] = acc;
                }
            // Q = S^-1 G: Gaussian elimination with partial pivoting on [S | G]
            bool singular = false;
            for (int k = 0; k < m && !singular; ++k) {
                int piv = k;
                for (int i = k + 1; i < m; ++i) if (std::fabs(S[(size_t)i * m + k]) > std::fabs(S[(size_t)piv * m + k])) piv = i;
                if (!(std::fabs(S[(size_t)piv * m + k]) > 1e-300)) { singular = true; break; }
                if (piv != k)
                    for (int c = 0; c < m; ++c) { std::swap(S[(size_t)k * m + c], S[(size_t)piv * m + c]); std::swap(G[(size_t)k * m + c], G[(size_t)piv * m + c]); }
                const double inv = 1.0 / S[(size_t)k * m + k];
                for (int c = 0; c < m; ++c) { S[(size_t)k * m + c] *= inv; G[(size_t)k * m + c] *= inv; }
                for (int i = 0; i < m; ++i) {
                    if (i == k) continue;
                    const double f = S[(size_t)i * m + k];
                    if (f == 0.0) continue;
                    for (int c = 0; c < m; ++c) { S[(size_t)i * m + c] -= f * S[(size_t)k * m + c]; G[(size_t)i * m + c] -= f * G[(size_t)k * m + c]; }
                }
            }
            if (singular) std::fill(G.begin(), G.end(), 0.0);
            link_Q[g] = std::move(G);
        }
        link_on = true;
    }
    // the correction of problem pi after its chains were applied; returns the new r'z of the problem
    double link_correct(int pi) {
        const HostSystem& h = *H;
        const LinkPlan& L = link_plan;
        const size_t n = (size_t)h.n_tot;
        for (size_t g = 0; g < L.probs.size(); ++g) {
            const LinkProb& P = L.probs[g];
            if (P.prob != pi) continue;
            const int m = P.n_u;
            std::vector<double> v((size_t)m), t((size_t)m, 0.0);
            for (int a = 0; a < m; ++a) v[(size_t)a] = z[(size_t)L.ucol[(size_t)P.u_begin + a]];
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) t[(size_t)a] += link_Q[g][(size_t)a * m + b] * v[(size_t)b];
            for (int i = 0; i < P.item_count; ++i) {
                const LinkItem& it = L.items[(size_t)P.item_begin + i];
                const ChainDesc& ch = h.chains[(size_t)it.chain];
                for (int node = 0; node < ch.N; ++node)
                    for (int c = 0; c < h.bs; ++c) {
                        const size_t col = (size_t)h.node_col[(size_t)ch.node_begin + node] + c;
                        double zz = z[col];
                        for (int rd = 0; rd < L.rounds; ++rd)
                            if (it.u[rd] >= 0) zz -= link_Zr[(size_t)rd * n + col] * t[(size_t)(it.u[rd] - P.u_begin)];
                        z[col] = zz;
                    }
            }
        }
        double acc = 0.0;
        for (int64_t i = h.xoff[pi]; i < h.xoff[pi + 1]; ++i) acc += r[(size_t)i] * z[(size_t)i];
        return acc;
    }
    bool has_links(int pi) const {
        for (const LinkProb& P : link_plan.probs) if (P.prob == pi) return true;
        return false;
    }

    void upload_rho(const HostSystem& h) {
        link_refresh();  // (the chain factors have just changed: refresh_rho)
        // K changed: the carried product kx = K xt is recomputed once
        for (int pi = 0; pi < h.count; ++pi)
            for_rows(h.K, pi, 0, [&](int64_t row, int64_t o, int64_t sh) { kx[o] = row_dot(h.K, row, xtu.data() + sh); });
        // u = rho (b - s) - y depends on rho
        for (int pi = 0; pi < h.count; ++pi)
            for (int64_t i = h.roff[pi]; i < h.roff[pi + 1]; ++i)
                xtu[h.n_tot + i] = h.rho[pi] * (h.b[i] - s[i]) - xy[h.n_tot + i];
    }
    void set_done(const std::vector<int>& d) { done = d; }
    void reset() {
        std::fill(xtu.begin(), xtu.end(), 0.0);
        std::fill(xy.begin(), xy.end(), 0.0);
        std::fill(s.begin(), s.end(), 0.0);
        std::fill(r.begin(), r.end(), 0.0);
        std::fill(z.begin(), z.end(), 0.0);
        std::fill(p.begin(), p.end(), 0.0);
        std::fill(w.begin(), w.end(), 0.0);
        std::fill(kx.begin(), kx.end(), 0.0);
        upload_rho(*H);
    }

    static double row_dot(const Csr& M, int64_t i, const double* v) {
        double acc = 0;
        for (int k = M.ptr[i]; k < M.ptr[i + 1]; ++k) acc += M.val[k] * v[M.col[k]];
        return acc;
    }

    void precond(int pi, double& rz) {  // z = M^{-1} r for problem pi, returns r'z
        const HostSystem& h = *H;
        const int w0 = h.prec_part_ptr[pi], w1 = h.prec_part_ptr[pi + 1];
        // (a partial sum per work item, added up in the items' order -- what the chain kernel's workgroups and the reduction
        //  behind them do: the same bits whatever the team of threads; an OpenMP reduction adds in the order the threads finish)
        prec_part.assign((size_t)(w1 - w0), 0.0);
#pragma omp parallel for schedule(dynamic, 1)
        for (int wi = w0; wi < w1; ++wi) {
            double acc = 0;
            const PrecWork& pw = h.prec_work[wi];
            if (pw.kind == 0) {
                const ChainDesc& ch = h.chains[pw.index];
                chain_solve_host(ch, h.levels.data(), h.fac.data(), h.node_col.data(), h.bs, r.data(), z.data(),
                                 scr.data() + (size_t)ch.scratch_off * h.bs);
                for (int i = 0; i < ch.N; ++i)
                    for (int c = 0; c < h.bs; ++c) {
                        const int col = h.node_col[ch.node_begin + i] + c;
                        acc += r[col] * z[col];
                    }
            } else {
                for (int e = pw.index; e < pw.index + pw.count; ++e) {
                    const int col = h.diag_cols[e];
                    z[col] = r[col] * h.dinv[e];
                    acc += r[col] * z[col];
                }
            }
            prec_part[(size_t)(wi - w0)] = acc;
        }
        double acc = 0;
        for (double v : prec_part) acc += v;
        rz = acc;
        if (link_on && has_links(pi)) rz = link_correct(pi);
    }

    // linear mode: the same PCG as the loop above (precond() + row_dot), run to r'M^-1 r <= tol^2 r0'M^-1 r0
    bool linear_solve(const HostSystem& h, const double* rhs, double* x, double rel_tol, int max_iters, int* used_out) {
        const int64_t n = h.n_tot;
        link_refresh();  // (the driver has just refactored the chains for the values of this solve)
        std::vector<double> xs((size_t)n, 0.0);
        for (int64_t i = 0; i < n; ++i) r[i] = rhs[i];
        double rz = 0.0;
        precond(0, rz);
        for (int64_t i = 0; i < n; ++i) p[i] = z[i];
        const double ref = rz * rel_tol * rel_tol;
        int used = 0;
        double beta_prev = 0.0;
        bool ok = !(rz > 0.0);
        while (!ok) {
            if (used > 0 && !(rz > ref)) { ok = true; break; }  // the test the device gate makes before a STEP
            if (used >= max_iters) break;
            double pw = 0.0;
            // (every 32nd product directly, as the device does: the recurrence K z + beta w_old drifts over a long solve)
            const bool direct = (used == 0) || (used % 32 == 0);
            for (int64_t i = 0; i < n; ++i) {
                w[i] = direct ? row_dot(h.K, i, p.data()) : row_dot(h.K, i, z.data()) + beta_prev * w[i];
                pw += p[i] * w[i];
            }
            const double a = pw > 0.0 ? rz / pw : 0.0;
            for (int64_t i = 0; i < n; ++i) { xs[i] += a * p[i]; r[i] -= a * w[i]; }
            ++used;
            double rz_new = 0.0;
            precond(0, rz_new);
            const double beta = rz > 0.0 ? rz_new / rz : 0.0;
            for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
            beta_prev = beta;
            rz = rz_new;
        }
        for (int64_t i = 0; i < n; ++i) x[i] = xs[i];
        if (used_out) *used_out = used;
        return ok;
    }

    void iterate_problem(int pi, bool measure) {
        const HostSystem& h = *H;
        const int64_t x0 = h.xoff[pi], x1 = h.xoff[pi + 1];
        const double rho = h.rho[pi], sigma = h.sigma, al = st.alpha;
        double* xt = xtu.data();
        double* x = xy.data();
        double* u = xtu.data() + h.n_tot;
        double* y = xy.data() + h.n_tot;
        // r = sigma x - q + A'u - K xt   (K xt carried incrementally in kx)
        for_rows(h.G1, pi, 1, [&](int64_t row, int64_t o, int64_t sh) {
            r[o] = sigma * x[o] - h.q[o] + row_dot(h.G1, row, xtu.data() + sh) - kx[o];
        });
        double rz = 0;
        precond(pi, rz);
        const double rz_init = rz;
        for (int64_t i = x0; i < x1; ++i) p[i] = z[i];
        double beta_prev = 0.0;
        for (int j = 1; j <= cg_iters; ++j) {
            double pw = 0;
            // first product of a solve: w = K p.  Later ones: K (z + beta p_old) = K z + beta w_old (what the KPB
            // kernel computes: one gather per nonzero)
            for_rows(h.K, pi, 0, [&](int64_t row, int64_t o, int64_t sh) {
                w[o] = (j == 1) ? row_dot(h.K, row, p.data() + sh) : row_dot(h.K, row, z.data() + sh) + beta_prev * w[o];
            });
            {   // p'w in pieces of 1024 entries, the pieces added in order (see precond)
                const int64_t nch = (x1 - x0 + 1023) / 1024;
                dot_part.assign((size_t)nch, 0.0);
#pragma omp parallel for schedule(static)
                for (int64_t c = 0; c < nch; ++c) {
                    double acc = 0;
                    for (int64_t i = x0 + c * 1024; i < std::min(x1, x0 + (c + 1) * 1024); ++i) acc += p[i] * w[i];
                    dot_part[(size_t)c] = acc;
                }
                for (double v : dot_part) pw += v;
            }
            const double a = pw > 0 ? rz / pw : 0.0;
#pragma omp parallel for schedule(static)
            for (int64_t i = x0; i < x1; ++i) {
                xt[i] += a * p[i];
                kx[i] += a * w[i];
                r[i] -= a * w[i];
            }
            if (j == cg_iters && measure) {
                double rzf = 0;
                precond(pi, rzf);
                cg_red[pi] = rz_init > 0 ? std::sqrt(std::max(0.0, rzf) / rz_init) : 0.0;
            }
            if (j < cg_iters) {
                double rz2 = 0;
                precond(pi, rz2);
                const double beta = rz > 0 ? rz2 / rz : 0.0;
#pragma omp parallel for schedule(static)
                for (int64_t i = x0; i < x1; ++i) p[i] = z[i] + beta * p[i];
                beta_prev = beta;
                rz = rz2;
            }
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = x0; i < x1; ++i) x[i] = al * xt[i] + (1.0 - al) * x[i];
        // cones
        const int cb0 = h.cone_block_first[h.cone_part_ptr[pi]];
        const int cb1 = h.cone_block_first[h.cone_part_ptr[pi + 1]];
#pragma omp parallel for schedule(static)
        for (int c = cb0; c < cb1; ++c) {
            const int row = h.cone_row[c], dim = h.cone_dim[c];
            double nz2 = 0, t0 = 0;
            for (int k = 0; k < dim; ++k) {
                const int i = row + k;
                const double t = row_dot(h.A, i, xt);
                const double v = al * (h.b[i] - t) + (1.0 - al) * s[i];
                const double wv = v - y[i] / rho;
                // stash v in u and wv in s until the projection is known
                u[i] = v;
                s[i] = wv;
                if (k == 0) t0 = wv; else nz2 += wv * wv;
            }
            if (h.cone_type[c] == 0) {
                for (int k = 0; k < dim; ++k) s[row + k] = 0.0;
            } else {
                const double nz = std::sqrt(nz2);
                double sc_head, sc_tail;  // s+ = (head, sc_tail * tail)
                if (nz <= t0) { sc_head = t0; sc_tail = 1.0; }
                else if (nz <= -t0) { sc_head = 0.0; sc_tail = 0.0; }
                else { const double a = 0.5 * (t0 + nz); sc_head = a; sc_tail = a / nz; }
                s[row] = sc_head;
                for (int k = 1; k < dim; ++k) s[row + k] *= sc_tail;
            }
            for (int k = 0; k < dim; ++k) {
                const int i = row + k;
                const double v = u[i];
                y[i] += rho * (s[i] - v);
                u[i] = rho * (h.b[i] - s[i]) - y[i];
            }
        }
    }

    void run(int iters) {
        for (int it = 0; it < iters; ++it)
            for (int pi = 0; pi < H->count; ++pi)
                if (!done[pi]) iterate_problem(pi, it == iters - 1);
    }

    void residuals(std::vector<ResidualSums>& R) {
        const HostSystem& h = *H;
        const double* x = xy.data();
        const double* y = xy.data() + h.n_tot;
        for (int pi = 0; pi < h.count; ++pi) {
            ResidualSums a;
            for (int64_t i = h.roff[pi]; i < h.roff[pi + 1]; ++i) {
                const double t = row_dot(h.A, i, x);
                const double pr = t + s[i] - h.b[i];
                const double ie = 1.0 / h.E[i];
                a.rp_u = std::max(a.rp_u, std::fabs(pr) * ie);
                a.ax_u = std::max(a.ax_u, std::fabs(t) * ie);
                a.s_u = std::max(a.s_u, std::fabs(s[i]) * ie);
                a.rp_s = std::max(a.rp_s, std::fabs(pr));
                a.ax_s = std::max(a.ax_s, std::fabs(t));
                a.s_s = std::max(a.s_s, std::fabs(s[i]));
                a.by += h.b[i] * y[i];
                a.sy_yrp += y[i] * (s[i] - pr);
                if (pr != pr) a.rp_u = pr;
            }
            for (int64_t i = h.xoff[pi]; i < h.xoff[pi + 1]; ++i) {
                double px = 0, aty = 0;
                for (int k = h.G2.ptr[i]; k < h.g2_split[i]; ++k) px += h.G2.val[k] * xy[h.G2.col[k]];
                for (int k = h.g2_split[i]; k < h.G2.ptr[i + 1]; ++k) aty += h.G2.val[k] * xy[h.G2.col[k]];
                const double dr = px + h.q[i] + aty;
                const double id = 1.0 / h.D[i];
                a.rd_u = std::max(a.rd_u, std::fabs(dr) * id);
                a.px_u = std::max(a.px_u, std::fabs(px) * id);
                a.aty_u = std::max(a.aty_u, std::fabs(aty) * id);
                a.rd_s = std::max(a.rd_s, std::fabs(dr));
                a.px_s = std::max(a.px_s, std::fabs(px));
                a.aty_s = std::max(a.aty_s, std::fabs(aty));
                a.xPx += x[i] * px;
                a.qx += h.q[i] * x[i];
                a.xrd += x[i] * dr;
                if (dr != dr) a.rd_u = dr;
            }
            R[pi] = a;
        }
    }

    void download(const HostSystem& h, double* x, double* y, double* s_out) {
        if (x) for (int64_t i = 0; i < h.n_tot; ++i) x[i] = xy[i] * h.D[i];
        if (y) for (int64_t i = 0; i < h.m_tot; ++i) y[i] = xy[h.n_tot + i] * h.E[i];
        if (s_out) for (int64_t i = 0; i < h.m_tot; ++i) s_out[i] = s[i] / h.E[i];
    }

    int64_t get_vec(const char* name, double* out, int64_t len) {
        const HostSystem& h = *H;
        const double* src = nullptr;
        int64_t sz = 0;
        std::string nm(name);
        if (nm == "xt") { src = xtu.data(); sz = h.n_tot; }
        else if (nm == "u") { src = xtu.data() + h.n_tot; sz = h.m_tot; }
        else if (nm == "x") { src = xy.data(); sz = h.n_tot; }
        else if (nm == "y") { src = xy.data() + h.n_tot; sz = h.m_tot; }
        else if (nm == "s") { src = s.data(); sz = h.m_tot; }
        else if (nm == "r") { src = r.data(); sz = h.n_tot; }
        else if (nm == "z") { src = z.data(); sz = h.n_tot; }
        else if (nm == "p") { src = p.data(); sz = h.n_tot; }
        else if (nm == "w") { src = w.data(); sz = h.n_tot; }
        else if (nm == "kx") { src = kx.data(); sz = h.n_tot; }
        else if (nm == "D") { src = h.D.data(); sz = h.n_tot; }
        else if (nm == "E") { src = h.E.data(); sz = h.m_tot; }
        else if (nm == "Kval") { src = h.K.val.data(); sz = (int64_t)h.K.val.size(); }
        else if (nm == "rep") {
            const double v[3] = {(double)h.rep, (double)h.K.col.size(), (double)h.G1.col.size()};
            if (out && len > 0) std::memcpy(out, v, sizeof(double) * (size_t)std::min<int64_t>(len, 3));
            return 3;
        }
        else if (nm == "links") {  // the product's link plan for this handle (score_link.hpp::make_link_plan), as the HIP backend reports it + groups
            const LinkPlan& L = link_plan;
            int max_u = 0;
            for (const LinkProb& P : L.probs) max_u = std::max(max_u, (int)P.n_u);
            const double v[9] = {(double)L.pairs_total, (double)L.pairs_used, (double)L.ucol.size(), (double)L.items.size(), (double)L.rounds, 0.0,
                                 (double)L.probs.size(), (double)max_u, (double)L.max_items};
            if (out && len > 0) std::memcpy(out, v, sizeof(double) * (size_t)std::min<int64_t>(len, 9));
            return 9;
        }
        else if (nm == "link_pairs") {
            const int64_t np_ = L_pairs();
            if (out) for (int64_t i = 0; i < np_ && i < len; ++i) out[i] = (double)link_plan.pair_cols[(size_t)i];
            return np_;
        }
        else if (nm == "band_check" || nm == "band_check_h") {
            // Layout check of the product's band view (score_band.hpp) on the host: build the view of K (as the
            // backend would: owner chains, stored row segments) or of the Newton matrix pattern, fill V through dst,
            // apply it with band_apply_host and compare with the CSR rows.
            // -> [max |difference|, on, band tiles, csr tiles, diag tiles, slots, bytes of problem 0, source nnz, V size, layout build ms]
            double res[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const bool newton = nm == "band_check_h";
            PolishData Q;
            std::vector<double> hval;
            const Csr* M = &h.K;
            const double* mval = h.K.val.data();
            BandLayout L;
            int nr_rhs = h.rep;
            if (newton) {
                build_polish(h, Q, false, true);
                if (!Q.available) return -1;
                M = &Q.Hm;
                hval.resize(Q.Hm.col.size());
                for (size_t k = 0; k < hval.size(); ++k) hval[k] = Q.Pon[k] + 1e-3 * std::sin((double)k);  // (any values on the pattern)
                mval = hval.data();
                L = std::move(Q.band);
                nr_rhs = 1;
            } else {
                std::vector<char> use(h.chains.size());
                for (size_t ci = 0; ci < h.chains.size(); ++ci) use[ci] = h.chain_owner[ci] == (int32_t)ci;
                std::vector<RowSegment> sg;
                if (h.rep > 1) {
                    for (int p = 0; p < h.count; ++p) {
                        const int64_t nr = h.rep_n[(size_t)p];
                        sg.push_back(RowSegment{h.xoff[p], h.xoff[p] + nr, p, (int32_t)nr});
                        sg.push_back(RowSegment{h.xoff[p] + (int64_t)h.rep * nr, h.xoff[p + 1], p, 0});
                    }
                } else {
                    sg = plain_segments(h.xoff);
                }
                const double tb0 = now_ms();
                L = build_band_layout(h.K, sg, band_runs(h.chains, use, h.bs, h.rep, h.rep_n, true), h.bs, h.count);
                res[9] = now_ms() - tb0;
            }
            res[1] = L.on ? 1.0 : 0.0;
            if (L.on) {
                std::vector<double> V((size_t)L.v_size, 0.0), xv((size_t)h.n_tot), y0((size_t)h.n_tot, 0.0), y1((size_t)h.n_tot, 0.0);
                for (size_t k = 0; k < L.dst.size(); ++k)
                    if (L.dst[k] >= 0) V[(size_t)L.dst[k]] = mval[k];
                for (int64_t i = 0; i < h.n_tot; ++i) xv[(size_t)i] = std::cos(0.37 * (double)i) + 1e-3 * (double)(i % 11);
                band_apply_host(L, *M, V.data(), mval, nr_rhs, xv.data(), y1.data());
                // reference: every stored row applied to its right-hand sides (a split long row counts with its first segment)
                std::vector<char> covered((size_t)h.n_tot, 0);
                double worst = 0.0;
                for (int b = 0; b < L.nb(); ++b) {
                    const int rs = L.rs[(size_t)b], nrep = rs > 0 ? nr_rhs : 1;
                    const int32_t* m = &L.meta[4 * (size_t)b];
                    const int kind = (int)((uint32_t)L.meta2[4 * (size_t)b + 3] >> 28);
                    if (kind == BAND_KIND_CSR && m[1] - m[0] == 1 && m[2] != M->ptr[m[0]]) continue;  // a later segment
                    for (int row = m[0]; row < m[1]; ++row)
                        for (int q = 0; q < nrep; ++q) {
                            double acc = 0.0;
                            for (int k = M->ptr[row]; k < M->ptr[row + 1]; ++k) acc += mval[k] * xv[(size_t)M->col[k] + (size_t)q * rs];
                            worst = std::max(worst, std::fabs(acc - y1[(size_t)row + (size_t)q * rs]) / (1.0 + std::fabs(acc)));
                            covered[(size_t)row + (size_t)q * rs] += 1;
                        }
                }
                for (int64_t i = 0; i < h.n_tot; ++i)
                    if (covered[(size_t)i] != 1) worst = std::max(worst, 1e30);  // every unknown's row exactly once
                res[0] = worst;
                res[2] = L.n_band; res[3] = L.n_csr; res[4] = L.n_diag; res[5] = L.S; res[6] = L.bytes.empty() ? 0.0 : L.bytes[0];
                res[7] = (double)M->col.size(); res[8] = (double)L.v_size;
            }
            if (out && len > 0) std::memcpy(out, res, sizeof(double) * (size_t)std::min<int64_t>(len, 10));
            return 10;
        }
        else return -1;
        if (out) std::memcpy(out, src, sizeof(double) * (size_t)std::min(len, sz));
        return sz;
    }

    void time_kkt(int reps, double* ms, double* bytes) {
        const HostSystem& h = *H;
        for (int64_t i = 0; i < h.n_tot; ++i) p[i] = 1.0 + 1e-3 * (double)(i % 7);
        const double t0 = now_ms();
        for (int rep = 0; rep < reps; ++rep) {
            for (int pi = 0; pi < h.count; ++pi)
                for_rows(h.K, pi, 0, [&](int64_t row, int64_t o, int64_t sh) { w[o] = row_dot(h.K, row, p.data() + sh); });
        }
        *ms = (now_ms() - t0) / std::max(1, reps);
        double bsum = 0;
        for (double v : h.kkt_bytes) bsum += v;
        *bytes = bsum;
    }
};

}  // namespace

struct score_assembled {
    score::AssembledQP qp;
};

struct score_handle {
    score::Solver<CpuBackend> solver;
};

// Local refinement (score_gn.hpp) with every kernel restated as a loop; the damped normal equations go
// through this library's linear mode.
struct score_refine {
    score::GnProblem P;
    score_handle* lin = nullptr;
    double setup_ms = 0;
    std::vector<double> u, ut, step, hblk, gblk, rhs, vals;
    ~score_refine();
    void create(const score_graph& g, const score_settings* s);
    double eval_at(const std::vector<double>& w, bool with_blocks) {
        if (P.dim == 3) return eval_at3(w, with_blocks);
        const double* uu = w.data();
        double f = 0.0;
        for (int64_t m = 0; m < P.n_rel(); ++m) {
            double thi, xi, yi, thj, xj, yj;
            score::gn_pose(uu, P.pin, P.rel_i[m], thi, xi, yi);
            score::gn_pose(uu, P.pin, P.rel_j[m], thj, xj, yj);
            f += score::gn_rel_block(thi, xi, yi, thj, xj, yj, &P.rel_t[2 * m], &P.rel_R[4 * m], P.rel_kappa[m], P.rel_tau[m],
                                     with_blocks ? &hblk[36 * m] : nullptr, with_blocks ? &gblk[6 * m] : nullptr);
        }
        const int64_t hb = 36 * P.n_rel(), gb = 6 * P.n_rel();
        for (int64_t r = 0; r < P.n_rng(); ++r) {
            double xa, ya, xb, yb;
            score::gn_point(uu, P.pin, P.Np, P.rng_a[r], xa, ya);
            score::gn_point(uu, P.pin, P.Np, P.rng_b[r], xb, yb);
            f += score::gn_range_block(xa, ya, xb, yb, P.rng_dist[r], P.rng_prec[r], with_blocks ? &hblk[hb + 16 * r] : nullptr,
                                       with_blocks ? &gblk[gb + 4 * r] : nullptr);
        }
        const int64_t hp = hb + 16 * P.n_rng(), gp = gb + 4 * P.n_rng();
        for (int64_t e = 0; e < P.n_pri(); ++e) {
            const double* l = uu + 3 * (P.Np - 1) + 2 * (int64_t)P.pri_l[e];
            f += score::gn_prior_block(l[0], l[1], &P.pri_t[2 * e], P.pri_prec[e], with_blocks ? &hblk[hp + 2 * e] : nullptr,
                                       with_blocks ? &gblk[gp + 2 * e] : nullptr);
        }
        return f;
    }
    // 3-D: state X = [R | t] of every pose, then the landmarks (score_gn.hpp)
    double eval_at3(const std::vector<double>& w, bool with_blocks) {
        const double* X = w.data();
        double f = 0.0;
        for (int64_t m = 0; m < P.n_rel(); ++m)
            f += score::gn_rel_block3(X + 12 * (int64_t)P.rel_i[m], X + 12 * (int64_t)P.rel_j[m], &P.rel_t[3 * m], &P.rel_R[9 * m],
                                      P.rel_kappa[m], P.rel_tau[m], with_blocks ? &hblk[144 * m] : nullptr, with_blocks ? &gblk[12 * m] : nullptr);
        const int64_t hb = 144 * P.n_rel(), gb = 12 * P.n_rel();
        for (int64_t r = 0; r < P.n_rng(); ++r)
            f += score::gn_range_block3(score::gn_point3(X, P.Np, P.rng_a[r]), score::gn_point3(X, P.Np, P.rng_b[r]), P.rng_dist[r],
                                        P.rng_prec[r], with_blocks ? &hblk[hb + 36 * r] : nullptr, with_blocks ? &gblk[gb + 6 * r] : nullptr);
        const int64_t hp = hb + 36 * P.n_rng(), gp = gb + 6 * P.n_rng();
        for (int64_t e = 0; e < P.n_pri(); ++e)
            f += score::gn_prior_block3(X + 12 * P.Np + 3 * (int64_t)P.pri_l[e], &P.pri_t[3 * e], P.pri_prec[e],
                                        with_blocks ? &hblk[hp + 3 * e] : nullptr, with_blocks ? &gblk[gp + 3 * e] : nullptr);
        return f;
    }
    double eval_current(bool with_blocks) { return eval_at(u, with_blocks); }
    double eval_trial() {
        if (P.dim == 2) {
            for (int64_t i = 0; i < P.n; ++i) ut[i] = u[i] + step[i];
        } else {  // retraction: R <- R Exp(omega), t <- t + v; landmarks additive
            for (int64_t p = 0; p < P.Np; ++p) score::gn_pose3_retract(&u[12 * p], p > 0 ? &step[6 * (p - 1)] : nullptr, &ut[12 * p]);
            for (int64_t l = 0; l < 3 * P.Nl; ++l) ut[12 * P.Np + l] = u[12 * P.Np + l] + step[6 * (P.Np - 1) + l];
        }
        return eval_at(ut, false);
    }
    void accept() { u.swap(ut); }
    double assemble() {
        double m = 0.0;
        for (int64_t i = 0; i < P.n; ++i) {
            double acc = 0.0;
            for (int32_t c = P.gc_ptr[i]; c < P.gc_ptr[i + 1]; ++c) acc += gblk[P.gc_slot[c]];
            rhs[i] = -acc;
            m = std::max(m, acc == acc ? std::fabs(acc) : INFINITY);
        }
        return m;
    }
    bool solve(double lambda, double rel_tol, int* used);
    void run(const double* poses_in, const double* lms_in, int max_iters, double tol, double* poses_out, double* lms_out,
             score::GnInfo& info) {
        if (P.dim == 3) {
            std::copy(poses_in, poses_in + 12 * P.Np, u.begin());
            if (P.Nl) std::copy(lms_in, lms_in + 3 * P.Nl, u.begin() + (std::ptrdiff_t)(12 * P.Np));
            score::gn_levenberg_marquardt(*this, max_iters, tol, 1e-9, info);
            std::copy(u.begin(), u.begin() + (std::ptrdiff_t)(12 * P.Np), poses_out);
            if (P.Nl) std::copy(u.begin() + (std::ptrdiff_t)(12 * P.Np), u.begin() + (std::ptrdiff_t)(12 * P.Np + 3 * P.Nl), lms_out);
            return;
        }
        for (int k = 0; k < 3; ++k) P.pin[k] = poses_in[k];
        for (int64_t p = 1; p < P.Np; ++p)
            for (int k = 0; k < 3; ++k) u[3 * (p - 1) + k] = poses_in[3 * p + k];
        for (int64_t l = 0; l < 2 * P.Nl; ++l) u[3 * (P.Np - 1) + l] = lms_in[l];
        score::gn_levenberg_marquardt(*this, max_iters, tol, 1e-9, info);
        for (int k = 0; k < 3; ++k) poses_out[k] = poses_in[k];
        for (int64_t p = 1; p < P.Np; ++p)
            for (int k = 0; k < 3; ++k) poses_out[3 * p + k] = u[3 * (p - 1) + k];
        for (int64_t l = 0; l < 2 * P.Nl; ++l) lms_out[l] = u[3 * (P.Np - 1) + l];
    }
};

struct score_generated { score::GeneratedBatch B; };

extern "C" {

void score_default_settings(score_settings* s) { score::default_settings(s); }

int score_create_batch(const score_problem* p, int32_t count, const score_settings* s, score_handle** out) {
    try {
#ifdef _OPENMP
        // Default team of the twin: at most 16 threads.  Its loops are short (a test problem has a
        // few hundred rows); on a 256-thread host the default team makes every parallel region
        // cost milliseconds.  Set once: a caller that sizes the team itself (bench.py's cpu_baseline
        // calibrates it with omp_set_num_threads) is not overridden afterwards.
        static std::once_flag team_once;
        std::call_once(team_once, [] {
            if (omp_get_max_threads() > 16) omp_set_num_threads(16);
        });
#endif
        score_settings st;
        if (s) st = *s; else score::default_settings(&st);
        auto* h = new score_handle();
        try {
            h->solver.create(p, count, st);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int score_create(const score_problem* p, const score_settings* s, score_handle** out) {
    return score_create_batch(p, 1, s, out);
}
// (the twin builds the model with the host assembler -- the specification the product's device assembler is tested against)
int score_create_from_graphs(const score_graph* graphs, int32_t count, const score_settings* s, score_handle** out) {
    try {
        if (!graphs || !out || count <= 0) throw std::runtime_error("null argument");
        std::vector<score::AssembledQP> qps((size_t)count);
        std::vector<score::AssembledQP*> ptrs((size_t)count);
        for (int i = 0; i < count; ++i) ptrs[(size_t)i] = &qps[(size_t)i];
        for (int i = 0; i < count; ++i) {  // (the graph checks of score_create_from_graphs)
            score::AssembledQP sk;
            try { score::graph_skeleton(graphs[i], sk); }
            catch (const std::exception& e) { throw std::runtime_error(count > 1 ? "graph " + std::to_string(i) + ": " + e.what() : std::string(e.what())); }
        }
        score::assemble_graphs(graphs, count, ptrs.data());
        std::vector<score_problem> probs((size_t)count);
        for (int i = 0; i < count; ++i) qps[(size_t)i].view(&probs[(size_t)i]);
        const int rc = score_create_batch(probs.data(), count, s, out);
        if (rc == 0) score::est_layout_from_graphs(graphs, count, (*out)->solver.H.xoff, (*out)->solver.est, (*out)->solver.hf.empty() ? -1 : 0);
        return rc;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int score_read_estimates(score_handle* h, int32_t qcqp_directions, double* poses, double* relaxed, double* landmarks, double* ranges,
                         int32_t* degenerate) {
    try {
        if (!h) throw std::runtime_error("null handle");
        if (!h->solver.est.valid()) throw std::runtime_error("score_read_estimates: the handle was not made by score_create_from_graphs");
        std::vector<double> x((size_t)h->solver.H.n_tot);
        h->solver.be.download(h->solver.H, x.data(), nullptr, nullptr);
        score::read_estimates_host(h->solver.est, (qcqp_directions || h->solver.est.dirs_always) ? 1 : 0, x.data(), poses, relaxed, landmarks, ranges, degenerate);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_graphs_connected(const score_graph* graphs, int32_t count) {
    if (!graphs || count < 0) { g_err = "null argument"; return -1; }
    for (int32_t i = 0; i < count; ++i)
        if (!score::graph_connected(graphs[i])) return i + 1;
    return 0;
}
int score_dims(const score_handle* h, int64_t* n_total, int64_t* m_total, int32_t* count) {
    if (!h) { g_err = "null handle"; return -1; }
    if (n_total) *n_total = h->solver.user_n();  // (the programs as given: score_headform.hpp)
    if (m_total) *m_total = h->solver.user_m();
    if (count) *count = h->solver.H.count;
    return 0;
}
int score_solve(score_handle* h, double* x, double* y, double* s, score_info* infos) {
    try { return h->solver.solve(x, y, s, infos); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_reset(score_handle* h) {
    try { h->solver.reset(); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_solve_steps(score_handle* h, int32_t iters, double* x, double* y, double* s, score_info* infos) {
    try { return h->solver.steps(iters, x, y, s, infos); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_newton_steps(score_handle* h, int32_t iters, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        return h->solver.newton_steps(iters, x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_linear_create(const score_problem* pattern, const score_settings* s, score_handle** out) {
    try {
        if (!pattern || !out) throw std::runtime_error("null argument");
        score::LinearPattern L;
        score::make_linear_pattern(*pattern, s, L);
        score_handle* h = nullptr;
        if (score_create_batch(&L.prob, 1, &L.st, &h) != 0) return -1;
        auto& S = h->solver;
        if ((int64_t)S.H.K0.size() != (int64_t)pattern->P_rowptr[pattern->n]) {
            score_destroy(h);
            throw std::runtime_error("score_linear_create: internal pattern differs from the given one");
        }
        S.linear_mode = true;
        S.linear_nnz = (int64_t)S.H.K0.size();
        *out = h;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_linear_solve(score_handle* h, const double* values, const double* rhs, double* x, double rel_tol,
                       int32_t max_iters, int32_t* iters_used, double* rel_residual) {
    try {
        if (!h) throw std::runtime_error("null handle");
        int used = 0;
        const int rc = h->solver.linear_solve(values, rhs, x, rel_tol, max_iters, &used, rel_residual);
        if (iters_used) *iters_used = used;
        return rc;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_time_kkt_apply(score_handle* h, int32_t reps, double* ms, double* bytes) {
    try { h->solver.be.time_kkt(reps, ms, bytes); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_time_iteration(score_handle*, int32_t, int32_t, double* us, int32_t) {
    if (us) for (int k = 0; k < 12; ++k) us[k] = 0.0;  // the twin has no kernels to time
    return 0;
}
int score_debug_time(score_handle*, const char*, int32_t, double* ms) {
    if (ms) *ms = 0.0;  // the twin has no kernels to time
    return 0;
}
int64_t score_debug_get(score_handle* h, const char* name, double* out, int64_t len) {
    return h->solver.be.get_vec(name, out, len);
}
void score_destroy(score_handle* h) { delete h; }
int score_assemble(const score_graph* g, score_assembled** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        auto* a = new score_assembled();
        try {
            score::assemble_graph(*g, a->qp);
        } catch (...) {
            delete a;
            throw;
        }
        *out = a;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_assemble_batch(const score_graph* graphs, int32_t count, score_assembled** out) {
    try {
        if (!graphs || !out || count <= 0) throw std::runtime_error("null argument");
        std::vector<score_assembled*> made((size_t)count, nullptr);
        std::vector<score::AssembledQP*> qps((size_t)count, nullptr);
        try {
            for (int i = 0; i < count; ++i) { made[(size_t)i] = new score_assembled(); qps[(size_t)i] = &made[(size_t)i]->qp; }
            score::assemble_graphs(graphs, count, qps.data());
        } catch (...) {
            for (auto* a : made) delete a;
            throw;
        }
        for (int i = 0; i < count; ++i) out[i] = made[(size_t)i];
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_assembled_view(const score_assembled* a, score_problem* view) {
    if (!a || !view) { g_err = "null argument"; return -1; }
    a->qp.view(view);
    return 0;
}
void score_assembled_free(score_assembled* a) { delete a; }
// (the generator as host loops: the specification the product's kernels are tested against)
int score_generate_manhattan(const score_manhattan_spec* spec, int32_t count, int32_t /*device*/, score_generated** out) {
    try {
        if (!spec || !out) throw std::runtime_error("null argument");
        score::GenSpec S{spec->n_robots, spec->n_poses, spec->n_beacons, spec->side, spec->p_range, spec->sigma_t, spec->sigma_theta, spec->sigma_range, spec->seed,
                         spec->dim == 0 ? 2 : spec->dim};
        auto* g = new score_generated();
        try { score::generate_manhattan_host(S, count, g->B); } catch (...) { delete g; throw; }
        *out = g;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_generated_graph(const score_generated* g, int32_t index, score_graph* view) {
    try {
        if (!g || !view) throw std::runtime_error("null argument");
        g->B.view(index, view);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_generated_truth(const score_generated* g, int32_t index, double* poses, double* beacons) {
    try {
        if (!g) throw std::runtime_error("null argument");
        g->B.truth(index, poses, beacons);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void score_generated_free(score_generated* g) { delete g; }
int score_create_from_generated(const score_generated* g, int32_t first, int32_t count, int32_t relaxation, const score_settings* s, score_handle** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        if (first < 0 || count <= 0 || first + count > g->B.count) throw std::runtime_error("score_create_from_generated: worlds out of range");
        if (relaxation != 0 && relaxation != 1) throw std::runtime_error("score_create_from_generated: relaxation must be 0 (SOCP) or 1 (QCQP)");
        std::vector<score_graph> views((size_t)count);
        for (int i = 0; i < count; ++i) { g->B.view(first + i, &views[(size_t)i]); views[(size_t)i].relaxation = relaxation; }
        return score_create_from_graphs(views.data(), count, s, out);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_round_to_so(int32_t dim, int64_t n, const double* blocks, double* rotations, int32_t* degenerate, int32_t /*device*/) {
    if (dim != 2 && dim != 3) { g_err = "score_round_to_so: dim must be 2 or 3"; return -1; }
    if (n < 0 || (n > 0 && (!blocks || !rotations || !degenerate))) { g_err = "score_round_to_so: null argument"; return -1; }
    for (int64_t i = 0; i < n; ++i) {  // the same per-block function the HIP kernel runs per lane
        if (dim == 2) score::round_so2(blocks + 4 * i, rotations + 4 * i, degenerate + i);
        else score::round_so3(blocks + 9 * i, rotations + 9 * i, degenerate + i);
    }
    return 0;
}
int64_t score_trim_caches(void) { return 0; }  // the twin parks nothing
int32_t score_host_counters(double* out, int32_t len) {  // (the twin waits for no device)
    if (out) for (int i = 0; i < len && i < 4; ++i) out[i] = 0.0;
    return 4;
}
const char* score_last_error(void) { return g_err.c_str(); }
int32_t score_abi_version(void) { return SCORE_ABI_VERSION * 1000 + (int32_t)sizeof(score_problem); }
const char* score_backend(void) { return "cpu-twin"; }
}

score_refine::~score_refine() { if (lin) score_destroy(lin); }
void score_refine::create(const score_graph& g, const score_settings* s) {
    const double t0 = score::now_ms();
    score::gn_build(g, P);
    score_problem pat{};
    pat.n = (int32_t)P.n; pat.m = 0;
    pat.P_rowptr = P.hptr.data(); pat.P_col = P.hcol.data();
    pat.block_size = 3; pat.n_chains = (int32_t)P.chain_ptr.size() - 1;
    pat.chain_ptr = P.chain_ptr.data(); pat.node_first_col = P.node_first_col.data();
    if (score_linear_create(&pat, s, &lin) != 0) throw std::runtime_error(g_err);
    u.assign((size_t)P.state_size(), 0.0); ut = u;
    step.assign((size_t)P.n, 0.0); rhs = step;
    hblk.assign((size_t)std::max<int64_t>(1, P.hblk_size()), 0.0);
    gblk.assign((size_t)std::max<int64_t>(1, P.gblk_size()), 0.0);
    vals.assign(P.hcol.size(), 0.0);
    setup_ms = score::now_ms() - t0;
}
bool score_refine::solve(double lambda, double rel_tol, int* used) {
    for (size_t k = 0; k < P.hcol.size(); ++k) {
        double acc = 0.0;
        for (int32_t c = P.hc_ptr[k]; c < P.hc_ptr[k + 1]; ++c) acc += hblk[P.hc_slot[c]];
        vals[k] = acc;
    }
    for (int64_t i = 0; i < P.n; ++i) vals[(size_t)P.diag_pos[(size_t)i]] += lambda;
    int32_t it = 0;
    const int rc = score_linear_solve(lin, vals.data(), rhs.data(), step.data(), rel_tol, 4000, &it, nullptr);
    if (rc < 0) throw std::runtime_error(g_err);
    if (used) *used = it;
    return rc == 0;
}

extern "C" {
int score_refine_create(const score_graph* g, const score_settings* s, score_refine** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        auto* r = new score_refine();
        try {
            r->create(*g, s);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_refine_run(score_refine* r, const double* poses_in, const double* landmarks_in, int32_t max_iters, double tol,
                     double* poses_out, double* landmarks_out, score_refine_info* info) {
    try {
        if (!r || !poses_in || !poses_out || (r->P.Nl > 0 && (!landmarks_in || !landmarks_out))) throw std::runtime_error("null argument");
        const double t0 = score::now_ms();
        score::GnInfo gi;
        r->run(poses_in, landmarks_in, max_iters, tol, poses_out, landmarks_out, gi);
        if (info) {
            info->cost_initial = gi.cost_initial; info->cost_final = gi.cost_final; info->grad_inf = gi.grad_inf;
            info->iterations = gi.iterations; info->linear_solves = gi.linear_solves; info->pcg_iters = gi.pcg_iters;
            info->setup_ms = r->setup_ms; info->solve_ms = score::now_ms() - t0;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void score_refine_destroy(score_refine* r) { delete r; }
}
