// score_driver.hpp -- the ADMM outer loop, shared by every backend.
//
// The backend owns the iterates and runs blocks of ADMM iterations; the driver
// decides, between blocks, which problems have converged, whether a problem's
// penalty rho should change (which re-factors its preconditioner on the host
// and re-uploads the rho-dependent data), and when to stop.  This is the role
// Gurobi's `model.optimize()` plays for the reference (score/solve_score.py:76);
// `solved` there is `status == OPTIMAL` (gurobi_utils.py:195), here it is
// "all three termination tests passed".
#pragma once

#include <chrono>
#include <cstdio>
#include <vector>

#include "score_host.hpp"
#include "score_assemble.hpp"
#include "score_headform.hpp"

namespace score {

struct ResidualSums {  // per problem, filled by the backend
    // primal side (cone kernel): inf-norms of Ax+s-b, Ax, s  (unscaled, scaled), b'y
    double rp_u = 0, ax_u = 0, s_u = 0, rp_s = 0, ax_s = 0, s_s = 0, by = 0;
    // dual side (G2 SpMV): inf-norms of Px+q+A'y, Px, A'y (unscaled, scaled), x'Px, q'x
    double rd_u = 0, px_u = 0, aty_u = 0, rd_s = 0, px_s = 0, aty_s = 0, xPx = 0, qx = 0;
    // duality gap in its cancellation-free form: pobj - dobj = x'r_d + s'y - y'r_p
    double xrd = 0, sy_yrp = 0;
};

inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class Backend>
struct Solver {
    HostSystem H;
    score_settings st;
    Backend be;
    std::vector<score_info> infos;
    std::vector<int> done;
    int iters_done = 0;
    int cg_now = 2;       // PCG iterations per ADMM iteration currently in use
    int64_t cg_total = 0; // PCG iterations performed since reset
    int next_polish = 0;  // earliest iteration of the next polish attempt
    int next_rho = 0;     // earliest iteration of the next rho adaptation
    double setup_ms = 0;
    std::vector<double> dual_scale;  // per problem: |A'y|_inf of the last residual test (scale of the dual test)

    // Programs of the constant-head unit-ball kind (the reference's default "QCQP" relaxation) are solved in their head form
    // (score_headform.hpp): H, the backend and every iterate live there; sizes, x, y and s at the boundary are the caller's.
    std::vector<HeadForm> hf;          // per problem; empty: the programs are solved as given
    std::vector<int64_t> uxoff, uroff; // offsets of the programs as given (hf non-empty)
    int64_t user_n() const { return hf.empty() ? H.n_tot : uxoff.back(); }
    int64_t user_m() const { return hf.empty() ? H.m_tot : uroff.back(); }
    void download(double* x, double* y, double* s) {
        if (hf.empty()) { be.download(H, x, y, s); return; }
        if (!x && !y && !s) return;
        std::vector<double> xr((size_t)H.n_tot), yr(y ? (size_t)H.m_tot : 0), sr(s ? (size_t)H.m_tot : 0);
        be.download(H, xr.data(), y ? yr.data() : nullptr, s ? sr.data() : nullptr);
        parallel_ranges(H.count, 1, [&](int, int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; ++p)
                headform_expand(hf[(size_t)p], xr.data() + H.xoff[(size_t)p], y ? yr.data() + H.roff[(size_t)p] : nullptr,
                                s ? sr.data() + H.roff[(size_t)p] : nullptr, x ? x + uxoff[(size_t)p] : nullptr,
                                y ? y + uroff[(size_t)p] : nullptr, s ? s + uroff[(size_t)p] : nullptr);
        });
    }

    void create(const score_problem* probs, int count, const score_settings& s, bool keep_hf = false) {
        const double t0 = now_ms();
        st = s;
        std::vector<score_problem> head_probs;
        if (!keep_hf) hf.clear();
        if (!keep_hf && headform_enabled() && count > 0 && probs && probs[0].n_soc > 0 && probs[0].A_rowptr && probs[0].z < probs[0].m &&
            probs[0].A_rowptr[probs[0].z + 1] == probs[0].A_rowptr[probs[0].z]) {  // (first cone's head row is empty: worth a look)
            for (int p = 0; p < count; ++p) validate_problem(probs[p]);
            std::vector<HeadForm> F((size_t)count);
            std::atomic<bool> all{true};
            parallel_ranges(count, 1, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1 && all.load(std::memory_order_relaxed); ++p)
                    if (!headform_reduce(probs[p], F[(size_t)p])) all = false;
            });
            if (all) {
                hf = std::move(F);
                uxoff.assign((size_t)count + 1, 0); uroff.assign((size_t)count + 1, 0);
                head_probs.resize((size_t)count);
                for (int p = 0; p < count; ++p) {
                    headform_rebind(hf[(size_t)p]);
                    head_probs[(size_t)p] = hf[(size_t)p].view;
                    uxoff[(size_t)p + 1] = uxoff[(size_t)p] + probs[p].n;
                    uroff[(size_t)p + 1] = uroff[(size_t)p] + probs[p].m;
                }
                probs = head_probs.data();
                if (st.verbose) std::fprintf(stderr, "[score setup] constant-head unit-ball cones: solved in head form (n %lld -> %lld)\n",
                                             (long long)uxoff.back(), (long long)(hf[0].n1));
            }
        }
        if (st.check_interval < 1) st.check_interval = 25;
        if (st.cg_iters < 1) st.cg_iters = 1;
        if (st.adaptive_rho_interval < st.check_interval) st.adaptive_rho_interval = st.check_interval;
        if (st.max_cg_iters < st.cg_iters) st.max_cg_iters = st.cg_iters;
        if (!(st.cg_target > 0.0 && st.cg_target < 1.0)) st.cg_target = 0.5;
        cg_now = st.cg_iters;
        build_system(probs, count, st, H, Backend::kFactorOnHost, Backend::allow_rep(), be.ruiz_offload(st),
                     [&](const HostSystem& hs) { return be.device_setup_ok(hs, probs, st); });
        be.init(H, st, probs);
        infos.assign(count, score_info{});
        done.assign(count, 0);
        dual_scale.assign(count, 0.0);
        setup_ms = now_ms() - t0;
    }

    // A handle straight from factor graphs (score_create_from_graphs): when the backend can build the model on its device
    // (Backend::device_setup_ok_graphs) the host only lays out sizes, cones and chains (graph_skeleton) and the graphs'
    // arrays are all that crosses the link; otherwise the host assembler builds the programs and create() takes over.
    EstLayout est;  // handles made from factor graphs: what score_read_estimates needs of them
    template <class AssembleFn>
    void create_from_graphs(const score_graph* graphs, int count, const score_settings& s, AssembleFn&& assemble_on_host) {
        const double t0 = now_ms();
        if (count <= 0) throw std::runtime_error("score_create_from_graphs: count must be positive");
        struct KeepLayout {  // (filled once the handle exists, whichever path built it)
            Solver& S; const score_graph* g; int c;
            // (a QCQP graph solved in head form: the head form's columns are the SOCP program's, directions from the translations)
            ~KeepLayout() { if (!std::uncaught_exceptions() && S.H.count == c) est_layout_from_graphs(g, c, S.H.xoff, S.est, S.hf.empty() ? -1 : 0); }
        } keep_layout{*this, graphs, count};
        std::vector<AssembledQP> skel((size_t)count);
        std::vector<score_problem> probs((size_t)count);
        for (int i = 0; i < count; ++i) {
            try { graph_skeleton(graphs[i], skel[(size_t)i]); }
            catch (const std::exception& e) { throw std::runtime_error(count > 1 ? "graph " + std::to_string(i) + ": " + e.what() : std::string(e.what())); }
            skel[(size_t)i].view(&probs[(size_t)i]);
        }
        st = s;
        hf.clear();
        std::vector<score_graph> as_socp;
        bool all_qcqp = true;
        for (int i = 0; i < count; ++i) all_qcqp = all_qcqp && graphs[i].relaxation == 1;
        if (all_qcqp && headform_enabled()) {
            // the direct QCQP form, solved in its head form -- which for a factor graph is the graph's SOCP program
            // (score_headform.hpp): that one is built, on the device; sizes, x, y, s at the boundary stay the QCQP program's
            as_socp.assign(graphs, graphs + count);
            hf.resize((size_t)count);
            uxoff.assign((size_t)count + 1, 0); uroff.assign((size_t)count + 1, 0);
            for (int i = 0; i < count; ++i) {
                as_socp[(size_t)i].relaxation = 0;
                headform_from_graph(graphs[i], hf[(size_t)i]);
                uxoff[(size_t)i + 1] = uxoff[(size_t)i] + hf[(size_t)i].n0;
                uroff[(size_t)i + 1] = uroff[(size_t)i] + hf[(size_t)i].m0;
                graph_skeleton(as_socp[(size_t)i], skel[(size_t)i]);
                skel[(size_t)i].view(&probs[(size_t)i]);
            }
            graphs = as_socp.data();
        }
        if (st.check_interval < 1) st.check_interval = 25;
        if (st.cg_iters < 1) st.cg_iters = 1;
        if (st.adaptive_rho_interval < st.check_interval) st.adaptive_rho_interval = st.check_interval;
        if (st.max_cg_iters < st.cg_iters) st.max_cg_iters = st.cg_iters;
        if (!(st.cg_target > 0.0 && st.cg_target < 1.0)) st.cg_target = 0.5;
        cg_now = st.cg_iters;
        try {
            build_system(probs.data(), count, st, H, Backend::kFactorOnHost, Backend::allow_rep(), nullptr,
                         [&](const HostSystem& hs) { return be.device_setup_ok_graphs(hs, graphs, st); }, /*trusted=*/true);
        } catch (const DeviceSetupDeclined&) {  // (the backend declined: model construction on the host, then the ordinary create)
            assemble_on_host(graphs, !hf.empty());
            return;
        }
        be.init(H, st, nullptr, graphs);
        infos.assign(count, score_info{});
        done.assign(count, 0);
        dual_scale.assign(count, 0.0);
        setup_ms = now_ms() - t0;
    }

    void reset() {
        bool changed = false;
        for (int p = 0; p < H.count; ++p)
            if (H.rho[p] != st.rho) {
                H.rho[p] = st.rho;
                refresh_rho(H, p);
                changed = true;
            }
        if (changed) be.upload_rho(H);
        be.reset();
        if (cg_now != st.cg_iters) { cg_now = st.cg_iters; be.set_cg_iters(cg_now); }
        cg_total = 0;
        next_polish = 0;
        next_rho = st.adaptive_rho_interval;
        iters_done = 0;
        std::fill(done.begin(), done.end(), 0);
        be.set_done(done);
        for (auto& i : infos) i = score_info{};
    }

    // one convergence test; returns true when every problem is finished
    bool check(bool allow_rho) {
        std::vector<ResidualSums> R(H.count);
        be.residuals(R);
        bool all = true, any_rho = false, newly_done = false;
        for (int p = 0; p < H.count; ++p) {
            if (done[p]) continue;
            const ResidualSums& r = R[p];
            score_info& I = infos[p];
            I.status = SCORE_STATUS_UNSOLVED;  // re-evaluated below (steps() un-latches a problem that had passed)
            I.iters = iters_done;
            I.cg_iters = (int32_t)std::min<int64_t>(cg_total, 2147483647);
            I.rho = H.rho[p];
            I.res_pri = r.rp_u;
            I.res_dual = r.rd_u;
            I.pobj = 0.5 * r.xPx + r.qx + H.c0[p];
            // x'Px + q'x + b'y evaluated as x'r_d + s'y - y'r_p: every factor is small near the
            // optimum, whereas the textbook form cancels sums of magnitude 1e5..1e6
            I.gap = std::fabs(r.xrd + r.sy_yrp);
            I.dobj = I.pobj - (r.xrd + r.sy_yrp);
            I.kkt_bytes = H.kkt_bytes[p];
            const bool finite = std::isfinite(r.rp_u) && std::isfinite(r.rd_u) && std::isfinite(I.pobj);
            if (!finite) {
                I.status = SCORE_STATUS_NUMERICAL;
                done[p] = 1;
                newly_done = true;
                continue;
            }
            const double pn = std::max(std::max(r.ax_u, r.s_u), H.bnorm_u[p]);
            // dual scale: the constraint forces only.  |Px| and |q| carry the stiff
            // pinned-pose constants (1e5..1e6) that cancel in Px + q and would make a
            // relative test vacuous.
            const double dn = r.aty_u;
            dual_scale[p] = dn;
            const bool ok_p = r.rp_u <= st.eps_abs + st.eps_rel * pn;
            const bool ok_d = r.rd_u <= st.eps_abs + st.eps_rel * dn;
            // gap scale as in SCS: the magnitudes of the terms it is made of (|x'r_d| alone is up to
            // |x|_1 |r_d|_inf, far above eps * |pobj| for coordinates in the hundreds of metres)
            const bool ok_g = I.gap <= st.eps_abs + st.eps_rel * std::max(std::max(std::fabs(r.xPx), std::fabs(r.qx)), std::fabs(r.by));
            if (st.verbose)
                std::fprintf(stderr, "[score] prob %d it %d rp %.3e rd %.3e gap %.3e pobj %.9g rho %.3g\n", p,
                             iters_done, r.rp_u, r.rd_u, I.gap, I.pobj, H.rho[p]);
            if (ok_p && ok_d && ok_g) {
                I.status = SCORE_STATUS_SOLVED;
                done[p] = 1;
                newly_done = true;
                continue;
            }
            all = false;
            if (allow_rho && st.adaptive_rho && iters_done >= next_rho) {
                const double tiny = 1e-30;
                const double pns = std::max(std::max(std::max(r.ax_s, r.s_s), H.bnorm_s[p]), tiny);
                const double dns = std::max(std::max(std::max(r.px_s, r.aty_s), H.qnorm_s[p]), tiny);
                double ratio = std::sqrt((r.rp_s / pns) / (std::max(r.rd_s, tiny) / dns));
                double nr = std::min(1e6, std::max(1e-6, H.rho[p] * ratio));
                if (nr > st.adaptive_rho_tol * H.rho[p] || nr < H.rho[p] / st.adaptive_rho_tol) {
                    H.rho[p] = nr;
                    refresh_rho(H, p);
                    I.rho_updates++;
                    any_rho = true;
                }
            }
        }
        if (allow_rho && st.adaptive_rho && iters_done >= next_rho) next_rho = iters_done + st.adaptive_rho_interval;
        if (any_rho) be.upload_rho(H);
        if (newly_done) be.set_done(done);
        if (allow_rho && st.adaptive_cg && !all) {
            // worst measured reduction of the M^-1-norm KKT residual over the active problems
            std::vector<double> red(H.count, 0.0);
            be.cg_reduction(red);
            double worst = 0.0;
            for (int p = 0; p < H.count; ++p)
                if (!done[p] && red[p] == red[p]) worst = std::max(worst, red[p]);
            int want = cg_now;
            if (worst > st.cg_target) want = std::min(st.max_cg_iters, cg_now < 4 ? cg_now + 1 : cg_now + cg_now / 2);
            else if (worst < 0.01 * st.cg_target && cg_now > st.cg_iters) want = std::max(st.cg_iters, cg_now - std::max(1, cg_now / 4));
            if (st.verbose) std::fprintf(stderr, "[score] it %d cg %d reduction %.3e -> cg %d\n", iters_done, cg_now, worst, want);
            if (want != cg_now) { cg_now = want; be.set_cg_iters(cg_now); }
        }
        return all;
    }

    void finish(double* x, double* y, double* s, score_info* out, double t0) {
        const double ms = now_ms() - t0;
        for (int p = 0; p < H.count; ++p) {
            if (!done[p] && infos[p].status == SCORE_STATUS_UNSOLVED) infos[p].status = SCORE_STATUS_MAX_ITERS;
            infos[p].setup_ms = setup_ms;
            infos[p].solve_ms = ms;
        }
        download(x, y, s);
        if (out)
            for (int p = 0; p < H.count; ++p) out[p] = infos[p];
    }

    int solve(double* x, double* y, double* s, score_info* out) {
        const double t0 = now_ms();
        reset();
        auto mark = [&](const char* what) {
            if (st.verbose) std::fprintf(stderr, "[score] solve timeline: %-28s t %.3f ms\n", what, now_ms() - t0);
        };
        mark("reset");
        bool all = false;
        // with the polish on, the first block is shorter: Newton is globally convergent, a rough
        // ADMM iterate is all it needs (measured: 6 iterations beat 15 by 9-13 % of the solve time on the
        // BASELINE sizes and tie with 8-15 on 144 small random graphs; 15 had beaten 25 by ~8 %)
        const bool can_polish = st.polish && be.polish_available();
        while (!all && iters_done < st.max_iters) {
            int k = std::min(st.check_interval, st.max_iters - iters_done);
            if (iters_done == 0 && can_polish && st.polish_warmup > 0) k = std::min(k, st.polish_warmup);
            be.run(k);
            iters_done += k;
            cg_total += (int64_t)k * cg_now;
            mark("ADMM block queued");
            all = check(true);
            mark("residual test");
            if (!all && can_polish && iters_done >= next_polish) {
                // Newton is globally convergent here (convex, line search), so by default it starts
                // right after the first launch graph; polish_start can demand a closer ADMM iterate.
                double worst = 0.0;  // over the problems still running
                for (int p = 0; p < H.count; ++p)
                    if (!done[p]) worst = std::max(worst, std::max(infos[p].res_pri, infos[p].res_dual));
                if (worst <= st.polish_start) {
                    // First to the tolerance the residual test below will apply -- its dual test is relative
                    // to |A'y|, known from the last check -- and only if that test then fails (the scale
                    // moved) on to Newton's own, absolute, tolerance: a Newton iteration costs ten times a
                    // residual test.  Whether Newton converged or stalled in rounding, the point it hands
                    // back is a consistent ADMM state: the ordinary residual test decides.
                    for (int pass = 0; pass < 2 && !all; ++pass) {
                        int nit = 0, ncg = 0;
                        const bool ran = be.polish(H, st, done, &nit, &ncg, pass == 0 ? &dual_scale : nullptr);
                        for (int p = 0; p < H.count; ++p)
                            if (!done[p]) { infos[p].newton_iters += nit; infos[p].newton_cg_iters += ncg; }
                        if (!ran) break;
                        mark("Newton polish");
                        all = check(false);
                        mark("residual test");
                    }
                    next_polish = iters_done + 20 * st.check_interval;  // a failed attempt is retried later
                }
            }
        }
        finish(x, y, s, out, t0);
        mark("solution copied out");
        return 0;
    }

    // Intermediate iterates of the PRODUCT's default trajectory: after the ADMM warm-up the solver
    // switches to Newton's method on the reduced problem; this runs at most `iters` Newton iterations
    // from the current iterate and reports like steps().  Backends without the polish do nothing.
    int newton_steps(int iters, double* x, double* y, double* s, score_info* out) {
        const double t0 = now_ms();
        if (iters > 0 && be.polish_available()) {
            int nit = 0, ncg = 0;
            const std::vector<int> none(H.count, 0);
            be.set_newton_limit(iters);
            const bool ran = be.polish(H, st, none, &nit, &ncg, nullptr);
            be.set_newton_limit(0);
            if (ran)
                for (int p = 0; p < H.count; ++p) { infos[p].newton_iters += nit; infos[p].newton_cg_iters += ncg; }
        }
        return report(x, y, s, out, t0);
    }

    // ---- linear mode (score_linear_create / score_linear_solve): the handle is a chain-preconditioned
    //      PCG solver for SPD systems on the sparsity pattern it was created with ----
    bool linear_mode = false;
    int64_t linear_nnz = 0;
    // K x = rhs with the values of K given on the creation pattern; returns 0 converged, 1 iteration cap
    int linear_solve(const double* values, const double* rhs, double* x, double rel_tol, int max_iters,
                     int* iters_used, double* rel_residual) {
        if (!linear_mode) throw std::runtime_error("score_linear_solve: handle was not made by score_linear_create");
        if (!values || !rhs || !x) throw std::runtime_error("score_linear_solve: null argument");
        if (!(rel_tol > 0.0) || max_iters < 1) throw std::runtime_error("score_linear_solve: bad tolerance / iteration cap");
        bool zero_rhs = true;
        for (int64_t i = 0; i < H.n_tot && zero_rhs; ++i) zero_rhs = (rhs[i] == 0.0);
        if (zero_rhs) {
            std::fill(x, x + H.n_tot, 0.0);
            if (iters_used) *iters_used = 0;
            if (rel_residual) *rel_residual = 0.0;
            return 0;
        }
        std::copy(values, values + linear_nnz, H.K0.begin());  // K1 = 0 (no constraints): K = K0
        for (int p = 0; p < H.count; ++p) refresh_rho(H, p);     // (host factorisation: CPU twin only)
        int used = 0;
        const bool ok = be.linear_solve(H, rhs, x, rel_tol, max_iters, &used);
        if (iters_used) *iters_used = used;
        if (rel_residual) {  // |rhs - K x|_2 / |rhs|_2 on the host
            double rr = 0.0, bb = 0.0;
            for (int64_t i = 0; i < H.n_tot; ++i) {
                double acc = rhs[i];
                for (int k = H.K.ptr[i]; k < H.K.ptr[i + 1]; ++k) acc -= H.K0[k] * x[H.K.col[k]];
                rr += acc * acc;
                bb += rhs[i] * rhs[i];
            }
            *rel_residual = bb > 0.0 ? std::sqrt(rr / bb) : std::sqrt(rr);
        }
        return ok ? 0 : 1;
    }

    int steps(int iters, double* x, double* y, double* s, score_info* out) {
        const double t0 = now_ms();
        int left = iters;
        while (left > 0) {
            const int k = std::min(st.check_interval, left);
            be.run(k);
            iters_done += k;
            cg_total += (int64_t)k * cg_now;
            left -= k;
        }
        return report(x, y, s, out, t0);
    }

    // residual test + snapshot of the current iterate without latching any problem
    int report(double* x, double* y, double* s, score_info* out, double t0) {
        std::vector<int> keep = done;
        check(false);
        for (int p = 0; p < H.count; ++p) {
            // stepping never freezes a problem: report, do not latch
            if (!keep[p] && done[p] && infos[p].status == SCORE_STATUS_SOLVED) done[p] = 0;
        }
        be.set_done(done);
        std::vector<score_info> snap = infos;
        for (int p = 0; p < H.count; ++p)
            if (snap[p].status == SCORE_STATUS_UNSOLVED) snap[p].status = SCORE_STATUS_MAX_ITERS;
        const double ms = now_ms() - t0;
        download(x, y, s);
        if (out)
            for (int p = 0; p < H.count; ++p) {
                out[p] = snap[p];
                out[p].setup_ms = setup_ms;
                out[p].solve_ms = ms;
            }
        return 0;
    }
};

inline void default_settings(score_settings* s) {
    s->eps_abs = 1e-7;
    s->eps_rel = 1e-7;
    s->max_iters = 20000;
    s->check_interval = 25;
    s->rho = 0.1;
    s->sigma = 1e-6;
    s->alpha = 1.8;
    s->scale_iters = 10;
    s->cg_iters = 2;
    s->adaptive_cg = 1;
    s->max_cg_iters = 64;
    s->cg_target = 0.5;
    s->adaptive_rho = 1;
    s->adaptive_rho_interval = 100;
    s->adaptive_rho_tol = 5.0;
    s->chain_radix = 4;
    s->device = 0;
    s->use_graph = 1;
    s->polish = 1;
    s->polish_start = 1e30;
    s->polish_warmup = 6;  // (measured round 3: 6-10 beat 15 now that a Newton iteration costs less; profiles/scripts/r03_warmup.py)
    s->verbose = 0;
    s->chain_split = 0;
    s->fac_fp32 = 1;
}

// score_linear_create: the pattern problem behind a linear-mode handle.  `pat` gives the sparsity pattern
// of an SPD matrix (CSR, columns strictly increasing, every row holds its diagonal) and the chain hint;
// the placeholder values are the identity, so that the handle is valid before the first solve.
struct LinearPattern {
    std::vector<double> val, q, none_d{0.0};
    std::vector<int32_t> a_ptr{0}, none_i{0};
    score_problem prob{};
    score_settings st{};
};
inline void make_linear_pattern(const score_problem& pat, const score_settings* s, LinearPattern& L) {
    if (pat.n <= 0 || !pat.P_rowptr || !pat.P_col) throw std::runtime_error("score_linear_create: empty pattern");
    if (pat.m != 0 || pat.n_soc != 0 || pat.z != 0) throw std::runtime_error("score_linear_create: the pattern problem must have no constraints (m = 0)");
    const int64_t nnz = pat.P_rowptr[pat.n];
    L.val.assign((size_t)nnz, 0.0);
    for (int i = 0; i < pat.n; ++i) {
        bool diag = false;
        for (int k = pat.P_rowptr[i]; k < pat.P_rowptr[i + 1]; ++k) {
            const int j = pat.P_col[k];
            if (j < 0 || j >= pat.n) throw std::runtime_error("score_linear_create: column out of range");
            if (k > pat.P_rowptr[i] && pat.P_col[k - 1] >= j) throw std::runtime_error("score_linear_create: columns of a row must be strictly increasing");
            if (j == i) { diag = true; L.val[(size_t)k] = 1.0; }
        }
        if (!diag) throw std::runtime_error("score_linear_create: every row must hold its diagonal entry");
    }
    L.q.assign((size_t)pat.n, 0.0);
    L.prob = pat;
    L.prob.P_val = L.val.data();
    L.prob.q = L.q.data();
    L.prob.c0 = 0.0;
    L.prob.m = 0; L.prob.z = 0; L.prob.n_soc = 0;
    L.prob.A_rowptr = L.a_ptr.data(); L.prob.A_col = L.none_i.data(); L.prob.A_val = L.none_d.data();
    L.prob.b = L.none_d.data(); L.prob.soc_dims = L.none_i.data();
    if (s) L.st = *s; else default_settings(&L.st);
    L.st.scale_iters = 0;  // the caller's values are used as they are
    L.st.polish = 0;
    L.st.adaptive_rho = 0;
    L.st.use_graph = 0;
}


}  // namespace score
