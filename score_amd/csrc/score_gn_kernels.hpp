// score_gn_kernels.hpp -- device side of the local refinement (score_gn.hpp): per-measurement blocks,
// the gathers that turn them into J'J / J'r on the handle's pattern, trial points.
// All are small streaming kernels (one measurement / matrix entry / unknown per lane); the work that
// matters -- the damped normal equations -- runs in the solver's own k_factor / k_prec_pre / k_spmv.
#pragma once

#include "score_gn.hpp"
#include "score_kernels.hpp"

namespace score {

struct GnDev {
    int64_t Np, Nl, n, n_rel, n_rng, n_pri;
    const int32_t *rel_i, *rel_j, *rng_a, *rng_b, *pri_l;
    const double *rel_t, *rel_R, *rel_kappa, *rel_tau, *rng_dist, *rng_prec, *pri_t, *pri_prec;
    const double* pin;  // theta, x, y of the fixed pose
};

// one measurement per lane: cost (block partial sums) and, with_blocks, its J'J / J'r block
__global__ __launch_bounds__(kThreads) void k_gn_blocks(GnDev d, const double* __restrict__ u, double* __restrict__ hblk,
                                                        double* __restrict__ gblk, double* __restrict__ cost_part, int with_blocks) {
    __shared__ double red[8];
    const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double cost = 0.0;
    if (m < d.n_rel) {
        double thi, xi, yi, thj, xj, yj;
        gn_pose(u, d.pin, d.rel_i[m], thi, xi, yi);
        gn_pose(u, d.pin, d.rel_j[m], thj, xj, yj);
        double H[36], g[6];
        cost = gn_rel_block(thi, xi, yi, thj, xj, yj, d.rel_t + 2 * m, d.rel_R + 4 * m, d.rel_kappa[m], d.rel_tau[m],
                            with_blocks ? H : nullptr, g);
        if (with_blocks) {
#pragma unroll
            for (int k = 0; k < 36; ++k) hblk[36 * m + k] = H[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) gblk[6 * m + k] = g[k];
        }
    } else if (m < d.n_rel + d.n_rng) {
        const int64_t r = m - d.n_rel;
        double xa, ya, xb, yb;
        gn_point(u, d.pin, d.Np, d.rng_a[r], xa, ya);
        gn_point(u, d.pin, d.Np, d.rng_b[r], xb, yb);
        double H[16], g[4];
        cost = gn_range_block(xa, ya, xb, yb, d.rng_dist[r], d.rng_prec[r], with_blocks ? H : nullptr, g);
        if (with_blocks) {
            double* ho = hblk + 36 * d.n_rel + 16 * r;
            double* go = gblk + 6 * d.n_rel + 4 * r;
#pragma unroll
            for (int k = 0; k < 16; ++k) ho[k] = H[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) go[k] = g[k];
        }
    } else if (m < d.n_rel + d.n_rng + d.n_pri) {
        const int64_t e = m - d.n_rel - d.n_rng;
        const double* l = u + 3 * (d.Np - 1) + 2 * (int64_t)d.pri_l[e];
        double H[2], g[2];
        cost = gn_prior_block(l[0], l[1], d.pri_t + 2 * e, d.pri_prec[e], with_blocks ? H : nullptr, g);
        if (with_blocks) {
            double* ho = hblk + 36 * d.n_rel + 16 * d.n_rng + 2 * e;
            double* go = gblk + 6 * d.n_rel + 4 * d.n_rng + 2 * e;
            ho[0] = H[0]; ho[1] = H[1]; go[0] = g[0]; go[1] = g[1];
        }
    }
    const double tot = block_sum(cost, red);
    if (threadIdx.x == 0) cost_part[blockIdx.x] = tot;
}

// 3-D: the same over the state X = [R | t] per pose, landmarks (score_gn.hpp); blocks 12 x 12 / 6 x 6 / 3
__global__ __launch_bounds__(kThreads) void k_gn_blocks3(GnDev d, const double* __restrict__ X, double* __restrict__ hblk,
                                                         double* __restrict__ gblk, double* __restrict__ cost_part, int with_blocks) {
    __shared__ double red[8];
    const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double cost = 0.0;
    if (m < d.n_rel) {
        double Xi[12], Xj[12], H[144], g[12];
        const double* pi = X + 12 * (int64_t)d.rel_i[m];
        const double* pj = X + 12 * (int64_t)d.rel_j[m];
#pragma unroll
        for (int k = 0; k < 12; ++k) { Xi[k] = pi[k]; Xj[k] = pj[k]; }
        cost = gn_rel_block3(Xi, Xj, d.rel_t + 3 * m, d.rel_R + 9 * m, d.rel_kappa[m], d.rel_tau[m], with_blocks ? H : nullptr, g);
        if (with_blocks) {
            for (int k = 0; k < 144; ++k) hblk[144 * m + k] = H[k];
#pragma unroll
            for (int k = 0; k < 12; ++k) gblk[12 * m + k] = g[k];
        }
    } else if (m < d.n_rel + d.n_rng) {
        const int64_t r = m - d.n_rel;
        const double* pa = gn_point3(X, d.Np, d.rng_a[r]);
        const double* pb = gn_point3(X, d.Np, d.rng_b[r]);
        const double a3[3] = {pa[0], pa[1], pa[2]}, b3[3] = {pb[0], pb[1], pb[2]};
        double H[36], g[6];
        cost = gn_range_block3(a3, b3, d.rng_dist[r], d.rng_prec[r], with_blocks ? H : nullptr, g);
        if (with_blocks) {
            double* ho = hblk + 144 * d.n_rel + 36 * r;
            double* go = gblk + 12 * d.n_rel + 6 * r;
#pragma unroll
            for (int k = 0; k < 36; ++k) ho[k] = H[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) go[k] = g[k];
        }
    } else if (m < d.n_rel + d.n_rng + d.n_pri) {
        const int64_t e = m - d.n_rel - d.n_rng;
        const double* l = X + 12 * d.Np + 3 * (int64_t)d.pri_l[e];
        const double l3[3] = {l[0], l[1], l[2]};
        double H[3], g[3];
        cost = gn_prior_block3(l3, d.pri_t + 3 * e, d.pri_prec[e], with_blocks ? H : nullptr, g);
        if (with_blocks) {
            double* ho = hblk + 144 * d.n_rel + 36 * d.n_rng + 3 * e;
            double* go = gblk + 12 * d.n_rel + 6 * d.n_rng + 3 * e;
            for (int k = 0; k < 3; ++k) { ho[k] = H[k]; go[k] = g[k]; }
        }
    }
    const double tot = block_sum(cost, red);
    if (threadIdx.x == 0) cost_part[blockIdx.x] = tot;
}

// 3-D trial point: one pose or landmark per lane, Xt = retract(X, step)  (pose 0 has no step)
__global__ __launch_bounds__(kThreads) void k_gn_trial3(const double* __restrict__ X, const double* __restrict__ step,
                                                        double* __restrict__ Xt, int64_t Np, int64_t Nl) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < Np) {
        double in[12], st[6], out[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) in[k] = X[12 * i + k];
        if (i > 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) st[k] = step[6 * (i - 1) + k];
        }
        gn_pose3_retract(in, i > 0 ? st : nullptr, out);
#pragma unroll
        for (int k = 0; k < 12; ++k) Xt[12 * i + k] = out[k];
    } else if (i < Np + Nl) {
        const int64_t l = i - Np;
        for (int k = 0; k < 3; ++k) Xt[12 * Np + 3 * l + k] = X[12 * Np + 3 * l + k] + step[6 * (Np - 1) + 3 * l + k];
    }
}

// one entry of H per lane: the sum of its block slots in list order (+ lambda on the diagonal)
__global__ __launch_bounds__(kThreads) void k_gn_gather_h(const int32_t* __restrict__ hc_ptr, const int32_t* __restrict__ hc_slot,
                                                          const double* __restrict__ hblk, const int32_t* __restrict__ is_diag,
                                                          double lambda, double* __restrict__ out, int64_t nnz) {
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= nnz) return;
    double acc = 0.0;
    for (int32_t c = hc_ptr[k]; c < hc_ptr[k + 1]; ++c) acc += hblk[hc_slot[c]];
    out[k] = is_diag[k] ? acc + lambda : acc;
}

// one unknown per lane: g = J'r, rhs = -g, block partial of |g|_inf
__global__ __launch_bounds__(kThreads) void k_gn_gather_g(const int32_t* __restrict__ gc_ptr, const int32_t* __restrict__ gc_slot,
                                                          const double* __restrict__ gblk, double* __restrict__ rhs,
                                                          double* __restrict__ gmax_part, int64_t n) {
    __shared__ double red[8];
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double a = 0.0;
    if (i < n) {
        double acc = 0.0;
        for (int32_t c = gc_ptr[i]; c < gc_ptr[i + 1]; ++c) acc += gblk[gc_slot[c]];
        rhs[i] = -acc;
        a = acc == acc ? fabs(acc) : INFINITY;
    }
    const double mx = block_max(a, red);
    if (threadIdx.x == 0) gmax_part[blockIdx.x] = mx;
}

__global__ __launch_bounds__(kThreads) void k_gn_trial(const double* __restrict__ u, const double* __restrict__ step,
                                                       double* __restrict__ ut, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) ut[i] = u[i] + step[i];
}

}  // namespace score
