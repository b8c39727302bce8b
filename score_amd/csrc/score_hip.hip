// score_hip.hip -- MI355X (gfx950) backend of the SCORE conic solver + C ABI.
//
// One ADMM ("SOCP") iteration on the device, all problems of the batch in
// lock-step, per-problem scalars kept in device memory (no host round trip):
//
//   k_spmv<RHS>    r  = sigma x - q + [-K | A'] [xt ; u]       CSR-stream SpMV
//   k_prec<INIT>   z  = M^-1 r ; p = z ; partial r'z           chain + Jacobi
//   repeat cg_iters times:
//     k_spmv<KP>   w  = K p ; partial p'w                      CSR-stream SpMV
//     k_prec<STEP> a = r'z / p'w ; xt += a p ; r -= a w ; z = M^-1 r ; partial r'z
//     k_pupdate    p  = z + (r'z_new / r'z_old) p
//   (the last CG iteration replaces STEP/pupdate by k_xupdate:
//                  xt += a p ; x = alpha xt + (1 - alpha) x)
//   k_cone         v = alpha (b - A xt) + (1 - alpha) s ; s = Proj_K(v - y/rho) ;
//                  y += rho (s - v) ; u = rho (b - s) - y      one cone per lane
//
// Kernel design notes (gfx950):
//  * The SpMV is HBM/L2-bandwidth work (12 B per nonzero, 2 flop): each
//    256-thread workgroup owns a tile of <= 256 rows / <= 3072 nonzeros, reads
//    values and column indices with fully coalesced 8/4-byte loads (12
//    independent loads per lane in flight before the first use), gathers the
//    vector, stages the products in LDS and lets one lane per row add its
//    segment in CSR order -- so the result is bitwise independent of the
//    launch geometry.  Rows longer than 48 nonzeros (landmarks) get a
//    workgroup of their own and a shuffle/LDS tree reduction.
//  * Dot products are never atomics: every workgroup writes one partial, and
//    each consumer workgroup re-reduces the partials of its problem in a fixed
//    order (a few KiB from L2) -- deterministic and one launch shorter than a
//    separate finalise kernel.
//  * The preconditioner is a direct solve of the per-robot block-tridiagonal
//    part of K, factored on the host as a radix-p nested dissection: one
//    workgroup per chain, one lane per run of p-1 nodes (bs x bs blocks in
//    registers), coarse levels in LDS, O(p log_p N) dependent steps instead of
//    2N.
//  * No MFMA anywhere: nothing here is a dense contraction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "score_driver.hpp"

namespace {

using namespace score;

thread_local std::string g_err;

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));  \
    } while (0)

constexpr int kThreads = 256;
constexpr int kUnroll = kTileNnz / kThreads;  // 12 nonzeros per lane

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
// Sum over the 256-thread block, result in every thread.  `red` >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max(double v, double* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// NaN-propagating max for residual norms
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }

// Fixed-order re-reduction of per-workgroup partials [lo, hi).
__device__ __forceinline__ double reduce_partials(const double* __restrict__ part, int lo, int hi, double* red) {
    double acc = 0.0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kThreads) acc += part[i];
    return block_sum(acc, red);
}

struct CsrDev {
    const int32_t* ptr;
    const int32_t* col;
    const double* val;
    const int32_t* first_row;  // row blocks
    const int32_t* blk_prob;
    const int32_t* split;      // G2 only
    int nblocks;
};

struct SpmvArgs {
    CsrDev M;
    const double* xin;      // gathered vector
    const int32_t* done;
    // RHS
    const double* x;        // current x (same buffer as xy)
    const double* q;
    double* r;
    double sigma;
    // KP
    const double* p;
    double* w;
    double* pw_part;
    // DRES
    const double* invD;
    double* dres_part;      // 8 per block
};

enum { MODE_RHS = 0, MODE_KP = 1, MODE_DRES = 2 };

// ---------------------------------------------------------------------------
// CSR-stream SpMV with fused epilogues
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_spmv(SpmvArgs a) {
    __shared__ double prod[kTileNnz];
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = a.M.blk_prob[b];
    if (a.done[prob]) return;
    const int t = threadIdx.x;
    const int r0 = a.M.first_row[b], r1 = a.M.first_row[b + 1];
    const int k0 = a.M.ptr[r0], k1 = a.M.ptr[r1];
    const int nn = k1 - k0;
    const double* __restrict__ val = a.M.val;
    const int32_t* __restrict__ col = a.M.col;
    const double* __restrict__ xin = a.xin;

    int row = r0 + t;
    bool has_row = false;
    double sum = 0.0, sum2 = 0.0;  // sum2: A' part (MODE_DRES)

    if (r1 - r0 == 1 && nn > kLongRow) {
        // one long row: strided partial sums + tree reduction
        double acc = 0.0, acc2 = 0.0;
        const int split = (MODE == MODE_DRES) ? a.M.split[r0] : k1;
        for (int k = k0 + t; k < k1; k += kThreads) {
            const double v = val[k] * xin[col[k]];
            if (MODE == MODE_DRES && k >= split) acc2 += v; else acc += v;
        }
        sum = block_sum(acc, red);
        if (MODE == MODE_DRES) sum2 = block_sum(acc2, red);
        has_row = (t == 0);
        row = r0;
    } else {
        int32_t c[kUnroll];
        double v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int k = t + u * kThreads;
            if (k < nn) {
                c[u] = col[k0 + k];
                v[u] = val[k0 + k];
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int k = t + u * kThreads;
            if (k < nn) prod[k] = v[u] * xin[c[u]];
        }
        __syncthreads();
        if (row < r1) {
            has_row = true;
            const int a0 = a.M.ptr[row] - k0, a1 = a.M.ptr[row + 1] - k0;
            if (MODE == MODE_DRES) {
                const int sp = a.M.split[row] - k0;
                for (int k = a0; k < sp; ++k) sum += prod[k];
                for (int k = sp; k < a1; ++k) sum2 += prod[k];
            } else {
                for (int k = a0; k < a1; ++k) sum += prod[k];
            }
        }
    }

    if (MODE == MODE_RHS) {
        if (has_row) a.r[row] = a.sigma * a.x[row] - a.q[row] + sum;
    } else if (MODE == MODE_KP) {
        double local = 0.0;
        if (has_row) {
            a.w[row] = sum;
            local = a.p[row] * sum;
        }
        const double tot = block_sum(local, red);
        if (t == 0) a.pw_part[b] = tot;
    } else {  // MODE_DRES: sum = (P x)_i, sum2 = (A'y)_i ; xin = [x ; y]
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, s0 = 0, s1 = 0;
        if (has_row) {
            const double qi = a.q[row];
            const double dr = sum + qi + sum2;
            const double id = a.invD[row];
            m0 = fabs(dr) * id; m1 = fabs(sum) * id; m2 = fabs(sum2) * id;
            m3 = fabs(dr); m4 = fabs(sum); m5 = fabs(sum2);
            if (dr != dr) { m0 = dr; m3 = dr; }
            const double xi = xin[row];
            s0 = xi * sum;
            s1 = qi * xi;
        }
        // NaN-safe: a NaN anywhere makes the sums NaN, which the host checks
        const double nanflag = block_sum((m0 != m0) ? 1.0 : 0.0, red);
        m0 = block_max(m0 != m0 ? 0.0 : m0, red); m1 = block_max(m1, red); m2 = block_max(m2, red);
        m3 = block_max(m3 != m3 ? 0.0 : m3, red); m4 = block_max(m4, red); m5 = block_max(m5, red);
        s0 = block_sum(s0, red); s1 = block_sum(s1, red);
        if (t == 0) {
            double* o = a.dres_part + (size_t)b * 8;
            const double bad = nanflag > 0.0 ? __builtin_nan("") : 0.0;
            o[0] = m0 + bad; o[1] = m1; o[2] = m2; o[3] = m3 + bad; o[4] = m4; o[5] = m5; o[6] = s0; o[7] = s1;
        }
    }
}

// ---------------------------------------------------------------------------
// preconditioner: multi-level block-tridiagonal chain solve + Jacobi
// ---------------------------------------------------------------------------
struct PrecArgs {
    const PrecWork* work;
    const ChainDesc* chains;
    const ChainLevelDesc* levels;
    const double* fac;
    const int32_t* node_col;
    const int32_t* diag_cols;
    const double* dinv;
    const int32_t* done;
    const int32_t* prec_part_ptr;  // per problem: range of prec work items
    const int32_t* kblk_part_ptr;  // per problem: range of K row blocks
    double* r;
    double* z;
    double* p;
    const double* w;
    double* xt;
    const double* rz_in;   // partials of the previous r'z   (STEP)
    const double* pw_part; // partials of p'w                (STEP)
    double* rz_out;        // one partial per work item
};

enum { PREC_INIT = 0, PREC_STEP = 1 };

template <int BS>
__device__ __forceinline__ void matvec_sub(const double* __restrict__ M, const double (&v)[BS], double (&out)[BS]) {
    // out -= M v   (M row-major BS x BS)
#pragma unroll
    for (int c = 0; c < BS; ++c) {
        double s = out[c];
#pragma unroll
        for (int k = 0; k < BS; ++k) s -= M[c * BS + k] * v[k];
        out[c] = s;
    }
}
template <int BS>
__device__ __forceinline__ void matvec_t_sub(const double* __restrict__ M, const double (&v)[BS], double (&out)[BS]) {
    // out -= M' v
#pragma unroll
    for (int c = 0; c < BS; ++c) {
        double s = out[c];
#pragma unroll
        for (int k = 0; k < BS; ++k) s -= M[k * BS + c] * v[k];
        out[c] = s;
    }
}

template <int BS, int RMAX, int MODE>
__global__ __launch_bounds__(kThreads) void k_prec(PrecArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [0,8): reductions, then chain scratch
    double* red = lds;
    double* scr = lds + 8;
    const PrecWork wk = a.work[blockIdx.x];
    const int prob = wk.prob;
    if (a.done[prob]) return;
    const int t = threadIdx.x;
    double alpha = 0.0;
    if (MODE == PREC_STEP) {
        const double rz = reduce_partials(a.rz_in, a.prec_part_ptr[prob], a.prec_part_ptr[prob + 1], red);
        const double pw = reduce_partials(a.pw_part, a.kblk_part_ptr[prob], a.kblk_part_ptr[prob + 1], red);
        alpha = pw > 0.0 ? rz / pw : 0.0;
    }
    double local = 0.0;
    if (wk.kind == 1) {
        for (int e = wk.index + t; e < wk.index + wk.count; e += kThreads) {
            const int col = a.diag_cols[e];
            double rv = a.r[col];
            if (MODE == PREC_STEP) {
                a.xt[col] += alpha * a.p[col];
                rv -= alpha * a.w[col];
                a.r[col] = rv;
            }
            const double zv = rv * a.dinv[e];
            a.z[col] = zv;
            if (MODE == PREC_INIT) a.p[col] = zv;
            local += rv * zv;
        }
    } else {
        constexpr int B2 = BS * BS;
        const ChainDesc ch = a.chains[wk.index];
        const ChainLevelDesc* __restrict__ lv = a.levels + ch.level_begin;
        const int32_t* __restrict__ nc = a.node_col + ch.node_begin;
        const double* __restrict__ fac = a.fac;
        if (MODE == PREC_STEP) {
            for (int idx = t; idx < ch.N * BS; idx += kThreads) {
                const int col = nc[idx / BS] + idx % BS;
                a.xt[col] += alpha * a.p[col];
                a.r[col] -= alpha * a.w[col];
            }
            __syncthreads();
        }
        for (int l = 0; l < ch.n_levels; ++l) {
            const ChainLevelDesc L = lv[l];
            const bool last = (L.p == 0);
            const int nsep = last ? 0 : L.N / L.p;
            const double* __restrict__ rec = fac + (size_t)L.data_off * 4 * B2;
            for (int j = t; j <= nsep; j += kThreads) {
                const int lo = last ? 0 : j * L.p;
                const int hi = last ? L.N : min(j * L.p + L.p - 1, L.N);
                if (lo >= hi) continue;
                double y[RMAX][BS];
                // forward substitution
#pragma unroll
                for (int q = 0; q < RMAX; ++q) {
                    const int i = lo + q;
                    if (i < hi) {
                        if (l == 0) {
                            const int col = nc[i];
#pragma unroll
                            for (int c = 0; c < BS; ++c) y[q][c] = a.r[col + c];
                        } else {
#pragma unroll
                            for (int c = 0; c < BS; ++c) y[q][c] = scr[(size_t)(L.vec_off + i) * BS + c];
                        }
                        if (q > 0) matvec_sub<BS>(rec + ((size_t)i * 4 + 0) * B2, y[q - 1], y[q]);
                    }
                }
                // diagonal solve + backward substitution
#pragma unroll
                for (int q = RMAX - 1; q >= 0; --q) {
                    const int i = lo + q;
                    if (i < hi) {
                        double tmp[BS];
#pragma unroll
                        for (int c = 0; c < BS; ++c) tmp[c] = 0.0;
                        const double* __restrict__ Di = rec + ((size_t)i * 4 + 1) * B2;
#pragma unroll
                        for (int c = 0; c < BS; ++c) {
                            double s = 0.0;
#pragma unroll
                            for (int k = 0; k < BS; ++k) s += Di[c * BS + k] * y[q][k];
                            tmp[c] = s;
                        }
                        if (q + 1 < RMAX) {
                            if (i + 1 < hi) matvec_t_sub<BS>(rec + ((size_t)(i + 1) * 4 + 0) * B2, y[q + 1], tmp);
                        }
#pragma unroll
                        for (int c = 0; c < BS; ++c) y[q][c] = tmp[c];
                        if (l == 0) {
                            const int col = nc[i];
#pragma unroll
                            for (int c = 0; c < BS; ++c) a.z[col + c] = tmp[c];
                        } else {
#pragma unroll
                            for (int c = 0; c < BS; ++c) scr[(size_t)(L.vec_off + i) * BS + c] = tmp[c];
                        }
                    }
                }
            }
            __syncthreads();
            if (last) break;
            const ChainLevelDesc Ln = lv[l + 1];
            for (int j = t; j < nsep; j += kThreads) {
                const int s = j * L.p + L.p - 1;
                double v[BS], ym[BS], yp[BS];
                if (l == 0) {
                    const int cs = nc[s], cm = nc[s - 1];
#pragma unroll
                    for (int c = 0; c < BS; ++c) { v[c] = a.r[cs + c]; ym[c] = a.z[cm + c]; }
                } else {
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        v[c] = scr[(size_t)(L.vec_off + s) * BS + c];
                        ym[c] = scr[(size_t)(L.vec_off + s - 1) * BS + c];
                    }
                }
                matvec_sub<BS>(rec + ((size_t)s * 4 + 0) * B2, ym, v);
                if (s + 1 < L.N) {
                    if (l == 0) {
                        const int cp = nc[s + 1];
#pragma unroll
                        for (int c = 0; c < BS; ++c) yp[c] = a.z[cp + c];
                    } else {
#pragma unroll
                        for (int c = 0; c < BS; ++c) yp[c] = scr[(size_t)(L.vec_off + s + 1) * BS + c];
                    }
                    matvec_sub<BS>(rec + ((size_t)s * 4 + 1) * B2, yp, v);
                }
                // the separator's own slot on this level is not read again until
                // back-substitution, so the reduced right-hand side goes to level l+1
#pragma unroll
                for (int c = 0; c < BS; ++c) scr[(size_t)(Ln.vec_off + j) * BS + c] = v[c];
            }
            __syncthreads();
        }
        // back-substitution, coarse to fine
        for (int l = ch.n_levels - 2; l >= 0; --l) {
            const ChainLevelDesc L = lv[l];
            const ChainLevelDesc Ln = lv[l + 1];
            const int nsep = L.N / L.p;
            const double* __restrict__ rec = fac + (size_t)L.data_off * 4 * B2;
            for (int i = t; i < L.N; i += kThreads) {
                const int j = i / L.p;
                double v[BS];
                const bool is_sep = (i % L.p == L.p - 1) && (j < nsep);
                if (is_sep) {
#pragma unroll
                    for (int c = 0; c < BS; ++c) v[c] = scr[(size_t)(Ln.vec_off + j) * BS + c];
                } else {
                    if (l == 0) {
                        const int col = nc[i];
#pragma unroll
                        for (int c = 0; c < BS; ++c) v[c] = a.z[col + c];
                    } else {
#pragma unroll
                        for (int c = 0; c < BS; ++c) v[c] = scr[(size_t)(L.vec_off + i) * BS + c];
                    }
                    if (j >= 1) {
                        double ul[BS];
#pragma unroll
                        for (int c = 0; c < BS; ++c) ul[c] = scr[(size_t)(Ln.vec_off + j - 1) * BS + c];
                        matvec_sub<BS>(rec + ((size_t)i * 4 + 2) * B2, ul, v);
                    }
                    if (j < nsep) {
                        double ur[BS];
#pragma unroll
                        for (int c = 0; c < BS; ++c) ur[c] = scr[(size_t)(Ln.vec_off + j) * BS + c];
                        matvec_sub<BS>(rec + ((size_t)i * 4 + 3) * B2, ur, v);
                    }
                }
                if (l == 0) {
                    const int col = nc[i];
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        a.z[col + c] = v[c];
                        if (MODE == PREC_INIT) a.p[col + c] = v[c];
                        local += a.r[col + c] * v[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < BS; ++c) scr[(size_t)(L.vec_off + i) * BS + c] = v[c];
                }
            }
            __syncthreads();
        }
        if (ch.n_levels == 1) {
            // single-level chain: finish p and the dot product here
            for (int i = t; i < ch.N; i += kThreads) {
                const int col = nc[i];
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    const double zv = a.z[col + c];
                    if (MODE == PREC_INIT) a.p[col + c] = zv;
                    local += a.r[col + c] * zv;
                }
            }
        }
    }
    const double tot = block_sum(local, red);
    if (t == 0) a.rz_out[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// vector updates (grid = K row blocks, so the problem of a block is known)
// ---------------------------------------------------------------------------
struct VecArgs {
    const int32_t* first_row;
    const int32_t* blk_prob;
    const int32_t* done;
    const int32_t* prec_part_ptr;
    const int32_t* kblk_part_ptr;
    const double* rz_new;
    const double* rz_old;
    const double* pw_part;
    const double* z;
    double* p;
    double* xt;
    double* x;
    double alpha_relax;
    int apply_alpha;  // 0: xt already holds the final CG iterate
};

__global__ __launch_bounds__(kThreads) void k_pupdate(VecArgs a) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = a.blk_prob[b];
    if (a.done[prob]) return;
    const double rzn = reduce_partials(a.rz_new, a.prec_part_ptr[prob], a.prec_part_ptr[prob + 1], red);
    const double rzo = reduce_partials(a.rz_old, a.prec_part_ptr[prob], a.prec_part_ptr[prob + 1], red);
    const double beta = rzo > 0.0 ? rzn / rzo : 0.0;
    const int row = a.first_row[b] + threadIdx.x;
    if (row < a.first_row[b + 1]) a.p[row] = a.z[row] + beta * a.p[row];
}

__global__ __launch_bounds__(kThreads) void k_xupdate(VecArgs a) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = a.blk_prob[b];
    if (a.done[prob]) return;
    double alpha = 0.0;
    if (a.apply_alpha) {
        const double rz = reduce_partials(a.rz_old, a.prec_part_ptr[prob], a.prec_part_ptr[prob + 1], red);
        const double pw = reduce_partials(a.pw_part, a.kblk_part_ptr[prob], a.kblk_part_ptr[prob + 1], red);
        alpha = pw > 0.0 ? rz / pw : 0.0;
    }
    const int row = a.first_row[b] + threadIdx.x;
    if (row < a.first_row[b + 1]) {
        const double xt = a.xt[row] + alpha * a.p[row];
        a.xt[row] = xt;
        a.x[row] = a.alpha_relax * xt + (1.0 - a.alpha_relax) * a.x[row];
    }
}

// ---------------------------------------------------------------------------
// cones: one cone per lane
// ---------------------------------------------------------------------------
struct ConeArgs {
    const int32_t* A_ptr;
    const int32_t* A_col;
    const double* A_val;
    const int32_t* cone_row;
    const int32_t* cone_dim;
    const int32_t* cone_type;
    const int32_t* block_first;
    const int32_t* block_prob;
    const int32_t* done;
    const double* rho;
    const double* b;
    const double* xt;   // gathered (xt for the iteration, x for residuals)
    double* s;
    double* y;
    double* u;
    double alpha_relax;
    const double* invE;
    double* pres_part;  // 7 per block
};

__device__ __forceinline__ double a_row_dot(const ConeArgs& a, int i, const double* __restrict__ v) {
    double acc = 0.0;
    for (int k = a.A_ptr[i]; k < a.A_ptr[i + 1]; ++k) acc += a.A_val[k] * v[a.A_col[k]];
    return acc;
}

__global__ __launch_bounds__(kThreads) void k_cone(ConeArgs a) {
    const int b = blockIdx.x;
    const int prob = a.block_prob[b];
    if (a.done[prob]) return;
    const int c = a.block_first[b] + threadIdx.x;
    if (c >= a.block_first[b + 1]) return;
    const int row = a.cone_row[c], dim = a.cone_dim[c];
    const double rho = a.rho[prob], irho = 1.0 / rho, al = a.alpha_relax;
    double t0 = 0.0, nz2 = 0.0;
    for (int k = 0; k < dim; ++k) {
        const int i = row + k;
        const double tt = a_row_dot(a, i, a.xt);
        const double v = al * (a.b[i] - tt) + (1.0 - al) * a.s[i];
        const double wv = v - a.y[i] * irho;
        a.u[i] = v;   // stash v
        a.s[i] = wv;  // stash the point to project
        if (k == 0) t0 = wv; else nz2 += wv * wv;
    }
    double head, tail;  // s+ = (head, tail * w_tail)
    if (a.cone_type[c] == 0) {
        head = 0.0; tail = 0.0;
    } else {
        const double nz = sqrt(nz2);
        if (nz <= t0) { head = t0; tail = 1.0; }
        else if (nz <= -t0) { head = 0.0; tail = 0.0; }
        else { const double m = 0.5 * (t0 + nz); head = m; tail = m / nz; }
    }
    for (int k = 0; k < dim; ++k) {
        const int i = row + k;
        const double sn = (k == 0) ? head : tail * a.s[i];
        const double v = a.u[i];
        const double yn = a.y[i] + rho * (sn - v);
        a.s[i] = sn;
        a.y[i] = yn;
        a.u[i] = rho * (a.b[i] - sn) - yn;
    }
}

// u = rho (b - s) - y   (after a penalty update)
__global__ __launch_bounds__(kThreads) void k_refresh_u(ConeArgs a) {
    const int b = blockIdx.x;
    const int prob = a.block_prob[b];
    const int c = a.block_first[b] + threadIdx.x;
    if (c >= a.block_first[b + 1]) return;
    const int row = a.cone_row[c], dim = a.cone_dim[c];
    const double rho = a.rho[prob];
    for (int k = 0; k < dim; ++k) {
        const int i = row + k;
        a.u[i] = rho * (a.b[i] - a.s[i]) - a.y[i];
    }
}

// primal residual norms; a.xt points at x here
__global__ __launch_bounds__(kThreads) void k_pres(ConeArgs a) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = a.block_prob[b];
    if (a.done[prob]) return;
    const int c = a.block_first[b] + threadIdx.x;
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, sby = 0, bad = 0;
    if (c < a.block_first[b + 1]) {
        const int row = a.cone_row[c], dim = a.cone_dim[c];
        for (int k = 0; k < dim; ++k) {
            const int i = row + k;
            const double tt = a_row_dot(a, i, a.xt);
            const double si = a.s[i];
            const double pr = tt + si - a.b[i];
            const double ie = a.invE[i];
            if (pr != pr) bad = 1.0;
            m0 = fmax(m0, fabs(pr) * ie); m1 = fmax(m1, fabs(tt) * ie); m2 = fmax(m2, fabs(si) * ie);
            m3 = fmax(m3, fabs(pr)); m4 = fmax(m4, fabs(tt)); m5 = fmax(m5, fabs(si));
            sby += a.b[i] * a.y[i];
        }
    }
    bad = block_sum(bad, red);
    m0 = block_max(m0, red); m1 = block_max(m1, red); m2 = block_max(m2, red);
    m3 = block_max(m3, red); m4 = block_max(m4, red); m5 = block_max(m5, red);
    sby = block_sum(sby, red);
    if (threadIdx.x == 0) {
        double* o = a.pres_part + (size_t)b * 8;
        const double nanv = bad > 0.0 ? __builtin_nan("") : 0.0;
        o[0] = m0 + nanv; o[1] = m1; o[2] = m2; o[3] = m3 + nanv; o[4] = m4; o[5] = m5; o[6] = sby; o[7] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// backend
// ---------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T* d = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        release();
        n = count;
        HIP_CHECK(hipMalloc((void**)&d, std::max<size_t>(1, count) * sizeof(T)));
    }
    void upload(const std::vector<T>& h) {
        if (h.size() != n || !d) alloc(h.size());
        if (!h.empty()) HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void zero(hipStream_t st) {
        if (n) HIP_CHECK(hipMemsetAsync(d, 0, n * sizeof(T), st));
    }
    void release() {
        if (d) (void)hipFree(d);
        d = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

struct CsrBufs {
    DevBuf<int32_t> ptr, col, first_row, blk_prob, split;
    DevBuf<double> val;
    int nblocks = 0;
    void upload(const Csr& M, const RowBlocks& rb, const std::vector<int32_t>* sp = nullptr) {
        ptr.upload(M.ptr);
        col.upload(M.col);
        val.upload(M.val);
        first_row.upload(rb.first_row);
        blk_prob.upload(rb.prob);
        if (sp) split.upload(*sp);
        nblocks = rb.nb();
    }
    CsrDev dev() const { return CsrDev{ptr.d, col.d, val.d, first_row.d, blk_prob.d, split.d, nblocks}; }
};

struct HipBackend {
    const HostSystem* H = nullptr;
    score_settings st{};
    hipStream_t stream = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_iters = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    CsrBufs K, G1, G2;
    DevBuf<int32_t> A_ptr, A_col;
    DevBuf<double> A_val;
    DevBuf<double> q, b, invD, invE, rho, fac, dinv;
    DevBuf<int32_t> done, cone_row, cone_dim, cone_type, cone_block_first, cone_block_prob;
    DevBuf<int32_t> node_col, diag_cols, prec_part_ptr, kblk_part_ptr;
    DevBuf<PrecWork> prec_work;
    DevBuf<ChainDesc> chains;
    DevBuf<ChainLevelDesc> levels;
    DevBuf<double> xtu, xy, s, r, z, p, w;
    DevBuf<double> pw_part, rz_part0, rz_part1, rz_meas0, rz_meas1, pres_part, dres_part;
    int cg_iters = 2;
    double* h_pres = nullptr;  // pinned
    double* h_dres = nullptr;
    int n_cone_blocks = 0, n_prec = 0;
    size_t prec_lds = 0;

    ~HipBackend() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (h_pres) (void)hipHostFree(h_pres);
        if (h_dres) (void)hipHostFree(h_dres);
        if (stream) (void)hipStreamDestroy(stream);
    }

    void init(const HostSystem& h, const score_settings& s_) {
        H = &h;
        st = s_;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
        if (st.device < 0 || st.device >= ndev) throw std::runtime_error("score_settings.device out of range");
        HIP_CHECK(hipSetDevice(st.device));
        HIP_CHECK(hipStreamCreate(&stream));
        HIP_CHECK(hipEventCreate(&ev0));
        HIP_CHECK(hipEventCreate(&ev1));
        if (h.bs != 0 && h.bs != 3 && h.bs != 4 && h.bs != 1 && h.bs != 2)
            throw std::runtime_error("unsupported block size");
        K.upload(h.K, h.rbK);
        G1.upload(h.G1, h.rbG1);
        G2.upload(h.G2, h.rbG2, &h.g2_split);
        A_ptr.upload(h.A.ptr); A_col.upload(h.A.col); A_val.upload(h.A.val);
        q.upload(h.q); b.upload(h.b);
        std::vector<double> iD(h.D.size()), iE(h.E.size());
        for (size_t i = 0; i < iD.size(); ++i) iD[i] = 1.0 / h.D[i];
        for (size_t i = 0; i < iE.size(); ++i) iE[i] = 1.0 / h.E[i];
        invD.upload(iD); invE.upload(iE);
        cone_row.upload(h.cone_row); cone_dim.upload(h.cone_dim); cone_type.upload(h.cone_type);
        cone_block_first.upload(h.cone_block_first); cone_block_prob.upload(h.cone_block_prob);
        n_cone_blocks = (int)h.cone_block_prob.size();
        node_col.upload(h.node_col); diag_cols.upload(h.diag_cols);
        prec_part_ptr.upload(h.prec_part_ptr); kblk_part_ptr.upload(h.rbK.part_ptr);
        prec_work.upload(h.prec_work); chains.upload(h.chains); levels.upload(h.levels);
        n_prec = (int)h.prec_work.size();
        prec_lds = (8 + (size_t)h.max_chain_scratch * std::max(1, h.bs)) * sizeof(double);
        if (prec_lds > 64 * 1024) throw std::runtime_error("chain too long for the LDS-resident chain solver");
        xtu.alloc(h.n_tot + h.m_tot); xy.alloc(h.n_tot + h.m_tot); s.alloc(h.m_tot);
        r.alloc(h.n_tot); z.alloc(h.n_tot); p.alloc(h.n_tot); w.alloc(h.n_tot);
        pw_part.alloc(K.nblocks); rz_part0.alloc(n_prec); rz_part1.alloc(n_prec);
        rz_meas0.alloc(n_prec); rz_meas1.alloc(n_prec);
        cg_iters = st.cg_iters;
        pres_part.alloc((size_t)std::max(1, n_cone_blocks) * 8);
        dres_part.alloc((size_t)G2.nblocks * 8);
        HIP_CHECK(hipHostMalloc((void**)&h_pres, pres_part.n * sizeof(double)));
        HIP_CHECK(hipHostMalloc((void**)&h_dres, dres_part.n * sizeof(double)));
        std::vector<int32_t> dz(h.count, 0);
        done.upload(dz);
        upload_rho_values(h);
        reset();
    }

    void upload_rho_values(const HostSystem& h) {
        K.val.upload(h.K.val);
        G1.val.upload(h.G1.val);
        fac.upload(h.fac);
        dinv.upload(h.dinv);
        rho.upload(h.rho);
    }

    ConeArgs cone_args(const double* gathered) {
        ConeArgs a{};
        a.A_ptr = A_ptr.d; a.A_col = A_col.d; a.A_val = A_val.d;
        a.cone_row = cone_row.d; a.cone_dim = cone_dim.d; a.cone_type = cone_type.d;
        a.block_first = cone_block_first.d; a.block_prob = cone_block_prob.d;
        a.done = done.d; a.rho = rho.d; a.b = b.d; a.xt = gathered;
        a.s = s.d; a.y = xy.d + H->n_tot; a.u = xtu.d + H->n_tot;
        a.alpha_relax = st.alpha; a.invE = invE.d; a.pres_part = pres_part.d;
        return a;
    }

    void upload_rho(const HostSystem& h) {
        HIP_CHECK(hipStreamSynchronize(stream));
        upload_rho_values(h);
        if (n_cone_blocks) {
            hipLaunchKernelGGL(k_refresh_u, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    void set_done(const std::vector<int>& d) {
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<int32_t> v(d.begin(), d.end());
        HIP_CHECK(hipMemcpy(done.d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }

    void reset() {
        xtu.zero(stream); xy.zero(stream); s.zero(stream);
        r.zero(stream); z.zero(stream); p.zero(stream); w.zero(stream);
        pw_part.zero(stream); rz_part0.zero(stream); rz_part1.zero(stream);
        rz_meas0.zero(stream); rz_meas1.zero(stream);
        if (n_cone_blocks) {
            hipLaunchKernelGGL(k_refresh_u, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    template <int MODE>
    void launch_prec(const PrecArgs& pa) {
        if (n_prec == 0) return;
        const int bs = H->bs;
        const bool wide = H->radix > 4;
#define SCORE_LAUNCH_PREC(BS, RMAX)                                                                         \
    hipLaunchKernelGGL((k_prec<BS, RMAX, MODE>), dim3(n_prec), dim3(kThreads), prec_lds, stream, pa)
        if (bs <= 1) { if (wide) SCORE_LAUNCH_PREC(1, 7); else SCORE_LAUNCH_PREC(1, 3); }
        else if (bs == 2) { if (wide) SCORE_LAUNCH_PREC(2, 7); else SCORE_LAUNCH_PREC(2, 3); }
        else if (bs == 3) { if (wide) SCORE_LAUNCH_PREC(3, 7); else SCORE_LAUNCH_PREC(3, 3); }
        else { if (wide) SCORE_LAUNCH_PREC(4, 7); else SCORE_LAUNCH_PREC(4, 3); }
#undef SCORE_LAUNCH_PREC
    }

    SpmvArgs spmv_args(const CsrBufs& M, const double* xin) {
        SpmvArgs a{};
        a.M = M.dev(); a.xin = xin; a.done = done.d;
        a.x = xy.d; a.q = q.d; a.r = r.d; a.sigma = H->sigma;
        a.p = p.d; a.w = w.d; a.pw_part = pw_part.d;
        a.invD = invD.d; a.dres_part = dres_part.d;
        return a;
    }

    void launch_kp() {
        hipLaunchKernelGGL(k_spmv<MODE_KP>, dim3(K.nblocks), dim3(kThreads), 0, stream, spmv_args(K, p.d));
    }

    void set_cg_iters(int k) {
        cg_iters = std::max(1, k);
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    }

    // sqrt(r'z_final / r'z_initial) of the last measured KKT solve, per problem
    void cg_reduction(std::vector<double>& out) {
        const HostSystem& h = *H;
        out.assign(h.count, 0.0);
        if (n_prec == 0) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<double> a(n_prec), b2(n_prec);
        HIP_CHECK(hipMemcpy(a.data(), rz_meas0.d, sizeof(double) * n_prec, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(b2.data(), rz_meas1.d, sizeof(double) * n_prec, hipMemcpyDeviceToHost));
        for (int pi = 0; pi < h.count; ++pi) {
            double s0 = 0, s1 = 0;
            for (int i = h.prec_part_ptr[pi]; i < h.prec_part_ptr[pi + 1]; ++i) { s0 += a[i]; s1 += b2[i]; }
            out[pi] = s0 > 0 ? std::sqrt(std::max(0.0, s1) / s0) : 0.0;
        }
    }

    // enqueue one ADMM iteration on `stream`; a measuring iteration additionally
    // leaves r'z before and after the PCG sweep in rz_meas0 / rz_meas1
    void enqueue_iteration(bool measure) {
        hipLaunchKernelGGL(k_spmv<MODE_RHS>, dim3(G1.nblocks), dim3(kThreads), 0, stream, spmv_args(G1, xtu.d));
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = chains.d; pa.levels = levels.d; pa.fac = fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = dinv.d; pa.done = done.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = kblk_part_ptr.d;
        pa.r = r.d; pa.z = z.d; pa.p = p.d; pa.w = w.d; pa.xt = xtu.d;
        pa.pw_part = pw_part.d;
        double* rz_cur = measure ? rz_meas0.d : rz_part0.d;
        pa.rz_in = nullptr; pa.rz_out = rz_cur;
        launch_prec<PREC_INIT>(pa);
        VecArgs va{};
        va.first_row = K.first_row.d; va.blk_prob = K.blk_prob.d; va.done = done.d;
        va.prec_part_ptr = prec_part_ptr.d; va.kblk_part_ptr = kblk_part_ptr.d;
        va.pw_part = pw_part.d; va.z = z.d; va.p = p.d; va.xt = xtu.d; va.x = xy.d;
        va.alpha_relax = st.alpha; va.apply_alpha = 1;
        for (int j = 1; j <= cg_iters; ++j) {
            launch_kp();
            if (j < cg_iters) {
                double* rz_nxt = (rz_cur == rz_part0.d) ? rz_part1.d : rz_part0.d;
                pa.rz_in = rz_cur; pa.rz_out = rz_nxt;
                launch_prec<PREC_STEP>(pa);
                va.rz_new = rz_nxt; va.rz_old = rz_cur;
                hipLaunchKernelGGL(k_pupdate, dim3(K.nblocks), dim3(kThreads), 0, stream, va);
                rz_cur = rz_nxt;
            } else if (measure) {
                pa.rz_in = rz_cur; pa.rz_out = rz_meas1.d;
                launch_prec<PREC_STEP>(pa);  // also applies xt += a p, r -= a w
                va.rz_old = rz_cur; va.rz_new = rz_cur; va.apply_alpha = 0;
                hipLaunchKernelGGL(k_xupdate, dim3(K.nblocks), dim3(kThreads), 0, stream, va);
            } else {
                va.rz_old = rz_cur; va.rz_new = rz_cur;
                hipLaunchKernelGGL(k_xupdate, dim3(K.nblocks), dim3(kThreads), 0, stream, va);
            }
        }
        if (n_cone_blocks)
            hipLaunchKernelGGL(k_cone, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
    }

    void build_graph(int iters) {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        hipGraph_t g = nullptr;
        HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) enqueue_iteration(i == iters - 1);
        HIP_CHECK(hipStreamEndCapture(stream, &g));
        HIP_CHECK(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        graph_iters = iters;
    }

    void run(int iters) {
        if (st.use_graph && iters > 1) {
            if (!graph_exec || graph_iters != iters) build_graph(iters);
            HIP_CHECK(hipGraphLaunch(graph_exec, stream));
        } else {
            for (int i = 0; i < iters; ++i) enqueue_iteration(i == iters - 1);
            HIP_CHECK(hipGetLastError());
        }
    }

    void residuals(std::vector<ResidualSums>& R) {
        const HostSystem& h = *H;
        if (n_cone_blocks) {
            hipLaunchKernelGGL(k_pres, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xy.d));
            HIP_CHECK(hipMemcpyAsync(h_pres, pres_part.d, pres_part.n * sizeof(double), hipMemcpyDeviceToHost, stream));
        }
        hipLaunchKernelGGL(k_spmv<MODE_DRES>, dim3(G2.nblocks), dim3(kThreads), 0, stream, spmv_args(G2, xy.d));
        HIP_CHECK(hipMemcpyAsync(h_dres, dres_part.d, dres_part.n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipGetLastError());
        for (int pi = 0; pi < h.count; ++pi) {
            ResidualSums a;
            for (int bl = h.cone_part_ptr[pi]; bl < h.cone_part_ptr[pi + 1]; ++bl) {
                const double* o = h_pres + (size_t)bl * 8;
                a.rp_u = (o[0] != o[0]) ? o[0] : std::max(a.rp_u, o[0]);
                a.ax_u = std::max(a.ax_u, o[1]); a.s_u = std::max(a.s_u, o[2]);
                a.rp_s = (o[3] != o[3]) ? o[3] : std::max(a.rp_s, o[3]);
                a.ax_s = std::max(a.ax_s, o[4]); a.s_s = std::max(a.s_s, o[5]);
                a.by += o[6];
                if (a.rp_u != a.rp_u) break;
            }
            for (int bl = h.rbG2.part_ptr[pi]; bl < h.rbG2.part_ptr[pi + 1]; ++bl) {
                const double* o = h_dres + (size_t)bl * 8;
                a.rd_u = (o[0] != o[0]) ? o[0] : std::max(a.rd_u, o[0]);
                a.px_u = std::max(a.px_u, o[1]); a.aty_u = std::max(a.aty_u, o[2]);
                a.rd_s = (o[3] != o[3]) ? o[3] : std::max(a.rd_s, o[3]);
                a.px_s = std::max(a.px_s, o[4]); a.aty_s = std::max(a.aty_s, o[5]);
                a.xPx += o[6]; a.qx += o[7];
                if (a.rd_u != a.rd_u) break;
            }
            R[pi] = a;
        }
    }

    void download(const HostSystem& h, double* x, double* y, double* s_out) {
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<double> hx(h.n_tot + h.m_tot), hs(h.m_tot);
        HIP_CHECK(hipMemcpy(hx.data(), xy.d, hx.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (h.m_tot) HIP_CHECK(hipMemcpy(hs.data(), s.d, hs.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (x) for (int64_t i = 0; i < h.n_tot; ++i) x[i] = hx[i] * h.D[i];
        if (y) for (int64_t i = 0; i < h.m_tot; ++i) y[i] = hx[h.n_tot + i] * h.E[i];
        if (s_out) for (int64_t i = 0; i < h.m_tot; ++i) s_out[i] = hs[i] / h.E[i];
    }

    int64_t get_vec(const char* name, double* out, int64_t len) {
        const HostSystem& h = *H;
        const double* src = nullptr;
        int64_t sz = 0;
        bool host = false;
        std::string nm(name);
        if (nm == "xt") { src = xtu.d; sz = h.n_tot; }
        else if (nm == "u") { src = xtu.d + h.n_tot; sz = h.m_tot; }
        else if (nm == "x") { src = xy.d; sz = h.n_tot; }
        else if (nm == "y") { src = xy.d + h.n_tot; sz = h.m_tot; }
        else if (nm == "s") { src = s.d; sz = h.m_tot; }
        else if (nm == "r") { src = r.d; sz = h.n_tot; }
        else if (nm == "z") { src = z.d; sz = h.n_tot; }
        else if (nm == "p") { src = p.d; sz = h.n_tot; }
        else if (nm == "w") { src = w.d; sz = h.n_tot; }
        else if (nm == "D") { src = h.D.data(); sz = h.n_tot; host = true; }
        else if (nm == "E") { src = h.E.data(); sz = h.m_tot; host = true; }
        else if (nm == "Kval") { src = K.val.d; sz = (int64_t)h.K.val.size(); }
        else return -1;
        if (out && len > 0) {
            const size_t bytes = sizeof(double) * (size_t)std::min(len, sz);
            if (host) std::memcpy(out, src, bytes);
            else {
                if (hipStreamSynchronize(stream) != hipSuccess) return -2;
                if (hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -2;
            }
        }
        return sz;
    }

    // roofline probe: average launch duration of the KKT SpMV (w = K p), HIP
    // events on the stream the solver launches on
    void time_kkt(int reps, double* ms, double* bytes) {
        const HostSystem& h = *H;
        std::vector<double> hp(h.n_tot);
        for (int64_t i = 0; i < h.n_tot; ++i) hp[i] = 1.0 + 1e-3 * (double)(i % 7);
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipMemcpy(p.d, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<int32_t> zero(h.count, 0), keep(h.count);
        HIP_CHECK(hipMemcpy(keep.data(), done.d, keep.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(done.d, zero.data(), zero.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        for (int i = 0; i < 10; ++i) launch_kp();
        HIP_CHECK(hipStreamSynchronize(stream));
        HIP_CHECK(hipEventRecord(ev0, stream));
        for (int i = 0; i < reps; ++i) launch_kp();
        HIP_CHECK(hipEventRecord(ev1, stream));
        HIP_CHECK(hipEventSynchronize(ev1));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, ev0, ev1));
        *ms = (double)t / std::max(1, reps);
        double bsum = 0;
        for (double v : h.kkt_bytes) bsum += v;
        *bytes = bsum;
        HIP_CHECK(hipMemcpy(done.d, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
};

}  // namespace

struct score_handle {
    score::Solver<HipBackend> solver;
};

extern "C" {

void score_default_settings(score_settings* s) { score::default_settings(s); }

int score_create_batch(const score_problem* p, int32_t count, const score_settings* s, score_handle** out) {
    try {
        if (!p || !out) throw std::runtime_error("null argument");
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
int score_dims(const score_handle* h, int64_t* n_total, int64_t* m_total, int32_t* count) {
    if (!h) { g_err = "null handle"; return -1; }
    if (n_total) *n_total = h->solver.H.n_tot;
    if (m_total) *m_total = h->solver.H.m_tot;
    if (count) *count = h->solver.H.count;
    return 0;
}
int score_solve(score_handle* h, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        return h->solver.solve(x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_reset(score_handle* h) {
    try {
        if (!h) throw std::runtime_error("null handle");
        h->solver.reset();
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_solve_steps(score_handle* h, int32_t iters, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        return h->solver.steps(iters, x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_time_kkt_apply(score_handle* h, int32_t reps, double* ms, double* bytes) {
    try {
        if (!h) throw std::runtime_error("null handle");
        h->solver.be.time_kkt(reps, ms, bytes);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int64_t score_debug_get(score_handle* h, const char* name, double* out, int64_t len) {
    if (!h || !name) return -1;
    return h->solver.be.get_vec(name, out, len);
}
void score_destroy(score_handle* h) { delete h; }
const char* score_last_error(void) { return g_err.c_str(); }
const char* score_backend(void) { return "hip-gfx950"; }
}
