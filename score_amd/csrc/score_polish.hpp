// score_polish.hpp -- device kernels of the semismooth-Newton polish (see
// score_polish_host.hpp for the mathematics).  These kernels run a handful of
// times per solve; they are written for clarity, the hot kernels (SpMV, chain
// solve) are shared with the ADMM loop.
#pragma once

#include <hip/hip_runtime.h>

#include "score_kernels.hpp"
#include "score_polish_host.hpp"

namespace score {

struct PolishArgs {
    // cones
    int ncones;
    int T;
    const int32_t* cone_row;
    const int32_t* head_col;
    const double* a_abs;
    const double* ck;
    const double* theta;
    const double* xstar;
    const int32_t* A_ptr;
    const int32_t* A_col;
    const double* A_val;
    const double* b;
    // vectors
    const double* u;      // [u | nu] buffer: Newton iterate (head entries zero) followed by nu (m)
    double* nu;           // = u + n_tot
    double* Bbuf;         // T*T per cone
    double* fpart;        // per cone block: partial sum of the cone terms of F
    int64_t n_tot;
    // active-set bookkeeping for the preconditioner (k_newton_cone_b): per cone, bit 0 = active when the chain
    // factors of its problem were last computed, bit 1 = active at the last evaluation; flip_part[b] = cones of
    // the block whose activity differs from bit 0.  The host refactors a problem only when its active set has
    // moved (polish_lockstep) and says so in reref[prob]: the first evaluation after a factorisation then
    // takes the previous evaluation's activity (the point the factors were built at) as the new reference.
    int32_t* act;
    double* flip_part;
    const int32_t* reref;
};

struct HAsmArgs {
    int64_t nnz;
    const double* Pon;
    const int32_t* cptr;
    const int32_t* ccone;
    const int32_t* cab;
    const double* ccoef;
    const double* Bbuf;
    int T2;
    double* Hval;
    const int32_t* dst;  // band view of H (score_band.hpp): entry p is stored at V[dst[p]] as well (null: no view)
    double* V;
    // Jacobi part
    int ndiag;
    const int32_t* diag_pos;
    double* dinv;
    // lock-step batches: entries [ent_part[q], ent_part[q + 1]) belong to problem q = blockIdx.y; a problem with
    // skip[q] != 0 (frozen: converged or stalled) keeps its matrix -- nothing reads it any more
    const int64_t* ent_part;
    const int32_t* skip;
};


// 1-D grid: `count` x `blocks_per_problem` entry blocks (problem q = block / blocks_per_problem: entries
// [ent_part[q], ent_part[q + 1]), 256 per block), then one block per long entry.  A frozen problem (skip[q] != 0) keeps
// its matrix, the long entries included (long_prob: their problems) -- nothing reads it any more.
__global__ __launch_bounds__(kThreads) void k_hassemble(HAsmArgs a, const int32_t* long_entries, const int32_t* long_prob, int blocks_per_problem,
                                                        int count) {
    const int first_long_block = blocks_per_problem * count;
    if ((int)blockIdx.x >= first_long_block) {
        __shared__ double red[8];
        const int li = (int)blockIdx.x - first_long_block;
        if (a.skip && a.skip[long_prob[li]]) return;
        const int64_t pl = long_entries[li];
        const int c0 = a.cptr[pl], c1 = a.cptr[pl + 1];
        double v = 0.0;
        for (int c = c0 + (int)threadIdx.x; c < c1; c += kThreads) v += a.ccoef[c] * a.Bbuf[(size_t)a.ccone[c] * a.T2 + a.cab[c]];
        v = block_sum(v, red);
        if (threadIdx.x == 0) {
            a.Hval[pl] = a.Pon[pl] + v;
            if (a.dst && a.dst[pl] >= 0) a.V[a.dst[pl]] = a.Pon[pl] + v;
        }
        return;
    }
    const int q = (int)blockIdx.x / blocks_per_problem;
    if (a.skip && a.skip[q]) return;
    const int64_t p = a.ent_part[q] + (int64_t)((int)blockIdx.x - q * blocks_per_problem) * kThreads + threadIdx.x;
    if (p >= a.ent_part[q + 1]) return;
    const int c0 = a.cptr[p], c1 = a.cptr[p + 1];
    if (c1 - c0 > kLongContrib) return;  // (a long entry: its own block)
    double v = a.Pon[p];
    const int d = a.dst ? a.dst[p] : -1;
    for (int c = c0; c < c1; ++c) v += a.ccoef[c] * a.Bbuf[(size_t)a.ccone[c] * a.T2 + a.cab[c]];
    a.Hval[p] = v;
    if (d >= 0) a.V[d] = v;
}
// (landmark entries collect a contribution from every cone that touches the landmark -- thousands:
//  strided partial sums + a fixed-order tree instead of one serial lane: the long-entry blocks above)

// ---------------------------------------------------------------------------
// device-side multi-level factorisation of the chains of H (mirror of
// factor_chain_levels in score_host.hpp; one 256-thread workgroup per chain)
// ---------------------------------------------------------------------------
template <int BS>
struct SmallMat {
    static constexpr int B2 = BS * BS;
    __device__ static void mul(const double* A, const double* B, double* C) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < BS; ++k) s += A[i * BS + k] * B[k * BS + j];
                C[i * BS + j] = s;
            }
    }
    __device__ static void mul_bt(const double* A, const double* B, double* C) {  // A B'
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < BS; ++k) s += A[i * BS + k] * B[j * BS + k];
                C[i * BS + j] = s;
            }
    }
    __device__ static void mul_at(const double* A, const double* B, double* C) {  // A' B
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < BS; ++k) s += A[k * BS + i] * B[k * BS + j];
                C[i * BS + j] = s;
            }
    }
    // Gauss-Jordan without pivoting: the blocks are SPD Schur complements
    __device__ static void inv(const double* A, double* Ai) {
        double M[BS][2 * BS];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) { M[i][j] = A[i * BS + j]; M[i][BS + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            const double ip = 1.0 / M[c][c];
#pragma unroll
            for (int j = 0; j < 2 * BS; ++j) M[c][j] *= ip;
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                if (r == c) continue;
                const double f = M[r][c];
#pragma unroll
                for (int j = 0; j < 2 * BS; ++j) M[r][j] -= f * M[c][j];
            }
        }
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < BS; ++j) Ai[i * BS + j] = M[i][BS + j];
    }
};

struct FactorArgs {
    const PrecWork* work;      // only kind == 0 items are processed
    const ChainDesc* chains;
    const ChainLevelDesc* levels;
    const double* Hval;
    const int32_t* pos_diag;   // per level-0 node, bs*bs positions in Hval
    const int32_t* pos_sub;
    double* fac;               // output, same layout as the ADMM factor
    float* fac32;              // optional: the same values as floats, written with them (what k_prec_pre<.., float> streams) -- the
                               // separate rounding launch is then only needed when something reads `fac` itself (the streaming kernel)
    double* work_mat;          // level >= 1 matrices: 2*bs*bs doubles per scratch node
    const int32_t* skip;       // optional, per problem: chains of a frozen problem keep their factors
    // Jacobi work items (columns outside every chain): dinv[e] = 1 / H[diag_pos[e]]
    const int32_t* diag_pos;
    double* dinv;
    // > 0: the Schur complements handed from one level to the next (and the spike blocks a separator
    // needs from the run on its right) stay in dynamic LDS -- lds_wmat doubles for the matrices, the
    // spike exchange behind them -- instead of making a round trip through global memory between the
    // phases of every level (the host enables it when the longest chain of the launch fits)
    int lds_wmat;
};

template <int BS>
__global__ __launch_bounds__(kThreads) void k_factor(FactorArgs a) {
    constexpr int B2 = BS * BS;
    using SM = SmallMat<BS>;
    extern __shared__ __attribute__((aligned(16))) double fl[];
    const bool in_lds = a.lds_wmat > 0;
    const PrecWork wk = a.work[blockIdx.x];
    if (a.skip && a.skip[wk.prob]) return;
    if (wk.kind != 0) {  // Jacobi block: reciprocal diagonal
        for (int e = wk.index + (int)threadIdx.x; e < wk.index + wk.count; e += kThreads) a.dinv[e] = 1.0 / a.Hval[a.diag_pos[e]];
        return;
    }
    const ChainDesc ch = a.chains[wk.index];
    const ChainLevelDesc* lv = a.levels + ch.level_begin;
    const int t = threadIdx.x;
    // block accessors: level 0 gathers from H through the position tables, levels >= 1 read
    // the Schur complements written by the previous level
    auto loadA = [&](int l, const ChainLevelDesc& L, int i, double* out) {
        if (l == 0) {
            // (unconditional loads on clamped positions, selected afterwards: a predicated load compiles to a
            //  branch with its own wait and serialises the gathers)
            const int32_t* pd = a.pos_diag + (size_t)(ch.node_begin + i) * B2;
            int32_t ix[B2];
#pragma unroll
            for (int e = 0; e < B2; ++e) ix[e] = pd[e];
            double v[B2];
#pragma unroll
            for (int e = 0; e < B2; ++e) v[e] = a.Hval[max(ix[e], 0)];
#pragma unroll
            for (int e = 0; e < B2; ++e) out[e] = ix[e] >= 0 ? v[e] : 0.0;
        } else {
            const double* src = in_lds ? fl + ((size_t)(L.vec_off + i) * 2 + 0) * B2
                                       : a.work_mat + ((size_t)(ch.scratch_off + L.vec_off + i) * 2 + 0) * B2;
            for (int e = 0; e < B2; ++e) out[e] = src[e];
        }
    };
    double* vx = fl + a.lds_wmat;  // (in_lds) V of the first node of every run of the current level
    auto loadB = [&](int l, const ChainLevelDesc& L, int i, double* out) {  // T[i, i-1]
        if (l == 0) {
            const int32_t* ps = a.pos_sub + (size_t)(ch.node_begin + i) * B2;
            int32_t ix[B2];
#pragma unroll
            for (int e = 0; e < B2; ++e) ix[e] = ps[e];
            double v[B2];
#pragma unroll
            for (int e = 0; e < B2; ++e) v[e] = a.Hval[max(ix[e], 0)];
#pragma unroll
            for (int e = 0; e < B2; ++e) out[e] = (i > 0 && ix[e] >= 0) ? v[e] : 0.0;
        } else {
            const double* src = in_lds ? fl + ((size_t)(L.vec_off + i) * 2 + 1) * B2
                                       : a.work_mat + ((size_t)(ch.scratch_off + L.vec_off + i) * 2 + 1) * B2;
            for (int e = 0; e < B2; ++e) out[e] = src[e];
        }
    };
    for (int l = 0; l < ch.n_levels; ++l) {
        const ChainLevelDesc L = lv[l];
        const bool last = (L.p == 0);
        const int nsep = L.nsep;
        double* R = a.fac + L.offR;
        double* S = a.fac + L.offS;
        double* Bk = a.fac + L.offB;
        float* R32 = a.fac32 ? a.fac32 + L.offR : nullptr;
        float* S32 = a.fac32 ? a.fac32 + L.offS : nullptr;
        float* B32 = a.fac32 ? a.fac32 + L.offB : nullptr;
        auto Rst = [&](int slot, int q, int j, const double* M) {
            for (int e = 0; e < B2; ++e) {
                const size_t o = ((size_t)(slot * B2 + e) * L.P + q) * L.nruns + j;
                R[o] = M[e];
                if (R32) R32[o] = (float)M[e];
            }
        };
        auto Bst = [&](int slot, int i, const double* M) {
            for (int e = 0; e < B2; ++e) {
                const size_t o = (size_t)(slot * B2 + e) * L.N + i;
                Bk[o] = M[e];
                if (B32) B32[o] = (float)M[e];
            }
        };
        auto Bld = [&](int slot, int i, double* M) {
            for (int e = 0; e < B2; ++e) M[e] = Bk[(size_t)(slot * B2 + e) * L.N + i];
        };
        // ---- runs: block LDL' and the two spikes.  Everything a run needs (its <= 3 diagonal and
        //      sub-diagonal blocks and the block coupling it to the right separator) is requested
        //      first; the factors then stay in registers -- no store -> load round trips ----
        constexpr int RM = kMaxBs - 1;  // nodes per run <= radix - 1 <= 3
        double Vl[B2], Wl[B2];          // spikes of the last node of this lane's (first) run
        // the blocks of this lane's separator (T[s, s-1], T[s, s], T[s+1, s]) are requested together with its
        // run's: at level 0 they are gathers through the position tables, two dependent trips to memory
        double pCl[B2], pAs[B2], pBn[B2];
        const bool pre = !last && t < nsep;
        if (pre) {
            const int s = t * L.p + L.p - 1;
            loadB(l, L, s, pCl);
            loadA(l, L, s, pAs);
            loadB(l, L, min(s + 1, L.N - 1), pBn);
        }
        for (int j = t; j < L.nruns; j += kThreads) {
            const int lo = last ? 0 : j * L.p;
            const int hi = last ? L.N : min(j * L.p + L.p - 1, L.N);
            const int len = hi - lo;
            if (len <= 0) continue;
            const bool hasL = !last && (j >= 1), hasR = !last && (j < nsep);
            double Aq[RM][B2], Bq[RM][B2], Bhi[B2];
#pragma unroll
            for (int q = 0; q < RM; ++q) {
                loadA(l, L, lo + min(q, len - 1), Aq[q]);
                loadB(l, L, lo + min(q, len - 1), Bq[q]);
            }
            loadB(l, L, hasR ? hi : lo, Bhi);
            double Lfq[RM][B2], Diq[RM][B2];
#pragma unroll
            for (int q = 0; q < RM; ++q) {
                double D[B2], T1[B2];
                if (q == 0) {
#pragma unroll
                    for (int e = 0; e < B2; ++e) { D[e] = Aq[0][e]; Lfq[0][e] = 0.0; }
                } else {
                    SM::mul(Bq[q], Diq[q - 1], Lfq[q]);  // Lf = B Dinv_prev
                    SM::mul_bt(Lfq[q], Bq[q], T1);       // Lf B'
#pragma unroll
                    for (int e = 0; e < B2; ++e) D[e] = Aq[q][e] - T1[e];
                }
                SM::inv(D, Diq[q]);
                if (q < len) {
                    Rst(0, q, j, Lfq[q]);
                    Rst(1, q, j, Diq[q]);
                }
            }
            if (last) continue;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                double Y[RM][B2];
                const bool on = (side == 0) ? hasL : hasR;
#pragma unroll
                for (int q = 0; q < RM; ++q)
#pragma unroll
                    for (int e = 0; e < B2; ++e) Y[q][e] = 0.0;
                if (on) {
                    // right-hand side block: V: B[lo] at q = 0;  W: B[hi]' at q = len - 1
                    if (side == 0) {
#pragma unroll
                        for (int e = 0; e < B2; ++e) Y[0][e] = Bq[0][e];
                    } else {
#pragma unroll
                        for (int q = 0; q < RM; ++q)
                            if (q == len - 1) {
#pragma unroll
                                for (int r = 0; r < BS; ++r)
#pragma unroll
                                    for (int c = 0; c < BS; ++c) Y[q][r * BS + c] = Bhi[c * BS + r];
                            }
                    }
#pragma unroll
                    for (int q = 1; q < RM; ++q) {  // forward: a_q -= Lf_q a_{q-1}
                        if (q < len) {
                            double T1[B2];
                            SM::mul(Lfq[q], Y[q - 1], T1);
#pragma unroll
                            for (int e = 0; e < B2; ++e) Y[q][e] -= T1[e];
                        }
                    }
#pragma unroll
                    for (int q = RM - 1; q >= 0; --q) {  // y_q = Dinv_q a_q - Lf_{q+1}' y_{q+1}
                        if (q < len) {
                            double T1[B2];
                            SM::mul(Diq[q], Y[q], T1);
                            if (q + 1 < RM) {
                                if (q + 1 < len) {
                                    double T2[B2];
                                    SM::mul_at(Lfq[q + 1], Y[q + 1], T2);
#pragma unroll
                                    for (int e = 0; e < B2; ++e) T1[e] -= T2[e];
                                }
                            }
#pragma unroll
                            for (int e = 0; e < B2; ++e) Y[q][e] = T1[e];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < RM; ++q) {
                    if (q < len) Bst(side, lo + q, Y[q]);
                    if (in_lds && side == 0 && q == 0) {
#pragma unroll
                        for (int e = 0; e < B2; ++e) vx[(size_t)j * B2 + e] = Y[0][e];
                    }
                    if (q == len - 1 && j == t) {
#pragma unroll
                        for (int e = 0; e < B2; ++e) {
                            if (side == 0) Vl[e] = Y[q][e]; else Wl[e] = Y[q][e];
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (last) break;
        // ---- separators: coupling blocks and the Schur complement = next level's matrix ----
        const ChainLevelDesc Ln = lv[l + 1];
        for (int j = t; j < nsep; j += kThreads) {
            const int s = j * L.p + L.p - 1;
            double Cl[B2], Cr[B2], As[B2], M1[B2], T1[B2];
            if (pre && j == t) {
                for (int e = 0; e < B2; ++e) { Cl[e] = pCl[e]; As[e] = pAs[e]; }
            } else {
                loadB(l, L, s, Cl);  // T[s, s-1]
                loadA(l, L, s, As);
            }
            for (int e = 0; e < B2; ++e) Cr[e] = 0.0;
            if (j == t) {        // W_{s-1}: last node of this lane's own run
                for (int e = 0; e < B2; ++e) M1[e] = Wl[e];
            } else {
                Bld(1, s - 1, M1);
            }
            SM::mul(Cl, M1, T1);
            for (int e = 0; e < B2; ++e) As[e] -= T1[e];
            if (s + 1 < L.N) {
                double Bn[B2];
                if (pre && j == t) {
                    for (int e = 0; e < B2; ++e) Bn[e] = pBn[e];
                } else {
                    loadB(l, L, s + 1, Bn);
                }
                for (int r = 0; r < BS; ++r)
                    for (int c = 0; c < BS; ++c) Cr[r * BS + c] = Bn[c * BS + r];
                if (in_lds) {   // V_{s+1}: first node of the run on the right
                    for (int e = 0; e < B2; ++e) M1[e] = vx[(size_t)(j + 1) * B2 + e];
                } else {
                    Bld(0, s + 1, M1);
                }
                SM::mul(Cr, M1, T1);
                for (int e = 0; e < B2; ++e) As[e] -= T1[e];
            }
            for (int e = 0; e < B2; ++e) {
                S[(size_t)(0 * B2 + e) * nsep + j] = Cl[e];
                S[(size_t)(1 * B2 + e) * nsep + j] = Cr[e];
                if (S32) { S32[(size_t)(0 * B2 + e) * nsep + j] = (float)Cl[e]; S32[(size_t)(1 * B2 + e) * nsep + j] = (float)Cr[e]; }
            }
            double* dstA = in_lds ? fl + ((size_t)(Ln.vec_off + j) * 2 + 0) * B2
                                  : a.work_mat + ((size_t)(ch.scratch_off + Ln.vec_off + j) * 2 + 0) * B2;
            double* dstB = in_lds ? fl + ((size_t)(Ln.vec_off + j) * 2 + 1) * B2
                                  : a.work_mat + ((size_t)(ch.scratch_off + Ln.vec_off + j) * 2 + 1) * B2;
            for (int e = 0; e < B2; ++e) dstA[e] = As[e];
            if (j >= 1) {
                if (j == t) {       // V_{s-1}
                    for (int e = 0; e < B2; ++e) M1[e] = Vl[e];
                } else {
                    Bld(0, s - 1, M1);
                }
                SM::mul(Cl, M1, T1);
                for (int e = 0; e < B2; ++e) dstB[e] = -T1[e];
            } else {
                for (int e = 0; e < B2; ++e) dstB[e] = 0.0;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// small vector kernels of the Newton loop
// ---------------------------------------------------------------------------
struct NewtonVecArgs {
    int64_t n;
    const int32_t* is_head;
    const double* u;
    const double* delta;
    double step;
    double* out;          // trial point / accepted point (head entries stay 0)
    const double* g;      // gradient (= -r)
    double* part;         // per block: partial g'delta
};

__global__ __launch_bounds__(kThreads) void k_newton_trial(NewtonVecArgs a) {
    __shared__ double red[8];
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double gd = 0.0;
    if (i < a.n) {
        const bool head = a.is_head[i] != 0;
        const double d = head ? 0.0 : a.delta[i];
        a.out[i] = head ? 0.0 : a.u[i] + a.step * d;
        gd = a.g[i] * d;
    }
    const double tot = block_sum(gd, red);
    if (threadIdx.x == 0) a.part[blockIdx.x] = tot;
}

struct FinishArgs {
    PolishArgs P;        // P.u = accepted Newton iterate (+ nu after it)
    double* x;           // ADMM x (n)  -- receives u with the head variables filled in
    double* xt;          // ADMM x~
    double* s;
    double* y;
};

// copy the non-head entries
__global__ __launch_bounds__(kThreads) void k_polish_copy_x(NewtonVecArgs a, double* x, double* xt) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < a.n && !a.is_head[i]) { x[i] = a.u[i]; xt[i] = a.u[i]; }
}
// per cone: head variable, s = b - A x, y = (|nu_tail|, nu_tail)
__global__ __launch_bounds__(kThreads) void k_polish_finish(FinishArgs f) {
    const PolishArgs& a = f.P;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= a.ncones) return;
    const int T = a.T;
    const int r0 = a.cone_row[k];
    double rho2 = 0.0, y2 = 0.0;
    for (int c = 0; c < T; ++c) {
        const int r = r0 + 1 + c;
        double acc = a.b[r];
        for (int e = a.A_ptr[r]; e < a.A_ptr[r + 1]; ++e) acc -= a.A_val[e] * a.u[a.A_col[e]];
        f.s[r] = acc;
        rho2 += acc * acc;
        const double yv = a.nu[r];
        f.y[r] = yv;
        y2 += yv * yv;
    }
    const double rho = sqrt(rho2);
    const double xh = fmax(a.xstar[k], rho / a.a_abs[k]);
    const int h = a.head_col[k];
    f.x[h] = xh;
    f.xt[h] = xh;
    f.s[r0] = a.a_abs[k] * xh;  // b_head = 0, A[head, h] = -a_abs
    f.y[r0] = sqrt(y2);
}

// ---------------------------------------------------------------------------
// lock-step variants for a batch of problems (one Newton iteration advances every live problem
// through the same launches): grids follow the per-problem partitions of the ADMM kernels (cone
// blocks, row blocks of H), per-problem scalars come from device arrays, and `skip[prob] != 0`
// freezes a problem (converged, line search finished, PCG converged).
// ---------------------------------------------------------------------------
struct BatchTables {
    const int32_t* cone_block_first;  // per cone block (+1)
    const int32_t* cone_block_prob;
    const int32_t* row_first;         // row blocks of H (+1)
    const int32_t* row_prob;
    const int32_t* skip;              // per problem
    const double* step;               // per problem
};

// Per cone: t = b_tail - A_tail u, multiplier nu = -c max(0,|t|-theta) t/|t|, Hessian block
// B = c [ (1 - theta/rho)(I - uu') + uu' ] on active cones, 0 otherwise.
__global__ __launch_bounds__(kThreads) void k_newton_cone_b(PolishArgs a, BatchTables bt) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = bt.cone_block_prob[b];
    if (bt.skip[prob]) return;
    const int k = bt.cone_block_first[b] + threadIdx.x;
    const bool new_ref = a.reref[prob] != 0;
    double phi = 0.0, flips = 0.0;
    if (k < bt.cone_block_first[b + 1]) {
        const int T = a.T;
        const int r0 = a.cone_row[k];
        double t[kPolishMaxTail];
        double rho2 = 0.0;
        for (int c = 0; c < T; ++c) {
            const int r = r0 + 1 + c;
            double acc = a.b[r];
            for (int e = a.A_ptr[r]; e < a.A_ptr[r + 1]; ++e) acc -= a.A_val[e] * a.u[a.A_col[e]];
            t[c] = acc;
            rho2 += acc * acc;
        }
        const double rho = sqrt(rho2);
        const double th = a.theta[k], ck = a.ck[k];
        const double ex = rho > th ? rho - th : 0.0;
        phi = 0.5 * ck * ex * ex;
        const bool act = ex > 0.0 && rho > 0.0;
        {
            const int32_t old = a.act[k];
            const int32_t ref = new_ref ? (old >> 1) & 1 : old & 1;
            flips = (ref != (int32_t)act) ? 1.0 : 0.0;
            a.act[k] = ref | (act ? 2 : 0);
        }
        const double ir = act ? 1.0 / rho : 0.0;
        a.nu[r0] = 0.0;
        for (int c = 0; c < T; ++c) a.nu[r0 + 1 + c] = act ? -ck * ex * t[c] * ir : 0.0;
        const double w1 = act ? ck * (1.0 - th * ir) : 0.0;
        const double w2 = act ? ck * th * ir : 0.0;
        for (int c = 0; c < T; ++c)
            for (int d = 0; d < T; ++d)
                a.Bbuf[(size_t)k * T * T + c * T + d] = (c == d ? w1 : 0.0) + w2 * (t[c] * ir) * (t[d] * ir);
    }
    block_sum2(phi, flips, red);
    if (threadIdx.x == 0) { a.fpart[b] = phi; a.flip_part[b] = flips; }
}

// out = u + step[prob] * delta on the rows of the live problems; partial g'delta per row block
__global__ __launch_bounds__(kThreads) void k_newton_trial_b(NewtonVecArgs a, BatchTables bt) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = bt.row_prob[b];
    if (bt.skip[prob]) return;
    const double step = bt.step[prob];
    const int64_t i = (int64_t)bt.row_first[b] + threadIdx.x;
    double gd = 0.0;
    if (i < bt.row_first[b + 1]) {
        const bool head = a.is_head[i] != 0;
        const double d = head ? 0.0 : a.delta[i];
        a.out[i] = head ? 0.0 : a.u[i] + step * d;
        gd = a.g[i] * d;
    }
    const double tot = block_sum(gd, red);
    if (threadIdx.x == 0) a.part[b] = tot;
}

// dst[seg] = src[seg] for every flagged problem: segments 2 p (unknowns) and 2 p + 1 (cone rows,
// shifted by n_tot) of the [u | nu] buffers
__global__ __launch_bounds__(kThreads) void k_copy_segments(double* dst, const double* src, const int64_t* seg_begin,
                                                            const int64_t* seg_end, const int32_t* flag) {
    const int sgm = blockIdx.y;
    if (!flag[sgm >> 1]) return;
    const int64_t e = seg_end[sgm];
    for (int64_t i = seg_begin[sgm] + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < e; i += (int64_t)gridDim.x * kThreads)
        dst[i] = src[i];
}

__global__ __launch_bounds__(kThreads) void k_polish_copy_x_b(NewtonVecArgs a, double* x, double* xt, BatchTables bt) {
    const int b = blockIdx.x;
    if (bt.skip[bt.row_prob[b]]) return;
    const int64_t i = (int64_t)bt.row_first[b] + threadIdx.x;
    if (i < bt.row_first[b + 1] && !a.is_head[i]) { x[i] = a.u[i]; xt[i] = a.u[i]; }
}

__global__ __launch_bounds__(kThreads) void k_polish_finish_b(FinishArgs f, BatchTables bt) {
    const PolishArgs& a = f.P;
    const int b = blockIdx.x;
    if (bt.skip[bt.cone_block_prob[b]]) return;
    const int k = bt.cone_block_first[b] + threadIdx.x;
    if (k >= bt.cone_block_first[b + 1]) return;
    const int T = a.T;
    const int r0 = a.cone_row[k];
    double rho2 = 0.0, y2 = 0.0;
    for (int c = 0; c < T; ++c) {
        const int r = r0 + 1 + c;
        double acc = a.b[r];
        for (int e = a.A_ptr[r]; e < a.A_ptr[r + 1]; ++e) acc -= a.A_val[e] * a.u[a.A_col[e]];
        f.s[r] = acc;
        rho2 += acc * acc;
        const double yv = a.nu[r];
        f.y[r] = yv;
        y2 += yv * yv;
    }
    const double rho = sqrt(rho2);
    const double xh = fmax(a.xstar[k], rho / a.a_abs[k]);
    const int h = a.head_col[k];
    f.x[h] = xh;
    f.xt[h] = xh;
    f.s[r0] = a.a_abs[k] * xh;
    f.y[r0] = sqrt(y2);
}

}  // namespace score
