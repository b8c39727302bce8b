// score_prec_wave.hpp -- the split chain preconditioner: one wavefront per chain PART.
//
// Same operator as k_prec / k_prec_pre (z = M^-1 r with the nested-dissection factor of the chain
// part of K, the PCG step folded in), other mapping -- see score_split.hpp for why and for the plan
// a part runs from.  A part is at most 255 nodes: 64 runs on level 0 (one lane each, blocks in
// registers), 16 / 4 / 1 runs on the coarser levels (blocks staged into LDS through a host-built
// gather list).  Every phase fits the wavefront, so phases are separated by wave-level LDS ordering
// only (no s_barrier).  The parts of a chain meet once: each publishes what it contributes to the
// reduced right-hand side of its two top separators (cL, cR) and the residual of the top separator it
// owns, waits for its siblings' slots, solves the 1-3 node top system redundantly and back-substitutes
// its own nodes.
//
// Hand-off (MI355X guide, "Valid forms": sc1 payload stores, the storing wave's s_waitcnt vmcnt(0),
// an sc1 flag store by one lane; the consumer polls the flags with sc1 loads and then reads the
// payload with sc1 loads in the SAME wave).  Flags carry a per-workgroup launch counter kept in global
// memory, so a slot is never mistaken for the previous launch's; the launch is sized by the host to
// be resident at once (<= one workgroup per CU), and the poll is bounded in time: a sibling that never
// arrives poisons the result with NaN (the driver then reports a numerical failure) instead of
// hanging the device.
#pragma once

#include <hip/hip_runtime.h>

#include "score_kernels.hpp"
#include "score_split.hpp"

namespace score {

struct WaveArgs {
    PrecArgs p;                    // vectors, partials, chain tables, gate (work / part_ptr: the SPLIT lists)
    const SplitItem* items;
    const SplitPlan* plans;
    const int32_t* stage_rel;
    double* xbuf;                  // n_slots * kSplitMaxParts * kSplitSlotDoubles
    unsigned int* xflag;           // n_slots * kSplitMaxParts
    unsigned int* epoch;           // one launch counter per work item
    unsigned long long poll_limit; // wall-clock ticks
};

// One workgroup = 4 wavefronts.  A single wavefront can keep only 63 vector loads in flight, and a part
// needs ~250 per lane: all four waves issue the loads (vectors, and every factor block of the part --
// level 0 included -- through the host-built gather list into LDS), then wave 0 alone runs the solve.
constexpr int kWaveThreads = 256;
constexpr int kWaveVec = 3;     // vector entries per lane: (255 + 1) nodes x 3 = 768 = 256 x 3
constexpr int kWaveStage = 13;  // gathered coarse-level factor doubles per lane (<= 3328 per part)
constexpr int kWaveG = 9 * 9;   // level-0 blocks of a lane: run (6 B2), separator (2 B2), Cr of the separator on the part's left

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(kWaveThreads) void k_prec_wave(WaveArgs wa) {
    constexpr int BS = 3, B2 = 9, RMAX = 3;
    constexpr int oCl = 2 * RMAX * B2, oCr = oCl + B2;
    const PrecArgs& a = wa.p;
    KernelStamp stamp(a.tstamp);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[16];
    __shared__ SplitPlan sp;   // the part's plan: read many times by the solve phases (LDS, not a trip to memory each)
    const int tid = threadIdx.x;
    const int t = tid & 63;            // lane; the solve phases run on wave 0 (tid == t)
    const bool solver_wave = tid < 64;
    const PrecWork wk = a.work[blockIdx.x];
    const int prob = wk.prob;
    if (MODE == PREC_INIT && a.gate_init && tid == 0 && (int)blockIdx.x == a.prec_part_ptr[prob]) {
        a.gate_init[prob] = a.done[prob];
        a.gate_used[prob] = 0;
    }
    if (a.done[prob]) return;
    // partial sums for alpha = r'z / p'w (every workgroup of the problem reduces the same partials in the same order)
    double acc_rz = 0.0, acc_pw = 0.0;
    if (MODE == PREC_STEP) {
        const int l0 = a.prec_part_ptr[prob], l1 = a.prec_part_ptr[prob + 1];
        const int k0 = a.kblk_part_ptr[prob], k1 = a.kblk_part_ptr[prob + 1];
        for (int i = l0 + tid; i < l1; i += kWaveThreads) acc_rz += a.rz_in[i];
        for (int i = k0 + tid; i < k1; i += kWaveThreads) acc_pw += a.pw_part[i];
    }
    const double gref = (MODE == PREC_STEP && a.gate_flag && !a.gate_first) ? a.gate_ref[prob] : 0.0;
    double local = 0.0;
    if (wk.kind == 1) {
        // ---- Jacobi block: z = r / diag, PCG step folded in ----
        double alpha = 0.0;
        if (MODE == PREC_STEP) {
            block_sum2(acc_rz, acc_pw, red);
            alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
            if (pcg_gate(a, prob, acc_rz, gref)) return;
        }
        const int e_end = wk.index + wk.count;
        for (int base = wk.index + tid; base < e_end; base += kWaveThreads * 4) {
            int cols[4];
            double rv[4], dv[4], pv[4], wv[4], xv[4], kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cols[u] = a.diag_cols[min(base + u * kWaveThreads, e_end - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rv[u] = a.r_in[cols[u]];
                dv[u] = a.dinv[min(base + u * kWaveThreads, e_end - 1)];
                if (MODE == PREC_STEP) { pv[u] = a.p[cols[u]]; wv[u] = a.w[cols[u]]; xv[u] = a.xt_zero ? 0.0 : a.xt[cols[u]]; kv[u] = a.kx[cols[u]]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u * kWaveThreads < e_end) {
                    double r_ = rv[u];
                    if (MODE == PREC_STEP) {
                        a.xt[cols[u]] = xv[u] + alpha * pv[u];
                        a.kx[cols[u]] = kv[u] + alpha * wv[u];
                        r_ -= alpha * wv[u];
                        a.r[cols[u]] = r_;
                    }
                    const double zv = r_ * dv[u];
                    a.z[cols[u]] = zv;
                    if (MODE == PREC_INIT) a.p[cols[u]] = zv;
                    local += r_ * zv;
                }
            }
        }
        const double tot = block_sum(local, red);
        if (tid == 0) a.rz_out[blockIdx.x] = tot;
        return;
    }
    // ---- chain part ----
    const SplitItem it = wa.items[wk.index];
    const ChainDesc ch = a.chains[it.chain];
    const SplitPlan* __restrict__ plan = wa.plans + it.plan;
    for (int i = tid; i < (int)(sizeof(SplitPlan) / sizeof(int32_t)); i += kWaveThreads)
        reinterpret_cast<int32_t*>(&sp)[i] = reinterpret_cast<const int32_t*>(plan)[i];
    const ChainLevelDesc G0 = a.levels[ch.level_begin];
    const double* __restrict__ fac = a.fac;
    const int64_t fbase = G0.offR;  // the chain's first factor double
    const int L = plan->n_levels, nparts = plan->nparts, part = plan->part;
    const bool has_left = plan->has_left != 0, has_right = plan->has_right != 0;
    const SplitLevel S0 = plan->lv[0];
    const int n0 = S0.n, nr0 = S0.nr, ns0 = nr0 - 1;
    const int n_own = n0 + (has_right ? 1 : 0);   // the part also carries the top separator on its right
    const int NBo = n_own * BS;
    const int col_first = ch.col0 + S0.g0 * BS;   // contiguous chain (the host checks col_stride == bs)
    auto vpos = [&](const SplitLevel& Sx, int i) -> int { return Sx.voff + (i + 4) * BS + ((i + 4) >> 2); };
    double* xch = lds + plan->xch_off;
    double* accL = lds + plan->misc_off;            // contribution to the LEFT top separator's right-hand side
    double* accR = accL + BS;                       // ... to the RIGHT one
    double* xtop = accR + BS;                       // solution of the chain's top separators
    double* xall = xtop + (kSplitMaxParts - 1) * BS;  // siblings' slots
    double* lfac = lds + plan->lfac_off;
    double* lfac0 = lfac;                    // level-0 blocks, [k][lane]
    lfac = lfac + kWaveG * 64;               // coarse levels + top system (plan offsets are relative to this)
    // ---- everything that has to come from memory is requested now, by all four waves.  Every load is
    //      unconditional on a clamped index (a predicated load compiles to a branch and serialises the
    //      memory pipeline); invalid entries are zeroed afterwards ----
    double rv[kWaveVec], wv[kWaveVec], pv[kWaveVec], xv[kWaveVec], kv[kWaveVec];
#pragma unroll
    for (int u = 0; u < kWaveVec; ++u) {
        const int c = col_first + min(tid + u * kWaveThreads, NBo - 1);
        rv[u] = a.r_in[c];
        if (MODE == PREC_STEP) { wv[u] = a.w[c]; pv[u] = a.p[c]; xv[u] = a.xt[c]; kv[u] = a.kx[c]; }
    }
    if (MODE == PREC_STEP && a.xt_zero) {
#pragma unroll
        for (int u = 0; u < kWaveVec; ++u) xv[u] = 0.0;
    }
    // level-0 tile: rows k = wave + 4 u, lane t
    constexpr int kG0 = (kWaveG + 3) / 4;   // 21 rows per wave
    double g0v[kG0];
    const int wave = tid >> 6;
#pragma unroll
    for (int u = 0; u < kG0; ++u) {
        const int k = min(wave + 4 * u, kWaveG - 1);
        const int rb = plan->row_base[k], cm = plan->row_cmax[k];
        const double v = fac[fbase + rb + min(min(t, nr0 - 1), max(cm, 0))];
        g0v[u] = cm >= 0 ? v : 0.0;
    }
    // coarser levels + top system: gather list
    const int n_stage = plan->n_stage;
    double sv[kWaveStage];
    {
        const int32_t* __restrict__ sl = wa.stage_rel + plan->stage_begin;
        int sidx[kWaveStage];
#pragma unroll
        for (int u = 0; u < kWaveStage; ++u) sidx[u] = sl[min(tid + u * kWaveThreads, max(n_stage - 1, 0))];
#pragma unroll
        for (int u = 0; u < kWaveStage; ++u) {
            const double v = fac[fbase + max(sidx[u], 0)];
            sv[u] = sidx[u] >= 0 ? v : 0.0;
        }
    }
    unsigned int epoch = 0;
    if (nparts > 1) epoch = wa.epoch[blockIdx.x] + 1u;
    // ---- alpha, the PCG gate, the vector update ----
    double alpha = 0.0;
    if (MODE == PREC_STEP) {
        block_sum2(acc_rz, acc_pw, red);
        alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
        if (pcg_gate(a, prob, acc_rz, gref)) return;
    }
    double* v0 = lds;  // (positions through vpos(S0, .))
#pragma unroll
    for (int u = 0; u < kWaveVec; ++u) {
        const int idx = tid + u * kWaveThreads;
        if (idx < NBo) {
            const int c = col_first + idx;
            double r_ = rv[u];
            if (MODE == PREC_STEP) {
                r_ -= alpha * wv[u];
                a.r[c] = r_;
                a.xt[c] = xv[u] + alpha * pv[u];
                a.kx[c] = kv[u] + alpha * wv[u];
            }
            rv[u] = r_;
            const int node = idx / BS, comp = idx - node * BS;
            v0[vpos(S0, node) + comp] = r_;   // (node n0 = the owned top separator: its residual, read below)
        }
    }
#pragma unroll
    for (int u = 0; u < kG0; ++u) {
        const int k = wave + 4 * u;
        if (k < kWaveG) lfac0[k * 64 + t] = g0v[u];
    }
#pragma unroll
    for (int u = 0; u < kWaveStage; ++u) {
        const int k = tid + u * kWaveThreads;
        if (k < n_stage) lfac[k] = sv[u];
    }
    if (tid < 2 * BS) accL[tid] = 0.0;
    if (tid < (kSplitMaxParts - 1) * BS) xtop[tid] = 0.0;
    // halos of every level (left: node -1, right: node n) start at zero
    if (tid < BS)
        for (int l = 0; l <= L; ++l) {
            const SplitLevel Sx = plan->lv[l];
            lds[vpos(Sx, -1) + tid] = 0.0;
            if (!(l == 0 && has_right)) lds[vpos(Sx, Sx.n) + tid] = 0.0;
        }
    __syncthreads();
    if (solver_wave && !(a.debug_skip & 4)) {   // (debug_skip: timing experiments only)
    double G[kWaveG];
#pragma unroll
    for (int k = 0; k < kWaveG; ++k) G[k] = lfac0[k * 64 + t];
    // in-place solve with the diagonal block of a run held in (Lf, Dv): y <- T_run^-1 y (first len nodes)
    auto run_solve = [&](double (&y)[RMAX][BS], int len, const double* Lf, const double* Dv) {
#pragma unroll
        for (int q = 1; q < RMAX; ++q) {
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double s_ = y[q][c];
#pragma unroll
                for (int k = 0; k < BS; ++k) s_ -= Lf[q * B2 + c * BS + k] * y[q - 1][k];
                y[q][c] = s_;
            }
        }
#pragma unroll
        for (int q = RMAX - 1; q >= 0; --q) {
            double tmp[BS];
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double s_ = 0.0;
#pragma unroll
                for (int k = 0; k < BS; ++k) s_ += Dv[q * B2 + c * BS + k] * y[q][k];
                tmp[c] = s_;
            }
            if (q + 1 < RMAX) {
                const bool has_next = (q + 1 < len);
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double s_ = 0.0;
#pragma unroll
                    for (int k = 0; k < BS; ++k) s_ += Lf[(q + 1) * B2 + k * BS + c] * y[q + 1][k];
                    tmp[c] = has_next ? tmp[c] - s_ : tmp[c];
                }
            }
#pragma unroll
            for (int c = 0; c < BS; ++c) y[q][c] = tmp[c];
        }
    };
    // ---- level 0: runs (registers) ----
    const int len0 = min(3, n0 - 4 * t);  // nodes of run t (<= 0: none)
    if (t < nr0 && len0 > 0) {
        double y[RMAX][BS];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int p_ = vpos(S0, 4 * t + min(q, len0 - 1));
#pragma unroll
            for (int c = 0; c < BS; ++c) y[q][c] = v0[p_ + c];
        }
        run_solve(y, len0, G, G + RMAX * B2);
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
            if (q < len0) {
                const int p_ = vpos(S0, 4 * t + q);
#pragma unroll
                for (int c = 0; c < BS; ++c) v0[p_ + c] = y[q][c];
            }
    }
    wave_sync();
    // ---- level 0: separators.  Internal ones feed level 1; the top separators collect cL / cR ----
    if (L >= 2 && t < ns0) {
        const int s = 4 * t + 3;
        const bool has_r = (s + 1 < n0);
        const int pv_ = vpos(S0, s), pm = vpos(S0, s - 1), pp = has_r ? vpos(S0, s + 1) : pv_;
        const int dst = vpos(sp.lv[1], t);
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            double acc = v0[pv_ + c];
#pragma unroll
            for (int k = 0; k < BS; ++k) acc -= G[oCl + c * BS + k] * v0[pm + k] + (has_r ? G[oCr + c * BS + k] * v0[pp + k] : 0.0);
            lds[dst + c] = acc;
        }
    }
    if (has_right && t == nr0 - 1 && n0 >= 1) {   // (a part left of a top separator is full: its last run has 3 nodes)
        const int pm = vpos(S0, n0 - 1);
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < BS; ++k) acc += G[oCl + c * BS + k] * v0[pm + k];
            accR[c] = acc;
        }
    }
    if (has_left && t == 0 && n0 >= 1) {
        const int pp = vpos(S0, 0);
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < BS; ++k) acc += G[8 * B2 + c * BS + k] * v0[pp + k];
            accL[c] = acc;
        }
    }
    wave_sync();
    // ---- coarser local levels: blocks from the staged factors ----
    for (int l = 1; l < L; ++l) {
        const SplitLevel Sx = sp.lv[l];
        const int nl_ = Sx.n, nrl = Sx.nr, PP = Sx.P;
        const double* Rl = lfac + Sx.offR;
        const double* Sl = lfac + Sx.offS;
        const int lenl = (PP == 3) ? min(3, nl_ - 4 * t) : nl_;   // (a chain's last level: one run of <= 3 nodes)
        if (t < nrl && lenl > 0) {
            double Lf[RMAX * B2], Dv[RMAX * B2];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int qq = min(q, PP - 1);
#pragma unroll
                for (int e = 0; e < B2; ++e) {
                    Lf[q * B2 + e] = Rl[(e * PP + qq) * nrl + t];
                    Dv[q * B2 + e] = Rl[((B2 + e) * PP + qq) * nrl + t];
                }
            }
            const int first = (PP == 3) ? 4 * t : 0;
            double y[RMAX][BS];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int p_ = vpos(Sx, first + min(q, lenl - 1));
#pragma unroll
                for (int c = 0; c < BS; ++c) y[q][c] = lds[p_ + c];
            }
            run_solve(y, lenl, Lf, Dv);
#pragma unroll
            for (int q = 0; q < RMAX; ++q)
                if (q < lenl) {
                    const int p_ = vpos(Sx, first + q);
#pragma unroll
                    for (int c = 0; c < BS; ++c) lds[p_ + c] = y[q][c];
                }
        }
        wave_sync();
        if (l + 1 < L && t < nrl - 1) {   // internal separators -> next level
            const int s = 4 * t + 3;
            const bool has_r = (s + 1 < nl_);
            const int pv_ = vpos(Sx, s), pm = vpos(Sx, s - 1), pp = has_r ? vpos(Sx, s + 1) : pv_;
            const int dst = vpos(sp.lv[l + 1], t);
            const int so = t + 1;  // staged separator index (0 = the separator on the part's left)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = lds[pv_ + c];
#pragma unroll
                for (int k = 0; k < BS; ++k)
                    acc -= Sl[(c * BS + k) * (nrl + 1) + so] * lds[pm + k] + (has_r ? Sl[(B2 + c * BS + k) * (nrl + 1) + so] * lds[pp + k] : 0.0);
                lds[dst + c] = acc;
            }
        }
        if (has_right && t == 0 && nl_ >= 1) {   // Cl of the top separator (staged index nrl) times the part's last node
            const int pm = vpos(Sx, nl_ - 1);
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < BS; ++k) acc += Sl[(c * BS + k) * (nrl + 1) + nrl] * lds[pm + k];
                accR[c] += acc;
            }
        }
        if (has_left && t == 1 && nl_ >= 1) {    // Cr of the separator on the left (staged index 0) times the first node
            const int pp = vpos(Sx, 0);
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < BS; ++k) acc += Sl[(B2 + c * BS + k) * (nrl + 1) + 0] * lds[pp + k];
                accL[c] += acc;
            }
        }
        wave_sync();
    }
    // ---- the parts of the chain meet: reduced right-hand sides of the top separators ----
    bool poisoned = false;
    if (nparts > 1 && !(a.debug_skip & 64)) {
        double* slot = wa.xbuf + ((size_t)it.slot * kSplitMaxParts + part) * kSplitSlotDoubles;
        if (t < 3 * BS) {
            double val;
            if (t < 2 * BS) val = accL[t];                                   // cL | cR
            else val = has_right ? v0[vpos(S0, n0) + (t - 2 * BS)] : 0.0;    // residual of the owned top separator
            __hip_atomic_store(slot + t, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int* flags = wa.xflag + (size_t)it.slot * kSplitMaxParts;
        if (t == 0) __hip_atomic_store(flags + part, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t_start = (unsigned long long)wall_clock64();
        for (; !(a.debug_skip & 32);) {
            const unsigned int f = (t < nparts) ? __hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all(f == epoch)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((unsigned long long)wall_clock64() - t_start > wa.poll_limit) { poisoned = true; break; }
        }
        if (t < nparts * kSplitSlotDoubles) {
            const double* src = wa.xbuf + ((size_t)it.slot * kSplitMaxParts) * kSplitSlotDoubles + t;
            xall[t] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wave_sync();
        // top system: n_top nodes, one sequential run (blocks staged at top_off: P = n_top, one run)
        const int ntop = sp.n_top;
        if (t == 0) {
            const double* Rt = lfac + sp.top_off;
            double y[RMAX][BS];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int k = min(q, ntop - 1);   // top separator k: between part k and part k + 1
#pragma unroll
                for (int c = 0; c < BS; ++c)
                    y[q][c] = xall[k * kSplitSlotDoubles + 2 * BS + c] - xall[k * kSplitSlotDoubles + BS + c] - xall[(k + 1) * kSplitSlotDoubles + c];
            }
            double Lf[RMAX * B2], Dv[RMAX * B2];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int qq = min(q, ntop - 1);
#pragma unroll
                for (int e = 0; e < B2; ++e) { Lf[q * B2 + e] = Rt[e * ntop + qq]; Dv[q * B2 + e] = Rt[(B2 + e) * ntop + qq]; }
            }
            run_solve(y, ntop, Lf, Dv);
            const double bad = poisoned ? __builtin_nan("") : 0.0;
#pragma unroll
            for (int q = 0; q < RMAX; ++q)
                if (q < ntop) {
#pragma unroll
                    for (int c = 0; c < BS; ++c) xtop[q * BS + c] = y[q][c] + bad;
                }
        }
        wave_sync();
        // the top separators' solution into the halos of every level (a separator takes the coarse solution)
        for (int l = 0; l <= L; ++l) {
            const SplitLevel Sx = sp.lv[l];
            if (t < BS) {
                if (has_left) lds[vpos(Sx, -1) + t] = xtop[(part - 1) * BS + t];
                if (has_right) lds[vpos(Sx, Sx.n) + t] = xtop[part * BS + t];
            }
        }
        wave_sync();
    }
    // ---- back-substitution of the coarser local levels, top down ----
    for (int l = L - 1; l >= 1; --l) {
        const SplitLevel Sx = sp.lv[l];
        const SplitLevel Sn = sp.lv[l + 1];
        if (Sx.P == 3 && t < Sx.n) {   // (a chain's last level has nothing above it)
            const double* Bl = lfac + Sx.offB;
            const int i = t, j = i >> 2;
            const bool is_sep = ((i & 3) == 3);
            const int pvx = vpos(Sx, i), pul = vpos(Sn, j - 1), pur = vpos(Sn, j);
            double v[BS], ul[BS], ur[BS];
#pragma unroll
            for (int c = 0; c < BS; ++c) { v[c] = lds[pvx + c]; ul[c] = lds[pul + c]; ur[c] = lds[pur + c]; }
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = v[c];
#pragma unroll
                for (int k = 0; k < BS; ++k) acc -= Bl[(c * BS + k) * Sx.n + i] * ul[k] + Bl[(B2 + c * BS + k) * Sx.n + i] * ur[k];
                lds[pvx + c] = is_sep ? ur[c] : acc;
            }
        }
        wave_sync();
    }
    // ---- back-substitution of level 0 (no spikes stored: one more run solve against the separator couplings) ----
    if (G0.p != 0) {
        const SplitLevel S1 = sp.lv[1];   // (L == 1: the virtual level, halos only)
        double c_last[BS];
#pragma unroll
        for (int c = 0; c < BS; ++c) c_last[c] = 0.0;
        const bool sep_after = (t < ns0) || (t == nr0 - 1 && has_right);
        if (t < nr0 && sep_after) {
            double xs[BS];
#pragma unroll
            for (int c = 0; c < BS; ++c) xs[c] = lds[vpos(S1, t) + c];   // (t == ns0: the right halo = top separator)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double sl = 0.0, sr = 0.0;
#pragma unroll
                for (int k = 0; k < BS; ++k) { sl += G[oCl + k * BS + c] * xs[k]; sr += G[oCr + k * BS + c] * xs[k]; }
                c_last[c] = sl;
                if (t < ns0) { xch[(t + 1) * BS + c] = sr; v0[vpos(S0, 4 * t + 3) + c] = xs[c]; }
            }
        }
        if (t == 0) {
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double sr = 0.0;
                if (has_left) {
#pragma unroll
                    for (int k = 0; k < BS; ++k) sr += G[8 * B2 + k * BS + c] * xtop[(part - 1) * BS + k];
                }
                xch[c] = sr;
            }
        }
        wave_sync();
        if (t < nr0 && len0 > 0) {
            double y[RMAX][BS];
#pragma unroll
            for (int q = 0; q < RMAX; ++q)
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double v = (q == len0 - 1) ? c_last[c] : 0.0;
                    if (q == 0) v += xch[t * BS + c];
                    y[q][c] = v;
                }
            run_solve(y, len0, G, G + RMAX * B2);
#pragma unroll
            for (int q = 0; q < RMAX; ++q)
                if (q < len0) {
                    const int p_ = vpos(S0, 4 * t + q);
#pragma unroll
                    for (int c = 0; c < BS; ++c) v0[p_ + c] -= y[q][c];
                }
        }
        wave_sync();
    }
    }  // solver_wave
    __syncthreads();
    // ---- z, p (INIT), r'z: all four waves ----
#pragma unroll
    for (int u = 0; u < kWaveVec; ++u) {
        const int idx = tid + u * kWaveThreads;
        if (idx < NBo) {
            const int node = idx / BS, comp = idx - node * BS;
            const double zz = v0[vpos(S0, node) + comp];   // (node n0: the right halo = the owned top separator's solution)
            const int c = col_first + idx;
            a.z[c] = zz;
            if (MODE == PREC_INIT) a.p[c] = zz;
            local += rv[u] * zz;
        }
    }
    const double tot = block_sum(local, red);
    if (tid == 0) {
        a.rz_out[blockIdx.x] = tot;
        if (nparts > 1) wa.epoch[blockIdx.x] = epoch;
    }
}

}  // namespace score
