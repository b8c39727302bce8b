// score_kernels.hpp -- gfx950 device kernels of the SCORE conic solver.
//
// Kernel design notes:
//  * k_spmv: the KKT operator is HBM/L2-bandwidth work (12 B per nonzero,
//    2 flop).  Each 256-thread workgroup owns a tile of <= 256 rows / <= 3072
//    nonzeros, reads values and column indices with fully coalesced loads (12
//    independent loads per lane in flight before the first use), gathers the
//    vector, stages the products in LDS and lets one lane per row add its
//    segment in CSR order, so results do not depend on the launch geometry.
//    Rows longer than 48 nonzeros (landmarks) get a workgroup of their own, an
//    8-way unrolled strided sweep and a shuffle/LDS tree reduction.
//  * dot products are never atomics: every workgroup writes one partial and
//    each consumer workgroup re-reduces the partials of its problem in a fixed
//    order (a few KiB from L2) -- deterministic, and one launch shorter than a
//    separate finalise kernel.
//  * k_prec: direct solve with the per-robot block-tridiagonal part of K,
//    factored on the host as a radix-p nested dissection.  One workgroup per
//    chain; the chain's vector lives in LDS; one lane per run of p-1 nodes with
//    the bs x bs blocks in registers; factor blocks are stored structure-of-
//    arrays so a wavefront's loads are 512 contiguous bytes; O(p log_p N)
//    dependent steps instead of 2N.
//  * k_cone: one cone per lane (cones have 3 or 4 rows; a wavefront per cone
//    would idle 60 lanes), rows kept in registers.
//  * no MFMA anywhere: nothing here is a dense contraction.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>

#include "score_host.hpp"
#include "score_band.hpp"

namespace score {

// Bulk vector stores of the loop's kernels.  SCORE_NT_STORES (build switch, experiments): non-temporal stores -- the
// data leaves the L2 as it is written instead of at the kernel's end (every kernel boundary first writes the
// predecessor's dirty lines back).
#ifdef SCORE_NT_STORES
#define NTS(lhs, val) __builtin_nontemporal_store((val), &(lhs))
#else
#define NTS(lhs, val) ((lhs) = (val))
#endif
// ... and the chain kernels' own (SCORE_NT_PREC: only those)
#if defined(SCORE_NT_STORES) || defined(SCORE_NT_PREC)
#define NTSP(lhs, val) __builtin_nontemporal_store((val), &(lhs))
#else
#define NTSP(lhs, val) ((lhs) = (val))
#endif

constexpr int kThreads = 256;
constexpr int kUnroll = kTileNnz / kThreads;  // 8 nonzeros per lane
constexpr int kLongUnroll = 2;  // nonzeros per lane and trip of a long row (segments hold <= kLongSeg = 512)
constexpr int kPartStride = 12;  // doubles per workgroup in the residual partial arrays

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
// Two sums with one pair of barriers.  `red` >= 8 doubles.
__device__ __forceinline__ void block_sum2(double& v1, double& v2, double* red) {
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v1; red[4 + (threadIdx.x >> 6)] = v2; }
    __syncthreads();
    v1 = (red[0] + red[1]) + (red[2] + red[3]);
    v2 = (red[4] + red[5]) + (red[6] + red[7]);
}
// Fixed-order re-reduction of per-workgroup partials [lo, hi).
__device__ __forceinline__ double reduce_partials(const double* __restrict__ part, int lo, int hi, double* red) {
    double acc = 0.0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kThreads) acc += part[i];
    return block_sum(acc, red);
}
// The same for workgroups of NW wavefronts (the preconditioner uses 8).  `red` >= NW doubles.
template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) tot += red[i];
    return tot;
}
template <int NW>
__device__ __forceinline__ void block_sum2_n(double& v1, double& v2, double* red) {  // `red` >= 2 NW doubles
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v1; red[NW + (threadIdx.x >> 6)] = v2; }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { t1 += red[i]; t2 += red[NW + i]; }
    v1 = t1; v2 = t2;
}
template <int NW>
__device__ __forceinline__ double reduce_partials_n(const double* __restrict__ part, int lo, int hi, double* red) {
    double acc = 0.0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += NW * 64) acc += part[i];
    return block_sum_n<NW>(acc, red);
}

struct CsrDev {
    const int32_t* ptr;
    const int32_t* col;
    const double* val;
    const int32_t* first_row;  // row blocks
    const int32_t* blk_prob;
    const int4* blk_meta;      // per block {first row, end row, first nonzero, end nonzero}: one load
    const int32_t* blk_rs;     // per block: replica stride (rows of replica 0 applied to NR right-hand sides), 0 = plain rows
    const int32_t* split;      // G2 only
    int nblocks;
    // split long rows (kLongSeg, score_host.hpp): per block {first block of the row, segments | segment << 16, first
    // partial slot, row ordinal} (segments == 0: an ordinary block), the published segment sums, the arrival counters
    const int4* blk_long;
    double* long_part;
    unsigned long long* long_cnt;
    // 1: the segments of a split row publish their sums and LEAVE; segment 0 polls the slots until every sum has arrived (a
    //    slot holds kLongSentinel until then), adds them in segment order and resets the slots.  Only for matrices whose whole
    //    launch is resident at once (the host decides by the number of tiles: a polling workgroup must never wait for one that
    //    has not been given a CU yet).  0: the ticket path (the segment that draws the last ticket finishes the row).
    int long_spin;
};
constexpr unsigned long long kLongSentinel = 0x7ff8dead7ff8deadull;  // a NaN no sum can be (payload), both halves equal: filled by a 32-bit memset
constexpr unsigned kLongSentinel32 = 0x7ff8deadu;

// Optional device-side timing of one launch (score_time_iteration): the first lane of every
// workgroup stores the wall clock at entry / exit into its own slot ts[2 b], ts[2 b + 1]; the host
// takes min / max over the workgroups.  This is what a profiler reports as the kernel's duration
// and, unlike HIP events, adds no command between two kernels of the loop (and, unlike atomics
// on one address, no serialisation between workgroups).  ts == nullptr (always, outside that
// probe): one uniform branch.
struct KernelStamp {
    unsigned long long* ts;
    __device__ explicit KernelStamp(unsigned long long* p) : ts(p) {
        if (ts && threadIdx.x == 0) ts[2 * blockIdx.x] = (unsigned long long)wall_clock64();
    }
    __device__ ~KernelStamp() {
        if (ts && threadIdx.x == 0) ts[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
    }
};

// Single-problem handles: the ranges of the per-workgroup partial sums (r'z over the preconditioner's work items, p'w over
// the K row blocks) are launch constants, and the problem index is 0 -- passed by value, the kernels request the
// partials (and the frozen flag, rho, the published step length) in their FIRST trip to memory instead of behind
// block -> problem -> range (two dependent trips in front of every alpha / beta reduction).  on == 0: a batch, the
// kernels look the ranges up.
struct UniRanges {
    int on;
    int l0, l1;  // prec_part_ptr[0], prec_part_ptr[1]
    int k0, k1;  // kblk_part_ptr[0], kblk_part_ptr[1]
};

// Band view of a matrix (score_band.hpp): what k_spmv_band reads next to the source arrays
struct BandDev {
    const double* val;        // V: band tiles (S x 256 doubles, slot-major), diag tiles, remainder entries
    const int32_t* rem_col;   // column of every remainder entry
    const int32_t* rowseg;    // per band tile x 256 lanes: (first << 16) | count of the row's remainder entries
    const int4* meta2;        // per tile {remainder entries, run begin, run end, ordinal | kind << 28}
    int32_t rem0;             // first remainder entry in V
    int32_t bs;
    int32_t offw[kBandMaxS / 2];  // slot pair j: window offset of its first slot, class c in byte c
};

struct SpmvArgs {
    UniRanges uni;
    CsrDev M;
    BandDev B;              // (k_spmv_band only)
    const double* xin;      // gathered vector
    const int32_t* done;
    int xcd_chunk;          // > 0: tiles are dealt to the XCDs in contiguous runs of this many (workgroup i runs on XCD i % 8 and
                            //      takes tile (i % 8) * xcd_chunk + i / 8): the tiles of one problem of a batch, and neighbouring
                            //      tiles of one problem, gather through the same L2.  Grid = 8 * xcd_chunk; 0 = tile i on workgroup i
    int n_tiles;            // (tiles beyond this exit: the grid is rounded up to 8 * xcd_chunk)
    int early_done;         // 1: test the frozen-problem flag before anything else is requested (gated PCG solves queue
                            //    launches that are MEANT to be no-ops once the gate has fired: they must stay cheap);
                            // 0: the flag is requested with everything else and tested before the first write (ADMM loop)
    // replicated rows (blk_rs[b] > 0): replica k gathers xin[col + k * rs_in] (rs_in == 0: the block's own stride)
    // and owns the vector entries row + k * blk_rs[b]
    int rs_in;
    // RHS
    const double* x;
    const double* q;
    const double* kx;       // K xt, carried incrementally by the PCG updates
    double* r;
    double sigma;
    // KP / KPB
    const double* p;        // KP: direction (== xin).  KPB: previous direction
    const double* z;        // KPB: preconditioned residual
    double* p_out;          // KPB: new direction p = z + beta p_old (own rows)
    double* w;
    double* pw_part;
    const double* rz_new;   // KPB: partials of r'z (new, old)
    const double* rz_old;
    const int32_t* prec_part_ptr;
    // RHS with the fused end-of-PCG update of the previous iteration
    int apply_update;       // 1: xt += a p, kx += a w (a = r'z / p'w of the last PCG step), x relaxed
    const double* pfin;     // last PCG direction
    const double* wfin;     // K * pfin
    double* xt_rw;
    double* kx_rw;
    double* x_rw;
    double alpha_relax;
    const double* step_in;  // per problem: step length of the last PCG step (written by k_cone)
    const int32_t* kblk_part_ptr;
    // GRAD
    const int32_t* is_head;
    double* gout;           // gradient (r receives its negative)
    // DRES
    const double* invD;
    double* dres_part;      // 8 per block
    unsigned long long* tstamp;  // see KernelStamp
};

// RHS : r = sigma x - q + M xin - kx                   (M = [0 | A'], xin = [xt ; u], kx = K xt)
// KP  : w = M p, partial p'w                           (M = K)
// KPB : p_new = z + beta p_old (beta from partials), w = M z + beta w_old (= M p_new: one gather per nonzero), partial p_new'w
// DRES: dual residual norms                            (M = [P | A'], xin = [x ; y])
// GRAD: gradient of the reduced (head-eliminated) problem, M = [P | A'], xin = [u ; nu]
enum { MODE_RHS = 0, MODE_KP = 1, MODE_DRES = 2, MODE_KPB = 3, MODE_GRAD = 4 };

constexpr int kMaxRep = 3;

__device__ __forceinline__ void soc_scales(int type, double t0, double nz2, double& head, double& tail) {
    if (type == 0) {  // zero cone
        head = 0.0; tail = 0.0;
        return;
    }
    const double nz = sqrt(nz2);
    if (nz <= t0) { head = t0; tail = 1.0; }
    else if (nz <= -t0) { head = 0.0; tail = 0.0; }
    else { const double m = 0.5 * (t0 + nz); head = m; tail = m / nz; }
}

// The FIRST trip of a tile kernel: everything that depends on the tile index alone -- the tile record(s), the problem, the
// replica stride, the split-row record -- and, for single-problem handles, the frozen flag of problem 0, requested back to back
// and waited for ONCE.  Written as one pinned group because the compiler otherwise sinks each of these uniform loads to its
// first use behind a branch of its own and a tile starts with four dependent trips (record -> problem -> stride -> flag)
// instead of one; the values are made scalar again by readfirstlane (they are uniform: the address is).
struct TileHead {
    int4 meta, m2, lg;
    int prob, rs, dn0;
};
#define SCORE_RFL(x) (x) = __builtin_amdgcn_readfirstlane(x)
template <int NR, bool BAND>
__device__ __forceinline__ TileHead tile_head(const SpmvArgs& a, const int b) {
    TileHead h;
    h.meta = a.M.blk_meta[b];
    h.m2 = BAND ? a.B.meta2[b] : make_int4(0, 0, 0, 0);
    h.lg = a.M.blk_long[b];
    h.prob = a.M.blk_prob[b];
    h.rs = (NR > 1) ? a.M.blk_rs[b] : 0;
    h.dn0 = a.done[0];
    asm volatile("" : "+v"(h.meta.x), "+v"(h.meta.y), "+v"(h.meta.z), "+v"(h.meta.w), "+v"(h.m2.x), "+v"(h.m2.y), "+v"(h.m2.z), "+v"(h.m2.w),
                      "+v"(h.lg.x), "+v"(h.lg.y), "+v"(h.lg.z), "+v"(h.lg.w), "+v"(h.prob), "+v"(h.rs), "+v"(h.dn0));
    SCORE_RFL(h.meta.x); SCORE_RFL(h.meta.y); SCORE_RFL(h.meta.z); SCORE_RFL(h.meta.w);
    SCORE_RFL(h.m2.x); SCORE_RFL(h.m2.y); SCORE_RFL(h.m2.z); SCORE_RFL(h.m2.w);
    SCORE_RFL(h.lg.x); SCORE_RFL(h.lg.y); SCORE_RFL(h.lg.z); SCORE_RFL(h.lg.w);
    SCORE_RFL(h.prob); SCORE_RFL(h.rs); SCORE_RFL(h.dn0);
    if (a.uni.on) h.prob = 0;
    return h;
}

// One tile (row block) of the SpMV with NR right-hand sides per matrix row.  NR == 1: plain CSR rows.  NR > 1: the
// rows of replica 0 of a problem whose operator is I_NR (x) K_row -- the matrix stream (12 B per nonzero) is read
// once, the gathers and the LDS products are per replica; sums are per replica in CSR order, so every replica gets
// exactly what a plain SpMV on its own copy of the rows would give.
template <int MODE, int NR, int UNR = kUnroll>
__device__ __forceinline__ void spmv_tile(const SpmvArgs& a, const int b_in, const int4 meta, const int prob, const int end_ptr,
                                          const int rs_out, double* __restrict__ prod, double* red, int32_t* srow, const int4 lg, const int dn0) {
    static_assert(NR == 1 || (MODE != MODE_DRES && MODE != MODE_GRAD), "residual / gradient modes run on plain rows");
    auto kpad = [](int k) -> int { return k + (k >> 3); };
    constexpr int kPlane = UNR * kThreads + UNR * kThreads / 8;  // one padded plane of products per right-hand side
    const int t = threadIdx.x;
    const int r0 = meta.x, r1 = meta.y, k0 = meta.z, k1 = meta.w;
    const int nn = k1 - k0;
    const int rs_in = (NR > 1) ? (a.rs_in ? a.rs_in : rs_out) : 0;
    const double* __restrict__ val = a.M.val;
    const int32_t* __restrict__ col = a.M.col;
    const double* __restrict__ xin = (MODE == MODE_KPB) ? a.z : a.xin;

    // (gated PCG solves queue launches that are MEANT to be no-ops: they test the flag before anything else is requested)
    if (a.early_done) {
        const int d = a.uni.on ? dn0 : a.done[prob];
        if (d) return;
    }
    // KPB: beta = r'z_new / r'z_old.  The ranges of the partial sums are requested with the second trip (a batch member's:
    // single problems carry them in the arguments), the partials with the third -- load_partials(), called once the tile's
    // matrix entries and gathers are in flight -- and reduced (two barriers) after that.
    double beta = 0.0, acc_n = 0.0, acc_o = 0.0;
    int pi0 = 0, pl1 = 0;
    if (MODE == MODE_KPB) {
        pi0 = (a.uni.on ? a.uni.l0 : a.prec_part_ptr[prob]) + t;
        pl1 = a.uni.on ? a.uni.l1 : a.prec_part_ptr[prob + 1];
    }
    auto load_partials = [&]() {
        if (MODE == MODE_KPB) {  // (the first 256 by their own lanes -- every handle so far --, the rest in a loop; the order of the sums is the loop's)
            if (pi0 < pl1) { acc_n = a.rz_new[pi0]; acc_o = a.rz_old[pi0]; }
            for (int i = pi0 + kThreads; i < pl1; i += kThreads) { acc_n += a.rz_new[i]; acc_o += a.rz_old[i]; }
        }
    };
    // RHS: the step length of the last PCG step was published by the cone kernel
    if (MODE == MODE_RHS && a.apply_update) beta = a.step_in[prob];
    auto finish_beta = [&]() {
        if (MODE == MODE_KPB) {
            block_sum2(acc_n, acc_o, red);
            beta = acc_o > 0.0 ? acc_n / acc_o : 0.0;
        }
    };

    int row = r0 + t;
    bool has_row = false;
    double sum[NR], sum2 = 0.0;  // sum2: A' part (MODE_DRES / MODE_GRAD, NR == 1)
#pragma unroll
    for (int q = 0; q < NR; ++q) sum[q] = 0.0;
    // The frozen-problem flag and the operands of the epilogue (this lane's own vector entries) are requested together
    // with the matrix entries, on a clamped row: every trip to memory the tile needs is then in flight before the
    // first wait -- tile record -> {matrix, row pointers, own entries, flag} -> gathers -- instead of five dependent trips.
    const int dn = a.uni.on ? dn0 : a.done[prob];
    const int my_ptr = a.M.ptr[min(r0 + t, r1)];
    const int nseg = lg.y & 0xffff;  // > 1: this block is one segment of a split long row
    const bool one_long = (r1 - r0 == 1 && (nn > kLongRow || nseg > 1));
    int b = b_in;                    // (the last segment to arrive writes the row's partial sums into the FIRST segment's slot)
    const int ro = one_long ? r0 : min(row, max(r1 - 1, r0));
    double e0[NR], e1[NR], e2[NR], e3[NR], e4[NR], e5[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int o = ro + q * rs_out;
        e0[q] = e1[q] = e2[q] = e3[q] = e4[q] = e5[q] = 0.0;
        if (MODE == MODE_RHS) {
            e0[q] = a.kx[o]; e1[q] = a.x[o]; e2[q] = a.q[o];
            if (a.apply_update) { e3[q] = a.xt_rw[o]; e4[q] = a.pfin[o]; e5[q] = a.wfin[o]; }
        } else if (MODE == MODE_KP) {
            e0[q] = a.p[o];
        } else if (MODE == MODE_KPB) {
            e0[q] = a.w[o]; e1[q] = a.z[o]; e2[q] = a.p[o];
        }
    }

    if (one_long) {
        // one long row: unrolled strided partial sums + tree reduction
        double acc[NR], acc2 = 0.0;
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[q] = 0.0;
        const int split = (MODE == MODE_DRES || MODE == MODE_GRAD) ? a.M.split[r0] : k1;
        // The first trip of the sweep -- for a segment (<= kLongSeg entries) the only one -- is requested BEFORE the partial sums
        // of beta are reduced: entries and gathers on clamped indices by every lane (the lanes beyond the row's end add exact
        // zeros); behind the reduction's two barriers the segments of a KPB product were one dependent trip behind every
        // other tile of the launch.
        const int kb0 = k0 + t;
        {
            int32_t c[kLongUnroll];
            double v[kLongUnroll], g[kLongUnroll][NR];
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
                const int k = min(kb0 + u * kThreads, k1 - 1);
                c[u] = col[k];
                v[u] = val[k];
            }
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
#pragma unroll
                for (int q = 0; q < NR; ++q) g[u][q] = xin[c[u] + q * rs_in];
            }
            load_partials();
            if (dn) return;  // (uniform over the workgroup)
            finish_beta();
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
                const int k = kb0 + u * kThreads;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const double pr = (k < k1) ? v[u] * g[u][q] : 0.0;
                    if ((MODE == MODE_DRES || MODE == MODE_GRAD) && k >= split) acc2 += pr; else acc[q] += pr;
                }
            }
        }
        for (int kb = kb0 + kThreads * kLongUnroll; kb < k1; kb += kThreads * kLongUnroll) {
            int32_t c[kLongUnroll];
            double v[kLongUnroll], g[kLongUnroll][NR];
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
                const int k = min(kb + u * kThreads, k1 - 1);
                c[u] = col[k];
                v[u] = val[k];
            }
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
#pragma unroll
                for (int q = 0; q < NR; ++q) g[u][q] = xin[c[u] + q * rs_in];
            }
#pragma unroll
            for (int u = 0; u < kLongUnroll; ++u) {
                const int k = kb + u * kThreads;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const double pr = (k < k1) ? v[u] * g[u][q] : 0.0;
                    if ((MODE == MODE_DRES || MODE == MODE_GRAD) && k >= split) acc2 += pr; else acc[q] += pr;
                }
            }
        }
        // (two sums per pair of barriers)
        if (MODE == MODE_DRES || MODE == MODE_GRAD) {
            block_sum2(acc[0], acc2, red);
            sum[0] = acc[0]; sum2 = acc2;
        } else if (NR == 1) {
            sum[0] = block_sum(acc[0], red);
        } else {
            block_sum2(acc[0], acc[1], red);
            sum[0] = acc[0]; sum[1] = acc[1];
            if (NR > 2) sum[NR - 1] = block_sum(acc[NR - 1], red);
        }
        const bool spin = nseg > 1 && a.M.long_spin && nseg * kLongVals <= NR * kPlane;
        if (spin) {
            // Polling mode (CsrDev::long_spin: the whole launch is resident): segments 1 .. nseg-1 publish their sums and leave;
            // segment 0 keeps requesting all slots at once until none holds the sentinel any more, adds the sums in SEGMENT
            // order -- as the ticket path does: the result does not depend on who arrives when -- and puts the sentinels back
            // for the next launch.  Against the ticket path a row saves the drain of its stores, the ticket's round trip and
            // a dependent read-back: the split rows are the last workgroups out of every product.
            constexpr int NV = NR + ((MODE == MODE_DRES || MODE == MODE_GRAD) ? 1 : 0);
            const int sgi = lg.y >> 16;
            unsigned long long* allb = reinterpret_cast<unsigned long long*>(a.M.long_part + (size_t)lg.z * kLongVals);
            if (sgi != 0) {
                if (t == 0) {
                    unsigned long long* slot = allb + (size_t)sgi * kLongVals;
#pragma unroll
                    for (int q = 0; q < NR; ++q) __hip_atomic_store(slot + q, (unsigned long long)__double_as_longlong(sum[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (MODE == MODE_DRES || MODE == MODE_GRAD)
                        __hip_atomic_store(slot + NR, (unsigned long long)__double_as_longlong(sum2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;  // (uniform)
            }
            const int total = nseg * kLongVals;
            int ok = 0;
            for (int tries = 0; !ok && tries < (1 << 22); ++tries) {  // (bounded: a launch that could not be resident ends in NaNs, not in a hang)
                ok = 1;
                for (int i = kLongVals + t; i < total; i += kThreads) {
                    if ((i % kLongVals) < NV) {
                        const unsigned long long bits = __hip_atomic_load(allb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (bits == kLongSentinel) ok = 0;
                        else prod[i] = __longlong_as_double((long long)bits);
                    }
                }
                ok = __syncthreads_and(ok);
            }
            for (int i = kLongVals + t; i < total; i += kThreads)
                if ((i % kLongVals) < NV) __hip_atomic_store(allb + i, kLongSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == 0) {
                double tot[kLongVals] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NR; ++q) tot[q] += sum[q];  // (0.0 + own sum: what the ticket path's loop starts with)
                if (MODE == MODE_DRES || MODE == MODE_GRAD) tot[NR] += sum2;
                for (int sg = 1; sg < nseg; ++sg) {
#pragma unroll
                    for (int q = 0; q < NR; ++q) tot[q] += prod[sg * kLongVals + q];
                    if (MODE == MODE_DRES || MODE == MODE_GRAD) tot[NR] += prod[sg * kLongVals + NR];
                }
                const double fail = ok ? 0.0 : __builtin_nan("");
#pragma unroll
                for (int q = 0; q < NR; ++q) sum[q] = tot[q] + fail;
                if (MODE == MODE_DRES || MODE == MODE_GRAD) sum2 = tot[NR] + fail;
            }
            if (MODE == MODE_KP || MODE == MODE_KPB)
                for (int i = 1 + t; i < nseg; i += kThreads) a.pw_part[lg.x + i] = 0.0;
            if (MODE == MODE_DRES || MODE == MODE_GRAD)
                for (int i = kPartStride + t; i < nseg * kPartStride; i += kThreads) a.dres_part[(size_t)lg.x * kPartStride + i] = 0.0;
            b = lg.x;
        } else if (nseg > 1) {
            // A segment publishes its sums (agent-scope stores, drained) and takes a ticket; every launch adds `nseg` to the
            // row's counter, so the segment that draws the last ticket of the launch knows that all sums are out: it adds them
            // in SEGMENT order -- the result does not depend on who arrives when -- and finishes the row; the others leave.
            double* slot = a.M.long_part + (size_t)(lg.z + (lg.y >> 16)) * kLongVals;
            if (t == 0) {
#pragma unroll
                for (int q = 0; q < NR; ++q) __hip_atomic_store(slot + q, sum[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (MODE == MODE_DRES || MODE == MODE_GRAD) __hip_atomic_store(slot + NR, sum2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long ticket = __hip_atomic_fetch_add(a.M.long_cnt + lg.w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                srow[0] = ((int)(ticket % (unsigned long long)nseg) == nseg - 1) ? 1 : 0;
            }
            __syncthreads();
            if (!srow[0]) return;  // (uniform)
            // (every published sum is requested at once -- one trip for the lanes together instead of a dependent load per
            //  segment on lane 0 -- staged in LDS and added there in segment order)
            const double* all = a.M.long_part + (size_t)lg.z * kLongVals;
            const bool staged = nseg * kLongVals <= NR * kPlane;  // (always, short of rows with > 70 000 nonzeros)
            if (staged)
                for (int i = t; i < nseg * kLongVals; i += kThreads)
                    prod[i] = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (t == 0) {
                double tot[kLongVals] = {0.0, 0.0, 0.0, 0.0};
                for (int sg = 0; sg < nseg; ++sg) {
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        tot[q] += staged ? prod[sg * kLongVals + q] : __hip_atomic_load(all + (size_t)sg * kLongVals + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (MODE == MODE_DRES || MODE == MODE_GRAD)
                        tot[NR] += staged ? prod[sg * kLongVals + NR] : __hip_atomic_load(all + (size_t)sg * kLongVals + NR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int q = 0; q < NR; ++q) sum[q] = tot[q];
                if (MODE == MODE_DRES || MODE == MODE_GRAD) sum2 = tot[NR];
            }
            // the partial-sum slots of the other segments' blocks (p'w, residual norms): zero, the row's own go to the first
            if (MODE == MODE_KP || MODE == MODE_KPB)
                for (int i = 1 + t; i < nseg; i += kThreads) a.pw_part[lg.x + i] = 0.0;
            if (MODE == MODE_DRES || MODE == MODE_GRAD)
                for (int i = kPartStride + t; i < nseg * kPartStride; i += kThreads) a.dres_part[(size_t)lg.x * kPartStride + i] = 0.0;
            b = lg.x;
        }
        has_row = (t == 0);
        row = r0;
    } else {
        // Loads are unconditional on clamped indices (a predicated load becomes a branch
        // and serialises the memory pipeline); only the LDS stores are predicated.
        int32_t c[UNR];
        double v[UNR], g[UNR][NR];
        const int klast = max(nn - 1, 0);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = min(t + u * kThreads, klast);
            c[u] = col[k0 + k];
            v[u] = val[k0 + k];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int q = 0; q < NR; ++q) g[u][q] = xin[c[u] + q * rs_in];
        }
        load_partials();
        if (dn) return;  // (uniform over the workgroup; nothing has been written)
        finish_beta();
        // (every lane stores all its products: slots beyond the tile's last nonzero hold copies of the last product and
        //  are never read -- a row adds the slots of its own nonzeros only -- and a predicated store is an exec-mask
        //  branch per product)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = t + u * kThreads;
#pragma unroll
            for (int q = 0; q < NR; ++q) prod[q * kPlane + kpad(k)] = v[u] * g[u][q];
        }
        if (r0 + t <= r1) srow[t] = my_ptr - k0;
        if (t == 0) srow[r1 - r0] = end_ptr - k0;
        __syncthreads();
        if (row < r1) {
            has_row = true;
            const int a0 = srow[t], a1 = srow[t + 1];
            if (MODE == MODE_DRES || MODE == MODE_GRAD) {
                const int sp = a.M.split[row] - k0;
                for (int k = a0; k < sp; ++k) sum[0] += prod[kpad(k)];
                for (int k = sp; k < a1; ++k) sum2 += prod[kpad(k)];
            } else {
                // four slots per trip (clamped reads, the surplus ones add an exact zero -- the order of the additions is
                // the CSR order, as before): one LDS wait per four nonzeros instead of one per nonzero
                for (int k = a0; k < a1; k += 4) {
                    const int k1_ = min(k + 1, a1 - 1), k2_ = min(k + 2, a1 - 1), k3_ = min(k + 3, a1 - 1);
                    double p0[NR], p1[NR], p2[NR], p3[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        p0[q] = prod[q * kPlane + kpad(k)];
                        p1[q] = prod[q * kPlane + kpad(k1_)];
                        p2[q] = prod[q * kPlane + kpad(k2_)];
                        p3[q] = prod[q * kPlane + kpad(k3_)];
                    }
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        sum[q] += p0[q];
                        sum[q] += (k + 1 < a1) ? p1[q] : 0.0;
                        sum[q] += (k + 2 < a1) ? p2[q] : 0.0;
                        sum[q] += (k + 3 < a1) ? p3[q] : 0.0;
                    }
                }
            }
        }
    }

    if (MODE == MODE_RHS) {
        if (has_row) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int o = row + q * rs_out;
                double kxv = e0[q];
                double xv = e1[q];
                if (a.apply_update) {
                    const double xt = e3[q] + beta * e4[q];
                    kxv += beta * e5[q];
                    xv = a.alpha_relax * xt + (1.0 - a.alpha_relax) * xv;
                    NTS(a.xt_rw[o], xt);
                    NTS(a.kx_rw[o], kxv);
                    NTS(a.x_rw[o], xv);
                }
                NTS(a.r[o], a.sigma * xv - e2[q] + sum[q] - kxv);
            }
        }
    } else if (MODE == MODE_KP || MODE == MODE_KPB) {
        double local = 0.0;
        if (has_row) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int o = row + q * rs_out;
                double pi, sq = sum[q];
                if (MODE == MODE_KPB) {
                    // K (z + beta p) = K z + beta w_old: one gather (z) per nonzero instead of two; w_old = K p_old is this
                    // row's own entry.  (The first product of a solve is always the direct one, MODE_KP.)
                    sq += beta * e0[q];
                    pi = e1[q] + beta * e2[q];
                    NTS(a.p_out[o], pi);
                } else {
                    pi = e0[q];
                }
                NTS(a.w[o], sq);
                local += pi * sq;
            }
        }
        const double tot = block_sum(local, red);
        if (t == 0) a.pw_part[b] = tot;
    } else if (MODE == MODE_GRAD) {  // sum = (P u)_i, sum2 = (A'nu)_i ; xin = [u ; nu]
        double m0 = 0, m1 = 0, s0 = 0, bad = 0;
        if (has_row) {
            const bool head = a.is_head[row] != 0;
            const double qi = a.q[row];
            const double g = head ? 0.0 : sum[0] + qi + sum2;
            if (g != g) bad = 1.0;
            a.gout[row] = g;
            NTS(a.r[row], -g);
            m0 = fabs(g) * a.invD[row];
            m1 = fabs(g);
            const double xi = xin[row];
            s0 = head ? 0.0 : xi * (0.5 * sum[0] + qi);
        }
        bad = block_sum(bad, red);
        m0 = block_max(m0, red); m1 = block_max(m1, red);
        s0 = block_sum(s0, red);
        if (t == 0) {
            double* o = a.dres_part + (size_t)b * kPartStride;
            const double nanv = bad > 0.0 ? __builtin_nan("") : 0.0;
            o[0] = m0 + nanv; o[1] = m1; o[2] = s0;
        }
    } else {  // MODE_DRES: sum = (P x)_i, sum2 = (A'y)_i ; xin = [x ; y]
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, s0 = 0, s1 = 0, s2 = 0, bad = 0;
        if (has_row) {
            const double qi = a.q[row];
            const double dr = sum[0] + qi + sum2;
            const double id = a.invD[row];
            if (dr != dr) bad = 1.0;
            m0 = fabs(dr) * id; m1 = fabs(sum[0]) * id; m2 = fabs(sum2) * id;
            m3 = fabs(dr); m4 = fabs(sum[0]); m5 = fabs(sum2);
            const double xi = xin[row];
            s0 = xi * sum[0];
            s1 = qi * xi;
            s2 = xi * dr;  // x'r_d: a cancellation-free piece of the duality gap
        }
        bad = block_sum(bad, red);
        m0 = block_max(m0, red); m1 = block_max(m1, red); m2 = block_max(m2, red);
        m3 = block_max(m3, red); m4 = block_max(m4, red); m5 = block_max(m5, red);
        s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
        if (t == 0) {
            double* o = a.dres_part + (size_t)b * kPartStride;
            const double nanv = bad > 0.0 ? __builtin_nan("") : 0.0;
            o[0] = m0 + nanv; o[1] = m1; o[2] = m2; o[3] = m3 + nanv; o[4] = m4; o[5] = m5; o[6] = s0; o[7] = s1; o[8] = s2;
        }
    }
}

// NR: right-hand sides per row of the replicated blocks of this launch (1: every block holds plain rows).
// UNR: nonzeros per lane of a tile (tiles of UNR * 256 nonzeros, HostSystem::tile_nnz): a single replicated problem
// has half the matrix of a general one -- smaller tiles keep every CU busy.
template <int MODE, int NR = 1, int UNR = kUnroll>
__global__ __launch_bounds__(kThreads) void k_spmv(SpmvArgs a) {
    KernelStamp stamp(a.tstamp);
    // products of the tile, one padded plane per right-hand side: the row sums read consecutive 8-entry segments from
    // consecutive lanes (stride 8 doubles = 16 banks -> 16-way conflicts unpadded, 2-way with stride 9)
    __shared__ double prod[NR * (UNR * kThreads + UNR * kThreads / 8)];
    __shared__ double red[8];
    __shared__ int32_t srow[kRowsPerBlock + 1];  // row pointers of the tile, relative to k0
    const int b = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (b >= a.n_tiles) return;
    const int t = threadIdx.x;
    // first trip: the tile record, its problem, stride, split-row record and (single problem) the frozen flag -- tile_head
    const TileHead h = tile_head<NR, false>(a, b);
    const int end_ptr = (t == 0) ? h.meta.w : 0;
    if (NR > 1 && h.rs > 0) spmv_tile<MODE, NR, UNR>(a, b, h.meta, h.prob, end_ptr, h.rs, prod, red, srow, h.lg, h.dn0);
    else spmv_tile<MODE, 1, UNR>(a, b, h.meta, h.prob, end_ptr, 0, prod, red, srow, h.lg, h.dn0);
}

// ---------------------------------------------------------------------------
// band view (score_band.hpp): K p for matrices whose rows follow the pose chains
// ---------------------------------------------------------------------------
// KP / KPB epilogue of one row with NR right-hand sides: stores w (and, KPB, the new direction), returns p'w of the row
template <int MODE, int NR>
__device__ __forceinline__ double kp_row_finish(const SpmvArgs& a, const int row, const int rs_out, const double beta, const double (&sum)[NR],
                                                const double (&e0)[NR], const double (&e1)[NR], const double (&e2)[NR]) {
    double local = 0.0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int o = row + q * rs_out;
        double pi, sq = sum[q];
        if (MODE == MODE_KPB) {  // K (z + beta p) = K z + beta w_old, see spmv_tile
            sq += beta * e0[q];
            pi = e1[q] + beta * e2[q];
            NTS(a.p_out[o], pi);
        } else {
            pi = e0[q];
        }
        NTS(a.w[o], sq);
        local += pi * sq;
    }
    return local;
}

// One band tile: every lane owns a row of the tile.  Nothing it requests depends on anything but the tile record:
// NP value pairs (pair-major: lane t reads the 16 bytes at V[base + j * 512 + 2 t], 1 KiB contiguous per wavefront and
// pair), the NP operand pairs of its window per right-hand side (16-byte loads; neighbouring lanes read neighbouring or
// equal addresses), the row's (first, count) remainder word, the tile's remainder entries (coalesced) -- then ONE
// dependent trip for the remainder's gathers (a tenth of the nonzeros), whose products go to LDS and are added row by
// row in CSR order.  A pair whose base falls outside the run [lo, hi) is loaded from the clamped base and put right by a
// select (the slots of positions outside the run hold zeros and must meet finite operands from the run itself: a
// neighbouring problem's entries may be NaN).
// LDSW (round 5): the operand window of the tile -- the positions [r0 - 8, r0 + 264) of the operand, clamped to the run, per
// right-hand side: ~2 KiB -- is loaded ONCE per tile, coalesced, into LDS, and a lane takes its NP pairs from there
// (ds_read) instead of issuing NP * NR sixteen-byte global loads whose addresses its neighbours request as well: two thirds
// of the tile's vector-memory instructions gone; same values, same order of additions.  kBandWin = positions per window.
constexpr int kBandWin = kBandLanes + 16;
template <int MODE, int NR, int NP, bool LDSW = false>
__device__ __forceinline__ void band_tile(const SpmvArgs& a, const int b, const int4 meta, const int4 m2, const int prob, const int rs_out,
                                          const int dn0, double* __restrict__ prod, double* red, double* __restrict__ win = nullptr) {
    static_assert(MODE == MODE_KP || MODE == MODE_KPB, "band tiles serve the K / H products");
    constexpr int kPlane = kBandRemMax + kBandRemMax / 8;
    auto kpad = [](int k) -> int { return k + (k >> 3); };
    const int t = threadIdx.x;
    const int r0 = meta.x, r1 = meta.y;
    const int rem_cnt = m2.x, lo = m2.y, hi = m2.z, ord = m2.w & 0x0fffffff;
    const int rs_in = (NR > 1) ? (a.rs_in ? a.rs_in : rs_out) : 0;
    const double* __restrict__ xin = (MODE == MODE_KPB) ? a.z : a.xin;
    double beta = 0.0, acc_n = 0.0, acc_o = 0.0;
    // (gated PCG solves queue launches that are MEANT to be no-ops: they test the flag before anything else is requested)
    if (a.early_done) {
        const int d = a.uni.on ? dn0 : a.done[prob];
        if (d) return;
    }
    // Second trip, requested in this order: the remainder's columns FIRST (the third trip -- their gathers -- waits for them
    // alone: loads return in order), the ranges of the r'z partials and the frozen flag of a batch member, then the bulk.
    constexpr int RU = kBandRemMax / kThreads;
    int32_t c[RU];
    double rv[RU], gx[RU][NR];
    const int wave0 = __builtin_amdgcn_readfirstlane(t & ~63);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        c[u] = lo; rv[u] = 0.0;
        if (u * kThreads + wave0 < rem_cnt) {
            const int k = meta.w + u * kThreads + t;
            c[u] = a.B.rem_col[k];
            rv[u] = a.B.val[a.B.rem0 + k];
        }
    }
    int pi0 = 0, pl1 = 0;
    if (MODE == MODE_KPB) {
        pi0 = (a.uni.on ? a.uni.l0 : a.prec_part_ptr[prob]) + t;
        pl1 = a.uni.on ? a.uni.l1 : a.prec_part_ptr[prob + 1];
    }
    const int dn = a.uni.on ? dn0 : a.done[prob];
    const int row = r0 + t;
    const bool has_row = row < r1;
    const int ro = min(row, r1 - 1);
    double e0[NR], e1[NR], e2[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int o = ro + q * rs_out;
        e0[q] = e1[q] = e2[q] = 0.0;
        if (MODE == MODE_KP) e0[q] = a.p[o];
        else { e0[q] = a.w[o]; e1[q] = a.z[o]; e2[q] = a.p[o]; }
    }
    const int bs = a.B.bs;
    const int cls = (bs == 3) ? t % 3 : (t & (bs - 1));
    const int nb = row - cls;
    const double2* __restrict__ bv = reinterpret_cast<const double2*>(a.B.val + meta.z) + t;
    double2 v[NP], g[NP][NR];
    int dsh[NP], cbs[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = bv[j * kBandLanes];
    const int w0 = max(lo, r0 - 8);  // first position of the staged window
    double wl[NR], wx[NR];
    if (LDSW) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {  // (positions beyond the run are never multiplied by a nonzero: any finite entry of the run will do)
            wl[q] = xin[min(w0 + t, hi - 1) + q * rs_in];
            wx[q] = (t < kBandWin - kBandLanes) ? xin[min(w0 + kBandLanes + t, hi - 1) + q * rs_in] : 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int base = nb + __builtin_amdgcn_sbfe(a.B.offw[j], 8 * cls, 8);
        const int cb = min(max(base, lo), hi - 2);
        dsh[j] = base - cb;
        cbs[j] = cb;
        if (!LDSW) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {  // (8-byte aligned 16-byte load)
                const double* src = xin + cb + q * rs_in;
                double2 ld;
                __builtin_memcpy(&ld, src, sizeof(double2));
                g[j][q] = ld;
            }
        }
    }
    // the remainder: entries [0, rem_cnt) of the tile, 64 per wavefront and trip (whole wavefronts beyond the count skip)
    const int seg = a.B.rowseg[ord * kBandLanes + t];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        if (u * kThreads + wave0 < rem_cnt) {
#pragma unroll
            for (int q = 0; q < NR; ++q) gx[u][q] = xin[c[u] + q * rs_in];
        } else {
#pragma unroll
            for (int q = 0; q < NR; ++q) gx[u][q] = 0.0;
        }
    }
    if (MODE == MODE_KPB) {  // the partials of r'z: the first 256 by their own lanes (every handle so far), the rest in a loop
        if (pi0 < pl1) { acc_n = a.rz_new[pi0]; acc_o = a.rz_old[pi0]; }
        for (int i = pi0 + kThreads; i < pl1; i += kThreads) { acc_n += a.rz_new[i]; acc_o += a.rz_old[i]; }
    }
    if (dn) return;  // (uniform over the workgroup; nothing has been written)
    if (LDSW) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            win[q * kBandWin + t] = wl[q];
            if (t < kBandWin - kBandLanes) win[q * kBandWin + kBandLanes + t] = wx[q];
        }
    }
    if (MODE == MODE_KPB) {
        block_sum2(acc_n, acc_o, red);  // (its barriers also publish the window)
        beta = acc_o > 0.0 ? acc_n / acc_o : 0.0;
    } else if (LDSW) {
        __syncthreads();
    }
    if (LDSW) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // (a row whose pair lies outside the staged window -- a surplus lane behind the tile's last row -- reads a clamped slot)
            const int o = min(max(cbs[j] - w0, 0), kBandWin - 2);
#pragma unroll
            for (int q = 0; q < NR; ++q) g[j][q] = make_double2(win[q * kBandWin + o], win[q * kBandWin + o + 1]);
        }
    }
    double sum[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) sum[q] = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const double x0 = dsh[j] > 0 ? g[j][q].y : g[j][q].x;
            const double x1 = dsh[j] < 0 ? g[j][q].x : g[j][q].y;
            sum[q] += v[j].x * x0;
            sum[q] += v[j].y * x1;
        }
    }
    if (rem_cnt > 0) {  // (uniform)
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (u * kThreads + wave0 < rem_cnt) {
                const int k = t + u * kThreads;
#pragma unroll
                for (int q = 0; q < NR; ++q) prod[q * kPlane + kpad(k)] = rv[u] * gx[u][q];
            }
        }
        __syncthreads();
        if (has_row) {
            // four slots per trip (clamped reads, exact zeros for the surplus: the additions keep the CSR order)
            const int k0 = seg >> 16, k1 = k0 + (seg & 0xffff);
            for (int k = k0; k < k1; k += 4) {
                const int ka = min(k + 1, k1 - 1), kb = min(k + 2, k1 - 1), kc = min(k + 3, k1 - 1);
                double p0[NR], p1[NR], p2[NR], p3[NR];
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    p0[q] = prod[q * kPlane + kpad(k)];
                    p1[q] = prod[q * kPlane + kpad(ka)];
                    p2[q] = prod[q * kPlane + kpad(kb)];
                    p3[q] = prod[q * kPlane + kpad(kc)];
                }
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    sum[q] += p0[q];
                    sum[q] += (k + 1 < k1) ? p1[q] : 0.0;
                    sum[q] += (k + 2 < k1) ? p2[q] : 0.0;
                    sum[q] += (k + 3 < k1) ? p3[q] : 0.0;
                }
            }
        }
    }
    double local = 0.0;
    if (has_row) local = kp_row_finish<MODE, NR>(a, row, rs_out, beta, sum, e0, e1, e2);
    const double tot = block_sum(local, red);
    if (t == 0) a.pw_part[b] = tot;
}

// One diag tile: plain rows whose only entry is their diagonal, kBandDiagRows / 256 per lane, everything requested at once.
template <int MODE>
__device__ __forceinline__ void diag_tile(const SpmvArgs& a, const int b, const int4 meta, const int prob, const int dn0, double* red) {
    constexpr int DU = kBandDiagRows / kThreads;
    const int t = threadIdx.x;
    const int r0 = meta.x, r1 = meta.y;
    double beta = 0.0, acc_n = 0.0, acc_o = 0.0;
    if (a.early_done) {
        const int d = a.uni.on ? dn0 : a.done[prob];
        if (d) return;
    }
    int pi0 = 0, pl1 = 0;
    if (MODE == MODE_KPB) {
        pi0 = (a.uni.on ? a.uni.l0 : a.prec_part_ptr[prob]) + t;
        pl1 = a.uni.on ? a.uni.l1 : a.prec_part_ptr[prob + 1];
    }
    const int dn = a.uni.on ? dn0 : a.done[prob];
    double v[DU], x[DU][1], e0[DU][1], e1[DU][1], e2[DU][1];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
        const int ro = min(r0 + u * kThreads + t, r1 - 1);
        v[u] = a.B.val[meta.z + (ro - r0)];
        e0[u][0] = e1[u][0] = e2[u][0] = 0.0;
        // (the operand of a diagonal row is the row's own entry of the vector the epilogue reads anyway)
        if (MODE == MODE_KP) { e0[u][0] = a.p[ro]; x[u][0] = (a.p == a.xin) ? e0[u][0] : a.xin[ro]; }
        else { e0[u][0] = a.w[ro]; e1[u][0] = a.z[ro]; e2[u][0] = a.p[ro]; x[u][0] = e1[u][0]; }
    }
    if (MODE == MODE_KPB) {
        if (pi0 < pl1) { acc_n = a.rz_new[pi0]; acc_o = a.rz_old[pi0]; }
        for (int i = pi0 + kThreads; i < pl1; i += kThreads) { acc_n += a.rz_new[i]; acc_o += a.rz_old[i]; }
    }
    if (dn) return;  // (uniform; nothing has been written)
    if (MODE == MODE_KPB) {
        block_sum2(acc_n, acc_o, red);
        beta = acc_o > 0.0 ? acc_n / acc_o : 0.0;
    }
    double local = 0.0;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
        const int row = r0 + u * kThreads + t;
        if (row < r1) {
            const double sum[1] = {v[u] * x[u][0]};
            local += kp_row_finish<MODE, 1>(a, row, 0, beta, sum, e0[u], e1[u], e2[u]);
        }
    }
    const double tot = block_sum(local, red);
    if (t == 0) a.pw_part[b] = tot;
}

// The SpMV over a band view: CSR tiles (the landmark rows; 2 nonzero slots per lane), band tiles and diag tiles in one
// launch, a uniform branch per workgroup.  NR as in k_spmv (the band tiles of a replicated matrix are all replicated:
// chains live in the replicas, never in the tail); NP = value-slot pairs per band row (4, 5 or 6).
template <int MODE, int NR, int NP, bool LDSW = false>
__device__ __forceinline__ void spmv_band_body(const SpmvArgs& a) {
    KernelStamp stamp(a.tstamp);
    constexpr int UNR = kBandCsrNnz / kThreads;
    constexpr int kCsrPlane = UNR * kThreads + UNR * kThreads / 8;
    constexpr int kRemPlane = kBandRemMax + kBandRemMax / 8;
    __shared__ double prod[NR * (kCsrPlane > kRemPlane ? kCsrPlane : kRemPlane)];
    __shared__ double win[LDSW ? NR * kBandWin : 1];
    __shared__ double red[8];
    __shared__ int32_t srow[kRowsPerBlock + 1];
    const int b = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (b >= a.n_tiles) return;
    const int t = threadIdx.x;
    const TileHead h = tile_head<NR, true>(a, b);
    const int4 meta = h.meta, m2 = h.m2;
    const int prob = h.prob, rs = h.rs;
    const int kind = (int)((unsigned)m2.w >> 28);
    if (kind == BAND_KIND_BAND) {
        band_tile<MODE, NR, NP, LDSW>(a, b, meta, m2, prob, rs, h.dn0, prod, red, win);
    } else if (kind == BAND_KIND_DIAG) {
        diag_tile<MODE>(a, b, meta, prob, h.dn0, red);
    } else {
        const int end_ptr = (t == 0) ? meta.w : 0;
        if (NR > 1 && rs > 0) spmv_tile<MODE, NR, UNR>(a, b, meta, prob, end_ptr, rs, prod, red, srow, h.lg, h.dn0);
        else spmv_tile<MODE, 1, UNR>(a, b, meta, prob, end_ptr, 0, prod, red, srow, h.lg, h.dn0);
    }
}
template <int MODE, int NR, int NP, bool LDSW = false>
__global__ __launch_bounds__(kThreads) void k_spmv_band(SpmvArgs a) { spmv_band_body<MODE, NR, NP, LDSW>(a); }

// V[dst[k]] = val[k] for the entries a band view serves (after k_kval: once per penalty update)
__global__ __launch_bounds__(kThreads) void k_band_pack(const int32_t* __restrict__ dst, const double* __restrict__ val, double* __restrict__ V, int64_t nnz) {
    const int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (k >= nnz) return;
    const int d = dst[k];
    if (d >= 0) V[d] = val[k];
}

// ---------------------------------------------------------------------------
// preconditioner: multi-level block-tridiagonal chain solve + Jacobi
// ---------------------------------------------------------------------------
// Everything k_prec_pre needs to know about its work item in ONE 512-byte record (work item, chain, level table):
// a single trip to memory by 32 lanes instead of the chain work -> chain -> levels (three dependent trips before the
// first factor or vector load could be requested).  Built by the backend from HostSystem::prec_work / chains / levels.
constexpr int kRecLevels = 7;
struct alignas(16) PrecRecord {
    PrecWork wk;
    ChainDesc ch;
    int32_t pad_;
    ChainLevelDesc lv[kRecLevels];
};
static_assert(sizeof(PrecRecord) == 512, "PrecRecord is read as 32 int4");

struct PrecArgs {
    const PrecWork* work;
    const PrecRecord* rec;     // k_prec_pre: one record per work item (see PrecRecord)
    const ChainDesc* chains;
    const ChainLevelDesc* levels;
    const double* fac;
    const float* fac32;    // the same factors as 4-byte values (k_fac_round); read by k_prec_pre<.., float>
    const float* deep;     // lane-major copy of the coarse-level factors (k_deep_pack); read by k_prec_pre<.., float, true>
    const int32_t* node_col;
    const int32_t* diag_cols;
    const double* dinv;
    const int32_t* done;
    const int32_t* prec_part_ptr;  // per problem: range of prec work items
    const int32_t* kblk_part_ptr;  // per problem: range of K row blocks
    UniRanges uni;                 // (k_prec_pre) single-problem handles: those ranges by value
    double* r;
    double* z;
    double* p;             // INIT: receives p = z.  STEP: the direction of the last K p
    const double* w;
    double* xt;
    double* kx;            // K xt: kx += alpha w whenever xt += alpha p
    const double* rz_in;   // partials of the previous r'z   (STEP)
    const double* pw_part; // partials of p'w                (STEP)
    double* rz_out;        // one partial per work item
    unsigned long long* tstamp;  // see KernelStamp
    int debug_skip;        // timing experiments only: 1 run, 2 separator, 4 back-subst, 8 head, 16 tail
    // INIT over several right-hand sides in ONE launch (score_link.hpp: the columns Z = T^-1 U of a refresh): grid.y = n_vec,
    // workgroup (x, y) applies item x to vector y -- r_in, z, p advance by y * vec_stride, rz_out by y * gridDim.x
    int n_vec;
    long long vec_stride;
    int split_update;      // STEP of k_prec_pre: xt += alpha p, kx += alpha w are done by the helper items (kind 2) of this
                           // launch, on CUs the chains leave idle, instead of by the chain / Jacobi workgroups themselves
    // Device-side termination of a PCG solve (the Newton polish; null in the ADMM loop, whose PCG
    // count is fixed).  The first STEP of a solve (gate_first) turns r'z_0 into the problem's
    // threshold tol2[prob] * r'z_0; every later STEP compares the r'z it is about to use with it
    // and, once below, raises done[prob] -- the flag every kernel of the loop tests on entry -- and
    // leaves without touching anything: the launches still queued for that problem become no-ops,
    // and the host reads one flag at its next synchronisation instead of polling r'z every few
    // iterations.  All workgroups of a problem reduce the same partials in the same order, so they
    // take the same decision.
    const double* r_in;    // residual as it stands at entry (== r, except for the first kernels of a Newton PCG solve,
                           // which read the right-hand side -g where the evaluation left it); r receives updates
    int xt_zero;           // STEP: treat xt as zero on entry (first step of a solve: no memset of the solution)
    int early_done;        // as SpmvArgs::early_done
    int32_t* gate_init;    // INIT: per problem, copy done[prob] here and clear gate_used (start of a gated solve)
    int32_t* gate_flag;        // == done (writable)
    const double* gate_tol2;   // per problem: (relative tolerance)^2
    double* gate_ref;          // per problem: threshold
    int32_t* gate_used;        // per problem: STEPs executed
    int gate_first;
    // The same two words per problem in HOST-mapped memory, stamped with the solve's epoch (stale values of an earlier solve
    // never match): [fired: epoch | used: epoch << 12 | STEPs] -- written by the problem's lead workgroup as it goes, so that the
    // host can queue the PCG a few iterations ahead of the device instead of guessing its length (polish_lockstep).
    int32_t* gate_host;        // 2 * gate_count words, or null
    int32_t gate_epoch, gate_count;
};

enum { PREC_INIT = 0, PREC_STEP = 1 };



// returns true when the problem's PCG has converged (uniform over the workgroup)
__device__ __forceinline__ bool pcg_gate(const PrecArgs& a, int prob, double rz, double ref_loaded) {
    if (!a.gate_flag) return false;
    const bool lead = (threadIdx.x == 0) && ((int)blockIdx.x == a.prec_part_ptr[prob]);
    if (a.gate_first) {
        if (!(rz > 0.0)) {
            if (lead) { a.gate_flag[prob] = 1; if (a.gate_host) a.gate_host[prob] = a.gate_epoch; }
            return true;
        }
        if (lead) {
            a.gate_ref[prob] = rz * a.gate_tol2[prob]; a.gate_used[prob] = 1;
            if (a.gate_host) a.gate_host[a.gate_count + prob] = (a.gate_epoch << 12) | 1;
        }
        return false;
    }
    if (!(rz > ref_loaded)) {  // converged (or NaN: stop, the host sees it in F)
        if (lead) { a.gate_flag[prob] = 1; if (a.gate_host) a.gate_host[prob] = a.gate_epoch; }
        return true;
    }
    if (lead) {
        const int32_t u = a.gate_used[prob] + 1;
        a.gate_used[prob] = u;
        if (a.gate_host) a.gate_host[a.gate_count + prob] = (a.gate_epoch << 12) | (u & 0xfff);
    }
    return false;
}

constexpr int kPrecThreads = 512;
constexpr int kPrecWaves = kPrecThreads / 64;
constexpr int kPrecChunk = 6;   // entries per lane whose loads are issued together
constexpr int kMaxLevels = 20;  // radix 2 .. 4: chains of up to 2^20 nodes (the host checks)

// Factor blocks of one run (thread j of level L): all loads unconditional on a
// clamped position so that they are issued back to back.
template <int BS, int RMAX>
__device__ __forceinline__ void load_run(const double* __restrict__ fac, const ChainLevelDesc& L, int j,
                                         double (&Lf)[RMAX][BS * BS], double (&Dv)[RMAX][BS * BS]) {
    constexpr int B2 = BS * BS;
    const bool last = (L.p == 0);
    const int lo = last ? 0 : j * L.p;
    const int hi = last ? L.N : min(j * L.p + L.p - 1, L.N);
    const int len = max(hi - lo, 1);
    const size_t eP = (size_t)L.P * L.nruns;
    const double* __restrict__ R = fac + L.offR;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
        const double* __restrict__ Rq = R + (size_t)min(q, len - 1) * L.nruns + j;
#pragma unroll
        for (int e = 0; e < B2; ++e) {
            Lf[q][e] = Rq[(size_t)e * eP];
            Dv[q][e] = Rq[(size_t)(B2 + e) * eP];
        }
    }
}

// Jacobi work item (columns outside every chain): z = r / diag(K), with the PCG step folded in.
template <int MODE>
__device__ __forceinline__ double prec_jacobi_item(const PrecArgs& a, const PrecWork& wk, double acc_rz, double acc_pw,
                                                   double* red, bool& stop) {
    const int t = threadIdx.x;
    double alpha = 0.0, local = 0.0;
    stop = false;
    if (MODE == PREC_STEP) {
        const double gref = (a.gate_flag && !a.gate_first) ? a.gate_ref[wk.prob] : 0.0;
        block_sum2_n<kPrecWaves>(acc_rz, acc_pw, red);
        alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
        if (pcg_gate(a, wk.prob, acc_rz, gref)) { stop = true; return 0.0; }
    }
    // Jacobi columns.  Every load is unconditional on a clamped index so that the
    // whole chunk is in flight at once; only the stores are predicated.
    const int e_end = wk.index + wk.count;
    for (int base = wk.index + t; base < e_end; base += kPrecThreads * kPrecChunk) {
        int cols[kPrecChunk];
        double rv[kPrecChunk], dv[kPrecChunk], pv[kPrecChunk], wv[kPrecChunk], xv[kPrecChunk], kv[kPrecChunk];
#pragma unroll
        for (int u = 0; u < kPrecChunk; ++u) cols[u] = a.diag_cols[min(base + u * kPrecThreads, e_end - 1)];
#pragma unroll
        for (int u = 0; u < kPrecChunk; ++u) {
            rv[u] = a.r_in[cols[u]];
            dv[u] = a.dinv[min(base + u * kPrecThreads, e_end - 1)];
            if (MODE == PREC_STEP) {
                wv[u] = a.w[cols[u]];
                if (!a.split_update) { pv[u] = a.p[cols[u]]; xv[u] = a.xt_zero ? 0.0 : a.xt[cols[u]]; kv[u] = a.kx[cols[u]]; }
                else { pv[u] = 0.0; xv[u] = 0.0; kv[u] = 0.0; }
            }
        }
#pragma unroll
        for (int u = 0; u < kPrecChunk; ++u) {
            if (base + u * kPrecThreads < e_end) {
                double r_ = rv[u];
                if (MODE == PREC_STEP) {
                    if (!a.split_update) {
                        NTSP(a.xt[cols[u]], xv[u] + alpha * pv[u]);
                        NTSP(a.kx[cols[u]], kv[u] + alpha * wv[u]);
                    }
                    r_ -= alpha * wv[u];
                    NTSP(a.r[cols[u]], r_);
                }
                const double zv = r_ * dv[u];
                NTSP(a.z[cols[u]], zv);
                if (MODE == PREC_INIT) NTSP(a.p[cols[u]], zv);
                local += r_ * zv;
            }
        }
    }
    return local;
}

// BS: block size, RMAX: radix - 1 (nodes per run), LDS0: level-0 vector in LDS.
// Phases of one level: (run) every run of <= RMAX nodes is solved by one lane;
// (sep) reduced right-hand sides of the separators go to the next level;
// afterwards (back) the levels are back-substituted coarse to fine.  The factor
// blocks of a phase never depend on vector data, so they are requested BEFORE the
// barrier that ends the previous phase and arrive while it drains.
// (INIT launches over several vectors: see PrecArgs::n_vec)
__device__ __forceinline__ void prec_select_vector(PrecArgs& a) {
    if (a.n_vec > 1) {
        const long long off = (long long)blockIdx.y * a.vec_stride;
        a.r_in += off; a.z += off; a.p += off;
        a.rz_out += (size_t)blockIdx.y * gridDim.x;
    }
}
template <int BS, int RMAX, int MODE, bool LDS0>
__global__ __launch_bounds__(kPrecThreads) void k_prec(PrecArgs a) {
    if (MODE == PREC_INIT) prec_select_vector(a);
    KernelStamp stamp(a.tstamp);
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [0,16) reductions, [16,16+6*kMaxLevels) level table, vectors
    __shared__ ChainLevelDesc sLv[kMaxLevels];
    double* red = lds;
    const PrecWork wk = a.work[blockIdx.x];
    const int prob = wk.prob;
    if (MODE == PREC_INIT && a.gate_init && threadIdx.x == 0 && (int)blockIdx.x == a.prec_part_ptr[prob]) {
        a.gate_init[prob] = a.done[prob];
        a.gate_used[prob] = 0;
    }
    if (a.done[prob]) return;
    const int t = threadIdx.x;
    // partial sums for alpha = r'z / p'w: loads first, reduction after the phase-0 loads are out
    double acc_rz = 0.0, acc_pw = 0.0;
    if (MODE == PREC_STEP) {
        const int l0 = a.prec_part_ptr[prob], l1 = a.prec_part_ptr[prob + 1];
        const int k0 = a.kblk_part_ptr[prob], k1 = a.kblk_part_ptr[prob + 1];
        for (int i = l0 + t; i < l1; i += kPrecThreads) acc_rz += a.rz_in[i];
        for (int i = k0 + t; i < k1; i += kPrecThreads) acc_pw += a.pw_part[i];
    }
    double alpha = 0.0;
    double local = 0.0;
    if (wk.kind == 1) {
        bool stop;
        local = prec_jacobi_item<MODE>(a, wk, acc_rz, acc_pw, red, stop);
        if (stop) return;
    } else {
        constexpr int B2 = BS * BS;
        const double gref = (MODE == PREC_STEP && a.gate_flag && !a.gate_first) ? a.gate_ref[prob] : 0.0;
        const ChainDesc ch = a.chains[wk.index];
        const ChainLevelDesc* __restrict__ lv = a.levels + ch.level_begin;
        const int32_t* __restrict__ nc = a.node_col + ch.node_begin;
        const double* __restrict__ fac = a.fac;
        const int N = ch.N;
        const int NB = N * BS;
        const int nl = ch.n_levels;
        const int stride = ch.col_stride, col0 = ch.col0;
        auto colof = [&](int node) -> int { return stride ? col0 + node * stride : nc[node]; };
        // the residual after this kernel's own update: STEP writes it to r, INIT leaves it where it was
        const double* __restrict__ rcur = (MODE == PREC_INIT) ? a.r_in : a.r;
        double* v0 = lds + 16;                                      // level-0 vector (LDS0)
        double* vup = lds + 16 + (LDS0 ? (size_t)NB : (size_t)0);  // levels >= 1
        // level table -> LDS (read by every phase; one batch of loads instead of one per level)
        if (t < nl) sLv[t] = lv[t];
        ChainLevelDesc L = lv[0];
        // level-0 run factors requested before anything else is waited for
        double Lf[RMAX][B2], Dv[RMAX][B2];
        const bool wave_has_run0 = ((t & ~63) < L.nruns);
        if (wave_has_run0) load_run<BS, RMAX>(fac, L, min(t, L.nruns - 1), Lf, Dv);
        // ---- residual: loads of the first chunk out, then alpha (uniform code, every
        //      lane joins the reduction), then the update; further chunks only for long chains
        {
            int cols[kPrecChunk];
            double rv[kPrecChunk], pv[kPrecChunk], wv[kPrecChunk], xv[kPrecChunk], kv[kPrecChunk];
            auto chunk_load = [&](int base) {
#pragma unroll
                for (int u = 0; u < kPrecChunk; ++u) {
                    const int idx = min(base + u * kPrecThreads, NB - 1);
                    const int node = idx / BS;
                    cols[u] = colof(node) + (idx - node * BS);
                }
#pragma unroll
                for (int u = 0; u < kPrecChunk; ++u) {
                    rv[u] = a.r_in[cols[u]];
                    if (MODE == PREC_STEP) { pv[u] = a.p[cols[u]]; wv[u] = a.w[cols[u]]; xv[u] = a.xt_zero ? 0.0 : a.xt[cols[u]]; kv[u] = a.kx[cols[u]]; }
                }
            };
            auto chunk_apply = [&](int base) {
#pragma unroll
                for (int u = 0; u < kPrecChunk; ++u) {
                    const int idx = base + u * kPrecThreads;
                    if (idx < NB) {
                        double r_ = rv[u];
                        if (MODE == PREC_STEP) {
                            NTSP(a.xt[cols[u]], xv[u] + alpha * pv[u]);
                            NTSP(a.kx[cols[u]], kv[u] + alpha * wv[u]);
                            r_ -= alpha * wv[u];
                            NTSP(a.r[cols[u]], r_);
                        }
                        if (LDS0) v0[idx] = r_;
                    }
                }
            };
            chunk_load(t);
            if (MODE == PREC_STEP) {
                block_sum2_n<kPrecWaves>(acc_rz, acc_pw, red);
                alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
                if (pcg_gate(a, prob, acc_rz, gref)) return;
            }
            chunk_apply(t);
            for (int base = t + kPrecThreads * kPrecChunk; base < NB; base += kPrecThreads * kPrecChunk) {
                chunk_load(base);
                chunk_apply(base);
            }
        }
        __syncthreads();
        auto vld = [&](int l, const ChainLevelDesc& Lx, int i, int c, bool input) -> double {
            if (l == 0) {
                if (LDS0) return v0[i * BS + c];
                return input ? rcur[colof(i) + c] : a.z[colof(i) + c];
            }
            return vup[(size_t)(Lx.vec_off + i) * BS + c];
        };
        auto vst = [&](int l, const ChainLevelDesc& Lx, int i, int c, double val) {
            if (l == 0) {
                if (LDS0) v0[i * BS + c] = val; else a.z[colof(i) + c] = val;
            } else {
                vup[(size_t)(Lx.vec_off + i) * BS + c] = val;
            }
        };
        // solves run j of level l with the blocks in (Lf, Dv)
        auto run_compute = [&](int l, const ChainLevelDesc& Lx, int j, const double (&Lfx)[RMAX][B2],
                               const double (&Dvx)[RMAX][B2]) {
            const bool last = (Lx.p == 0);
            const int lo = last ? 0 : j * Lx.p;
            const int hi = last ? Lx.N : min(j * Lx.p + Lx.p - 1, Lx.N);
            const int len = hi - lo;
            if (len <= 0) return;
            double y[RMAX][BS];
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int qq = min(q, len - 1);
#pragma unroll
                for (int c = 0; c < BS; ++c) y[q][c] = vld(l, Lx, lo + qq, c, true);
            }
#pragma unroll
            for (int q = 1; q < RMAX; ++q) {
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double s_ = y[q][c];
#pragma unroll
                    for (int k = 0; k < BS; ++k) s_ -= Lfx[q][c * BS + k] * y[q - 1][k];
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
                    for (int k = 0; k < BS; ++k) s_ += Dvx[q][c * BS + k] * y[q][k];
                    tmp[c] = s_;
                }
                if (q + 1 < RMAX) {
                    const bool has_next = (q + 1 < len);
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        double s_ = 0.0;
#pragma unroll
                        for (int k = 0; k < BS; ++k) s_ += Lfx[q + 1][k * BS + c] * y[q + 1][k];
                        tmp[c] = has_next ? tmp[c] - s_ : tmp[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < BS; ++c) y[q][c] = tmp[c];
                if (q < len) {
#pragma unroll
                    for (int c = 0; c < BS; ++c) vst(l, Lx, lo + q, c, tmp[c]);
                }
            }
        };
        for (int l = 0; l < nl; ++l) {
            const bool last = (L.p == 0);
            const int nsep = L.nsep;
            // separator blocks of this level: requested now, used after the barrier
            double Cl[B2], Cr[B2];
            constexpr bool kPrefetchSep = (BS <= 3);  // 4x4 blocks: the register file is full already
            const bool wave_has_sep = kPrefetchSep && !last && ((t & ~63) < nsep);
            if (wave_has_sep) {
                const double* __restrict__ S = fac + L.offS;
                const int j = min(t, nsep - 1);
#pragma unroll
                for (int e = 0; e < B2; ++e) {
                    Cl[e] = S[(size_t)e * nsep + j];
                    Cr[e] = S[(size_t)(B2 + e) * nsep + j];
                }
            }
            // ---- run phase ----
            if (!(a.debug_skip & 1)) {
                for (int j = t; j < L.nruns; j += kPrecThreads) {
                    if (j != t) load_run<BS, RMAX>(fac, L, j, Lf, Dv);  // long chains only: not prefetched
                    run_compute(l, L, j, Lf, Dv);
                }
            }
            __syncthreads();
            if (last) break;
            const ChainLevelDesc Ln = sLv[l + 1];
            // next level's run blocks: requested before the separator arithmetic
            if ((t & ~63) < Ln.nruns) load_run<BS, RMAX>(fac, Ln, min(t, Ln.nruns - 1), Lf, Dv);
            // ---- separator phase ----
            if (!(a.debug_skip & 2)) {
                for (int j = t; j < nsep; j += kPrecThreads) {
                    if (j >= kPrecThreads || !kPrefetchSep) {  // not prefetched
                        const double* __restrict__ S = fac + L.offS;
#pragma unroll
                        for (int e = 0; e < B2; ++e) {
                            Cl[e] = S[(size_t)e * nsep + j];
                            Cr[e] = S[(size_t)(B2 + e) * nsep + j];
                        }
                    }
                    const int s = j * L.p + L.p - 1;
                    const int sr = min(s + 1, L.N - 1);  // Cr is zero when there is no right run
                    double v[BS], ym[BS], yp[BS];
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        v[c] = vld(l, L, s, c, true);
                        ym[c] = vld(l, L, s - 1, c, false);
                        yp[c] = vld(l, L, sr, c, false);
                    }
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        double acc = v[c];
#pragma unroll
                        for (int k = 0; k < BS; ++k) acc -= Cl[c * BS + k] * ym[k] + Cr[c * BS + k] * yp[k];
                        vup[(size_t)(Ln.vec_off + j) * BS + c] = acc;
                    }
                }
            }
            __syncthreads();
            L = Ln;
        }
        // ---- back-substitution, coarse to fine; spikes of the next batch are requested
        //      before the barrier of the current one ----
        double V[B2], W[B2];
        auto load_spikes = [&](const ChainLevelDesc& Lx, int i) {
            const double* __restrict__ Bk = fac + Lx.offB;
#pragma unroll
            for (int e = 0; e < B2; ++e) {
                V[e] = Bk[(size_t)e * Lx.N + i];
                W[e] = Bk[(size_t)(B2 + e) * Lx.N + i];
            }
        };
        bool have_spikes = false;  // V/W hold this lane's spikes of the level about to be processed
        if (nl >= 2) {
            const ChainLevelDesc Lb = sLv[nl - 2];
            if ((t & ~63) < Lb.N) load_spikes(Lb, min(t, Lb.N - 1));
            have_spikes = true;
        }
        for (int l = nl - 2; l >= 0; --l) {
            const ChainLevelDesc Lb = sLv[l];
            const ChainLevelDesc Ln = sLv[l + 1];
            const int nsep = Lb.nsep;
            const bool single_batch = (Lb.N <= kPrecThreads);
            if (!(a.debug_skip & 4)) {
                for (int i = t; i < Lb.N; i += kPrecThreads) {
                    if (i >= kPrecThreads || !have_spikes) load_spikes(Lb, i);
                    const int j = i / Lb.p;
                    const bool is_sep = (i - j * Lb.p == Lb.p - 1) && (j < nsep);
                    // V is zero without a left separator, W without a right one: clamp and use
                    const int jl = max(j - 1, 0), jr = min(j, max(nsep - 1, 0));
                    double v[BS], ul[BS], ur[BS];
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        v[c] = vld(l, Lb, i, c, false);
                        ul[c] = vup[(size_t)(Ln.vec_off + jl) * BS + c];
                        ur[c] = vup[(size_t)(Ln.vec_off + jr) * BS + c];
                    }
                    double outv[BS];
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        double acc = v[c];
#pragma unroll
                        for (int k = 0; k < BS; ++k) acc -= V[c * BS + k] * ul[k] + W[c * BS + k] * ur[k];
                        outv[c] = is_sep ? ur[c] : acc;  // a separator takes the coarse solution
                    }
#pragma unroll
                    for (int c = 0; c < BS; ++c) vst(l, Lb, i, c, outv[c]);
                }
            }
            // this lane's spikes of the finer level are requested before the barrier; valid only
            // if this level was a single batch (otherwise V/W were reused above)
            have_spikes = false;
            if (l > 0 && single_batch) {
                const ChainLevelDesc Lf_ = sLv[l - 1];
                if ((t & ~63) < Lf_.N) load_spikes(Lf_, min(t, Lf_.N - 1));
                have_spikes = true;
            }
            __syncthreads();
        }
        // ---- write z (LDS0), p (INIT) and accumulate r'z ----
        for (int base = t; base < NB && !(a.debug_skip & 16); base += kPrecThreads * kPrecChunk) {
            int cols[kPrecChunk];
            double rv[kPrecChunk], zz[kPrecChunk];
#pragma unroll
            for (int u = 0; u < kPrecChunk; ++u) {
                const int idx = min(base + u * kPrecThreads, NB - 1);
                const int node = idx / BS;
                cols[u] = colof(node) + (idx - node * BS);
            }
#pragma unroll
            for (int u = 0; u < kPrecChunk; ++u) {
                const int idx = min(base + u * kPrecThreads, NB - 1);
                rv[u] = rcur[cols[u]];
                zz[u] = LDS0 ? v0[idx] : a.z[cols[u]];
            }
#pragma unroll
            for (int u = 0; u < kPrecChunk; ++u) {
                const int idx = base + u * kPrecThreads;
                if (idx < NB) {
                    if (LDS0) NTSP(a.z[cols[u]], zz[u]);
                    if (MODE == PREC_INIT) NTSP(a.p[cols[u]], zz[u]);
                    local += rv[u] * zz[u];
                }
            }
        }
    }
    const double tot = block_sum_n<kPrecWaves>(local, red);
    if (t == 0) a.rz_out[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// k_prec_pre: the same solve with every factor block requested up front.
//
// A phase of k_prec is a few dozen FMAs between two barriers -- far shorter than
// a trip to L2/HBM -- so requesting a phase's blocks one phase ahead still exposes
// most of the latency once per phase (measured: ~3 us per level).  Here nothing
// is requested inside the level loop:
//   * level 0 (3/4 of the data) lives in registers: lane j < 256 holds run j and
//     separator j for the whole kernel.  The level-0 spikes are not read at all: the
//     back-substitution re-solves each run against the separator couplings, which the
//     lane already holds (144 instead of 324 bytes per node);
//   * the factors of ALL coarser levels (contiguous in `fac`) are staged into LDS by
//     lanes 256..511 while lanes 0..255 fetch level 0, so the coarse phases only
//     touch LDS.
// Every load of the kernel is therefore in flight before the first barrier, apart
// from the operands of the fused xt / kx update, which are requested many phases
// before their use.  The host selects this kernel when every
// chain fits (HipBackend::init: <= 256 level-0 runs, coarse factors fit the
// staging registers and LDS); otherwise k_prec runs.
// ---------------------------------------------------------------------------
// orders the LDS traffic of one wavefront: what its lanes wrote before is visible to all of them after
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Block barrier for phases that only exchange data through LDS: __syncthreads() also waits for
// every outstanding global load and store (vmcnt(0)), which would serialise the prefetches this
// kernel keeps in flight across its phases.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int kPreRunLanes = 256;
template <int BS>
struct PreTile {
    static constexpr int B2 = BS * BS;
    static constexpr int NG = (8 * B2 > 32) ? 8 * B2 : 32;  // run (6 B2) + separator (2 B2) blocks | staging
    static constexpr int CH = (BS >= 4) ? 8 : kPrecChunk;    // vector entries per lane: chains of up to CH * 512 / BS nodes
};

// REGDEEP (4-byte stream, block size <= 3, every chain with a lane plan): the coarse levels' factors never touch LDS.
// k_deep_pack has laid them out lane by lane -- staging lane dt: [0, 2 B2) spike blocks of level-1 node dt, [2 B2, 10 B2)
// the run + separator blocks of the run it serves, [10 B2, 12 B2) the spike blocks of its own level's node (levels >= 2) --
// so a lane loads its slots straight into registers (16-byte packets, packet-major: coalesced) and every coarse phase starts its arithmetic
// at once instead of pulling 18..72 values from LDS first.  LDS then holds vectors only (41 KB instead of 135 KB at
// 1000 nodes): three chains per CU can be resident.  The level-0 tile is kept as floats too (converted where used).
template <int BS, int MODE, typename FT, bool REGDEEP>
__device__ __forceinline__ void prec_pre_body(const PrecArgs& a);
template <int BS, int MODE, typename FT = double, bool REGDEEP = false>
__global__ __launch_bounds__(kPrecThreads) void k_prec_pre(PrecArgs a) {
    if (MODE == PREC_INIT) prec_select_vector(a);
    prec_pre_body<BS, MODE, FT, REGDEEP>(a);
}
// (Round 5, measured and removed: the same body compiled for TWO workgroups per CU -- __launch_bounds__(512, 4): 128 registers
//  per lane, 432 of them spilled, 672 bytes of scratch per lane -- for lock-step batches whose chain work items otherwise run in
//  two rounds.  64 config-5 trials in 4 handles of 16: 4170-4230 -> 2180-2210 problems/s.  A two-per-CU chain kernel needs a
//  different division of labour, not a register cap; profiles/TRIED.md.)
template <int BS, int MODE, typename FT, bool REGDEEP>
__device__ __forceinline__ void prec_pre_body(const PrecArgs& a) {
    static_assert(!REGDEEP || (sizeof(FT) == 4 && BS <= 3), "register-resident coarse levels: 4-byte stream, blocks up to 3 x 3");
    KernelStamp stamp(a.tstamp);
    constexpr int RMAX = 3;
    constexpr int B2 = BS * BS;
    constexpr int oRun = REGDEEP ? 2 * B2 : 0;                 // first register of a lane's run blocks
    constexpr int oBk2 = 10 * B2;                              // REGDEEP: spike blocks of the lane's own level (>= 2)
    constexpr int NG = REGDEEP ? 12 * B2 : PreTile<BS>::NG;
    constexpr int oCl = oRun + 2 * RMAX * B2, oCr = oCl + B2;
    constexpr int kStageLanes = kPrecThreads - kPreRunLanes;
    constexpr int CH = PreTile<BS>::CH;  // vector entries per lane
    // 4 x 4 blocks (3-D): the level-0 tile stays in the factor stream's own type in registers and the coarse levels'
    // factors in LDS likewise -- with the 4-byte stream (the only one this block size is launched with) that is 128
    // registers and 84 KB instead of 256 and 169 KB; converted where used
    using LT = typename std::conditional<(BS >= 4 || REGDEEP), FT, double>::type;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ PrecRecord srec;
    double* red = lds;
    const int t = threadIdx.x;
    // trip 1: the work item's record (and, for a single-problem handle, the partial sums of alpha: their ranges are launch
    // constants).  Trip 2: everything else -- frozen flag, partial sums of a batch, vectors, factors.
    // (the loads of trip 1 back to back, the record first -- the barrier below waits for it alone --, the partials by their own
    //  lanes: a loop of loads in front of the record put two more round trips before everything else; the frozen flag of a
    //  single problem is that of problem 0)
    double acc_rz = 0.0, acc_pw = 0.0;
    int4 rec4 = make_int4(0, 0, 0, 0);
    if (t < 32) rec4 = reinterpret_cast<const int4*>(a.rec + blockIdx.x)[t];
    const int dn0 = a.uni.on ? a.done[0] : 0;
    if (MODE == PREC_STEP && a.uni.on) {
        const int i0 = a.uni.l0 + t, j0 = a.uni.k0 + t;
        if (i0 < a.uni.l1) acc_rz = a.rz_in[i0];
        if (j0 < a.uni.k1) acc_pw = a.pw_part[j0];
        for (int i = i0 + kPrecThreads; i < a.uni.l1; i += kPrecThreads) acc_rz += a.rz_in[i];
        for (int i = j0 + kPrecThreads; i < a.uni.k1; i += kPrecThreads) acc_pw += a.pw_part[i];
    }
    if (t < 32) reinterpret_cast<int4*>(&srec)[t] = rec4;
    __syncthreads();
    const PrecWork wk = srec.wk;
    const ChainLevelDesc* sLv = srec.lv;
    const int prob = wk.prob;
    // The frozen-problem flag.  Gated PCG solves (early_done) test it before anything else: their queue holds launches
    // that are meant to be no-ops.  The ADMM loop requests it here and tests it where the first write would happen --
    // waiting for it now would put one more round trip to memory in front of every load below.
    // (a batch member's flag and the ranges of its partial sums depend on the problem alone: requested together, the flag first)
    const int dn_ld = a.uni.on ? dn0 : a.done[prob];
    int bl0 = 0, bl1 = 0, bk0 = 0, bk1 = 0;
    if (MODE == PREC_STEP && !a.uni.on) {
        bl0 = a.prec_part_ptr[prob]; bl1 = a.prec_part_ptr[prob + 1];
        bk0 = a.kblk_part_ptr[prob]; bk1 = a.kblk_part_ptr[prob + 1];
    }
    int dn = 0;
    if (a.early_done) {
        dn = dn_ld;
        if (dn) return;
    }
    const int dn_late = a.early_done ? 0 : dn_ld;
    if (MODE == PREC_INIT && a.gate_init && threadIdx.x == 0 && (int)blockIdx.x == a.prec_part_ptr[prob]) {
        a.gate_init[prob] = a.early_done ? dn : dn_late;
        a.gate_used[prob] = 0;
    }
    if (MODE == PREC_STEP && !a.uni.on) {
        const int l0 = bl0, l1 = bl1, k0 = bk0, k1 = bk1;
        if (l0 + t < l1) acc_rz = a.rz_in[l0 + t];
        if (k0 + t < k1) acc_pw = a.pw_part[k0 + t];
        for (int i = l0 + t + kPrecThreads; i < l1; i += kPrecThreads) acc_rz += a.rz_in[i];
        for (int i = k0 + t + kPrecThreads; i < k1; i += kPrecThreads) acc_pw += a.pw_part[i];
    }
    double local = 0.0;
    if (wk.kind == 2) {
        // update helper (STEP with split_update; appended after the problem's own items, so never the gate's lead and
        // outside the r'z partials): entries [index, index + count) of xt and kx.  alpha and the gate's verdict are
        // derived from the same partial sums in the same order as in the chain workgroups -- the same bits.
        if (dn || dn_late) return;
        if (MODE == PREC_INIT) return;  // (helpers ride on STEP launches only)
        const double gref = (a.gate_flag && !a.gate_first) ? a.gate_ref[prob] : 0.0;
        const int e_end = wk.index + wk.count;
        int idx[kPrecChunk];
        double pv[kPrecChunk], wv[kPrecChunk], xv[kPrecChunk], kv[kPrecChunk];
#pragma unroll
        for (int u = 0; u < kPrecChunk; ++u) {
            idx[u] = min(wk.index + t + u * kPrecThreads, e_end - 1);
            pv[u] = a.p[idx[u]]; wv[u] = a.w[idx[u]]; xv[u] = a.xt_zero ? 0.0 : a.xt[idx[u]]; kv[u] = a.kx[idx[u]];
        }
        block_sum2_n<kPrecWaves>(acc_rz, acc_pw, red);
        const double alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
        if (a.gate_flag && (a.gate_first ? !(acc_rz > 0.0) : !(acc_rz > gref))) return;  // (pcg_gate's test, without its writes)
#pragma unroll
        for (int u = 0; u < kPrecChunk; ++u) {
            if (wk.index + t + u * kPrecThreads < e_end) {
                NTSP(a.xt[idx[u]], xv[u] + alpha * pv[u]);
                NTSP(a.kx[idx[u]], kv[u] + alpha * wv[u]);
            }
        }
        return;
    }
    if (wk.kind == 1) {
        if (dn || dn_late) return;
        bool stop;
        local = prec_jacobi_item<MODE>(a, wk, acc_rz, acc_pw, red, stop);
        if (stop) return;
    } else {
        const double gref = (MODE == PREC_STEP && a.gate_flag && !a.gate_first) ? a.gate_ref[prob] : 0.0;
        const ChainDesc ch = srec.ch;
        // factor stream: 8-byte values, or their float copies (half the bytes through this CU; converted on arrival)
        const FT* __restrict__ fac = sizeof(FT) == 4 ? (const FT*)(const void*)a.fac32 : (const FT*)(const void*)a.fac;
        const int N = ch.N;
        const int NB = N * BS;
        const int nl = ch.n_levels;
        const int stride = ch.col_stride, col0 = ch.col0;
        // (regular chains only -- the backend checks: no trip to node_col in front of the vector loads)
        auto colof = [&](int node) -> int { return col0 + node * stride; };
        // vectors of all levels, padded (see ChainLevelDesc::lds_off): node i, component c of level
        // L at vb[L.lds_off + i * BS + i / L.p + c]
        double* vb = lds + 16;
        double* v0 = vb;  // level 0
        const ChainLevelDesc L0 = sLv[0];
        const ChainLevelDesc Lz = sLv[nl - 1];
        LT* lfac = reinterpret_cast<LT*>(vb + Lz.lds_off + Lz.N * BS + 1);  // factors of the levels >= 1
        auto pad = [](const ChainLevelDesc& L, int i) -> int { return (int)(((uint32_t)i * L.inv_p) >> 20); };  // i / p (0 on the last level)
        auto nodep = [&](const ChainLevelDesc& L, int i) -> double* { return vb + L.lds_off + i * BS + pad(L, i); };
        int64_t deep_base = 0;
        int deep_cnt = 0;
        if (nl >= 2) {
            deep_base = sLv[1].offR;
            deep_cnt = (int)(Lz.offB + (int64_t)2 * B2 * Lz.N - deep_base);
        }
        // coupling terms handed from a separator to the run on its right (REGDEEP: no factors in LDS, they follow the vectors)
        double* xch = REGDEEP ? vb + Lz.lds_off + Lz.N * BS + 1 : reinterpret_cast<double*>(lfac + ((deep_cnt + 1) & ~1));
        // ---- vector loads (only r and w feed the solve; the operands of the xt / kx update
        //      are requested later: the register file is full here) ----
        int cols[CH];
        double rv[CH], wv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = min(t + u * kPrecThreads, NB - 1);
            const int node = idx / BS;
            cols[u] = colof(node) + (idx - node * BS);
        }
        // (4 x 4 blocks: w is requested after the factor tile, just before its use -- the tile fills the register file)
        constexpr bool kLateW = (BS >= 4);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            rv[u] = a.r_in[cols[u]];
            if (MODE == PREC_STEP && !kLateW) wv[u] = a.w[cols[u]];
        }
        // ---- factor loads: level 0 -> registers (lanes < 256), coarser levels -> staging ----
        FT Gr[NG];  // as loaded (converted to double once they have arrived: after the vector update below)
        const bool l0_last = (L0.p == 0);
        if (t < kPreRunLanes) {
            if ((t & ~63) < L0.nruns && !(a.debug_skip & 1)) {
                const int j = min(t, L0.nruns - 1);
                const int lo = l0_last ? 0 : j * L0.p;
                const int hi = l0_last ? L0.N : min(j * L0.p + L0.p - 1, L0.N);
                const int len = max(hi - lo, 1);
                const size_t eP = (size_t)L0.P * L0.nruns;
                const FT* __restrict__ R = fac + L0.offR;
#pragma unroll
                for (int q = 0; q < RMAX; ++q) {
                    const FT* __restrict__ Rq = R + (size_t)min(q, len - 1) * L0.nruns + j;
#pragma unroll
                    for (int e = 0; e < B2; ++e) {
                        Gr[oRun + q * B2 + e] = Rq[(size_t)e * eP];
                        Gr[oRun + (RMAX + q) * B2 + e] = Rq[(size_t)(B2 + e) * eP];
                    }
                }
                if (L0.nsep > 0) {
                    const FT* __restrict__ S = fac + L0.offS;
                    const int js = min(t, L0.nsep - 1);
#pragma unroll
                    for (int e = 0; e < B2; ++e) {
                        Gr[oCl + e] = S[(size_t)e * L0.nsep + js];
                        Gr[oCr + e] = S[(size_t)(B2 + e) * L0.nsep + js];
                    }
                }
            }
        } else if (REGDEEP) {
            // this lane's slots of the lane-major copy, as 16-byte packets; slot groups no lane of the wavefront uses are
            // skipped (scalar branches: the host has put the wavefront's needs into lv[0].pad_, chain_lane_plan)
            if (nl >= 2 && !(a.debug_skip & 2)) {
                constexpr int kPad = ((2 * B2 + 3) & ~3) - 2 * B2;    // padding of a spike group (deep_group_pad)
                constexpr int P1 = 2 * B2 + kPad;                      // packets [0, P1 / 4): level-1 spikes
                constexpr int NQ = (12 * B2 + 2 * kPad) / 4;           // packets per lane
                const float4* __restrict__ src = reinterpret_cast<const float4*>(a.deep + ch.deep_off) + (t - kPreRunLanes);
                const int wv_ = __builtin_amdgcn_readfirstlane((t - kPreRunLanes) >> 6);
                const int mask = __builtin_amdgcn_readfirstlane(L0.pad_);
                const bool need_bk1 = (mask >> wv_) & 1, need_run = (mask >> (4 + wv_)) & 1, need_bk2 = (mask >> (8 + wv_)) & 1;
                float4 pk[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) pk[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (need_bk1) {
#pragma unroll
                    for (int q = 0; q < P1 / 4; ++q) pk[q] = src[(size_t)q * kStageLanes];
                }
                if (need_run) {
#pragma unroll
                    for (int q = P1 / 4; q < P1 / 4 + 2 * B2; ++q) pk[q] = src[(size_t)q * kStageLanes];
                }
                if (need_bk2) {
#pragma unroll
                    for (int q = P1 / 4 + 2 * B2; q < NQ; ++q) pk[q] = src[(size_t)q * kStageLanes];
                }
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int g = k < 2 * B2 ? k : (k < 10 * B2 ? k + kPad : k + 2 * kPad);
                    const float4 v = pk[g >> 2];
                    Gr[k] = (FT)((g & 3) == 0 ? v.x : (g & 3) == 1 ? v.y : (g & 3) == 2 ? v.z : v.w);
                }
            }
        } else if (deep_cnt > 0 && !(a.debug_skip & 2)) {
            const FT* __restrict__ src = fac + deep_base;
            const int lane = t - kPreRunLanes;
#pragma unroll
            for (int k0 = 0; k0 < NG; k0 += 8) {
                if (k0 * kStageLanes < deep_cnt) {  // uniform
#pragma unroll
                    for (int k = k0; k < k0 + 8; ++k) Gr[k] = src[min(lane + k * kStageLanes, deep_cnt - 1)];
                }
            }
        }
        // ---- alpha (every lane joins the reduction), then the vector update into LDS ----
        if (dn_late) return;  // frozen problem (uniform): nothing has been written
        double alpha = 0.0;
        if (MODE == PREC_STEP) {
            block_sum2_n<kPrecWaves>(acc_rz, acc_pw, red);
            alpha = acc_pw > 0.0 ? acc_rz / acc_pw : 0.0;
            if (pcg_gate(a, prob, acc_rz, gref)) return;
        }
        if (MODE == PREC_STEP && kLateW) {
#pragma unroll
            for (int u = 0; u < CH; ++u) wv[u] = a.w[cols[u]];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = t + u * kPrecThreads;
            if (idx < NB) {
                double r_ = rv[u];
                if (MODE == PREC_STEP) {
                    r_ -= alpha * wv[u];
                    NTSP(a.r[cols[u]], r_);
                }
                v0[idx + pad(L0, idx / BS)] = r_;
                rv[u] = r_;
            }
        }
        if (t >= kPreRunLanes && !REGDEEP) {
            if (deep_cnt > 0 && !(a.debug_skip & 2)) {
                const int lane = t - kPreRunLanes;
#pragma unroll
                for (int k0 = 0; k0 < NG; k0 += 8) {
                    if (k0 * kStageLanes < deep_cnt) {
#pragma unroll
                        for (int k = k0; k < k0 + 8; ++k) {
                            const int idx = lane + k * kStageLanes;
                            if (idx < deep_cnt) lfac[idx] = (LT)Gr[k];
                        }
                    }
                }
            }
        }
        LT G[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) G[k] = (LT)Gr[k];
        lds_barrier();
        // in-place solve with the diagonal block of a level-0 run: y <- T_run^-1 y (first len nodes)
        auto run_solve0 = [&](double (&y)[RMAX][BS], int len) {
#pragma unroll
            for (int q = 1; q < RMAX; ++q) {
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double s_ = y[q][c];
#pragma unroll
                    for (int k = 0; k < BS; ++k) s_ -= G[oRun + q * B2 + c * BS + k] * y[q - 1][k];
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
                    for (int k = 0; k < BS; ++k) s_ += G[oRun + (RMAX + q) * B2 + c * BS + k] * y[q][k];
                    tmp[c] = s_;
                }
                if (q + 1 < RMAX) {
                    const bool has_next = (q + 1 < len);
#pragma unroll
                    for (int c = 0; c < BS; ++c) {
                        double s_ = 0.0;
#pragma unroll
                        for (int k = 0; k < BS; ++k) s_ += G[oRun + (q + 1) * B2 + k * BS + c] * y[q + 1][k];
                        tmp[c] = has_next ? tmp[c] - s_ : tmp[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < BS; ++c) y[q][c] = tmp[c];
            }
        };
        // ---- level 0, run phase (registers) ----
        const bool dbg_nophase = (a.debug_skip & 4) != 0;
        if (t < L0.nruns && !dbg_nophase) {
            const int j = t;
            const int lo = l0_last ? 0 : j * L0.p;
            const int hi = l0_last ? L0.N : min(j * L0.p + L0.p - 1, L0.N);
            const int len = hi - lo;
            const int jpad = l0_last ? 0 : j;
            if (len > 0) {
                double y[RMAX][BS];
#pragma unroll
                for (int q = 0; q < RMAX; ++q) {
                    const double* src = v0 + (lo + min(q, len - 1)) * BS + jpad;
#pragma unroll
                    for (int c = 0; c < BS; ++c) y[q][c] = src[c];
                }
                run_solve0(y, len);
#pragma unroll
                for (int q = 0; q < RMAX; ++q) {
                    if (q < len) {
                        double* dst = v0 + (lo + q) * BS + jpad;
#pragma unroll
                        for (int c = 0; c < BS; ++c) dst[c] = y[q][c];
                    }
                }
            }
        }
        lds_barrier();
        if (nl >= 2 && !dbg_nophase) {
            // ---- level 0, separator phase (registers) ----
            const ChainLevelDesc L1 = sLv[1];
            if (t < L0.nsep) {
                const int s = t * L0.p + L0.p - 1;
                const bool has_r = (s + 1 < L0.N);  // Cr is zero when there is no right run
                const double* pv_ = v0 + s * BS + t;
                const double* pm = v0 + (s - 1) * BS + t;
                const double* pp = has_r ? v0 + (s + 1) * BS + t + 1 : pv_;
                double v[BS], ym[BS], yp[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) { v[c] = pv_[c]; ym[c] = pm[c]; yp[c] = pp[c]; }
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double acc = v[c];
#pragma unroll
                    for (int k = 0; k < BS; ++k) acc -= G[oCl + c * BS + k] * ym[k] + G[oCr + c * BS + k] * yp[k];
                    nodep(L1, t)[c] = acc;
                }
            }
            lds_barrier();
        }
        // The step's xt += alpha p, kx += alpha w in two halves (the run blocks stay in registers,
        // there is no room for all operands at once): the first half is in flight during the coarse
        // levels, the second during the level-0 back-substitution.
        constexpr int kHalf = CH / 2;
        double pv[kHalf], wq[kHalf], xv[kHalf], kv[kHalf];
        auto upd_load = [&](auto half) {
            constexpr int u0 = decltype(half)::value * kHalf;
#pragma unroll
            for (int u = 0; u < kHalf; ++u) {
                pv[u] = a.p[cols[u0 + u]]; wq[u] = a.w[cols[u0 + u]]; xv[u] = a.xt_zero ? 0.0 : a.xt[cols[u0 + u]]; kv[u] = a.kx[cols[u0 + u]];
            }
        };
        auto upd_store = [&](auto half) {
            constexpr int u0 = decltype(half)::value * kHalf;
#pragma unroll
            for (int u = 0; u < kHalf; ++u) {
                if (t + (u0 + u) * kPrecThreads < NB) {
                    NTSP(a.xt[cols[u0 + u]], xv[u] + alpha * pv[u]);
                    NTSP(a.kx[cols[u0 + u]], kv[u] + alpha * wq[u]);
                }
            }
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        // (4 x 4 blocks: the register file holds the 128-entry factor tile; the update waits until the tile is dead)
        constexpr bool kLateUpdate = (BS >= 4);
        const bool own_update = (MODE == PREC_STEP) && !a.split_update;  // (uniform)
        if (own_update && !kLateUpdate) upd_load(H0());
        // ---- coarser levels: factors and vectors in LDS.  Lanes 256..511 do this work: their
        //      register tile G is free (the staging is over), so each phase first pulls all its
        //      blocks from LDS into G and only then starts the dependent arithmetic ----
        const int dt = t - kPreRunLanes;
        for (int l = 1; l < nl && !dbg_nophase; ++l) {
            const ChainLevelDesc L = sLv[l];
            const bool last = (L.p == 0);
            // (REGDEEP: the level's runs / separators / nodes are served by the staging lanes lane0 .., whose registers
            //  already hold the blocks)
            const int dl = REGDEEP ? dt - L.lane0 : dt;
            if (dt >= 0 && dl >= 0 && dl < L.nruns) {
                const int j = dl;
                const int lo = last ? 0 : j * L.p;
                const int hi = last ? L.N : min(j * L.p + L.p - 1, L.N);
                const int len = hi - lo;
                if (len > 0) {
                    if (!REGDEEP) {
                        const int eP = L.P * L.nruns;
                        const LT* R = lfac + (L.offR - deep_base);
#pragma unroll
                        for (int q = 0; q < RMAX; ++q) {
                            const LT* Rq = R + min(q, len - 1) * L.nruns + j;
#pragma unroll
                            for (int e = 0; e < B2; ++e) {
                                G[q * B2 + e] = Rq[e * eP];
                                G[(RMAX + q) * B2 + e] = Rq[(B2 + e) * eP];
                            }
                        }
                    }
                    double* vr = vb + L.lds_off + lo * BS + (last ? 0 : j);
                    double y[RMAX][BS];
#pragma unroll
                    for (int q = 0; q < RMAX; ++q)
#pragma unroll
                        for (int c = 0; c < BS; ++c) y[q][c] = vr[min(q, len - 1) * BS + c];
                    run_solve0(y, len);
#pragma unroll
                    for (int q = 0; q < RMAX; ++q) {
                        if (q < len) {
#pragma unroll
                            for (int c = 0; c < BS; ++c) vr[q * BS + c] = y[q][c];
                        }
                    }
                }
            }
            lds_barrier();
            if (last) break;
            const ChainLevelDesc Ln = sLv[l + 1];
            if (dt >= 0 && dl >= 0 && dl < L.nsep) {
                const int js = dl;
                if (!REGDEEP) {
                    const LT* S = lfac + (L.offS - deep_base);
#pragma unroll
                    for (int e = 0; e < B2; ++e) {
                        G[oCl + e] = S[e * L.nsep + js];
                        G[oCr + e] = S[(B2 + e) * L.nsep + js];
                    }
                }
                const int s = js * L.p + L.p - 1;
                const bool has_r = (s + 1 < L.N);
                const double* pv_ = vb + L.lds_off + s * BS + js;
                const double* pm = vb + L.lds_off + (s - 1) * BS + js;
                const double* pp = has_r ? vb + L.lds_off + (s + 1) * BS + js + 1 : pv_;
                double v[BS], ym[BS], yp[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) { v[c] = pv_[c]; ym[c] = pm[c]; yp[c] = pp[c]; }
                double* dst = nodep(Ln, js);
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double acc = v[c];
#pragma unroll
                    for (int k = 0; k < BS; ++k) acc -= G[oCl + c * BS + k] * ym[k] + G[oCr + c * BS + k] * yp[k];
                    dst[c] = acc;
                }
            }
            lds_barrier();
        }
        // ---- back-substitution of the coarser levels (LDS), coarse to fine ----
        for (int l = nl - 2; l >= 1 && !dbg_nophase; --l) {
            const ChainLevelDesc Lb = sLv[l];
            const ChainLevelDesc Ln = sLv[l + 1];
            const int db = REGDEEP ? dt - Lb.lane0 : dt;
            if (dt >= 0 && db >= 0 && db < Lb.N) {
                const int i = db;
                LT Vw[2 * B2];  // spike blocks V, W of node i
                if (REGDEEP) {
#pragma unroll
                    for (int e = 0; e < 2 * B2; ++e) Vw[e] = (l == 1) ? G[e] : G[oBk2 + e];
                } else {
                    const LT* Bk = lfac + (Lb.offB - deep_base);
#pragma unroll
                    for (int e = 0; e < 2 * B2; ++e) Vw[e] = Bk[e * Lb.N + i];
                }
                const int nsep = Lb.nsep;
                const int j = pad(Lb, i);
                const bool is_sep = (i - j * Lb.p == Lb.p - 1) && (j < nsep);
                const int jl = max(j - 1, 0), jr = min(j, max(nsep - 1, 0));
                double* pvx = vb + Lb.lds_off + i * BS + j;
                const double* pul = nodep(Ln, jl);
                const double* pur = nodep(Ln, jr);
                double v[BS], ul[BS], ur[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) { v[c] = pvx[c]; ul[c] = pul[c]; ur[c] = pur[c]; }
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double acc = v[c];
#pragma unroll
                    for (int k = 0; k < BS; ++k) acc -= Vw[c * BS + k] * ul[k] + Vw[B2 + c * BS + k] * ur[k];
                    pvx[c] = is_sep ? ur[c] : acc;
                }
            }
            lds_barrier();
        }
        if (own_update && !kLateUpdate) { upd_store(H0()); upd_load(H1()); }
        // ---- back-substitution of level 0.  No spikes are stored for this level: with the run
        //      blocks still in registers, x_run = y_run - T_run^-1 (e_first Cr_left' x_left +
        //      e_last Cl_right' x_right) costs one more run solve and no memory traffic ----
        if (nl >= 2 && !dbg_nophase) {
            const ChainLevelDesc L1 = sLv[1];
            const int nsep = L0.nsep;
            double c_last[BS];
#pragma unroll
            for (int c = 0; c < BS; ++c) c_last[c] = 0.0;
            if (t < nsep) {  // separator t: takes the coarse solution; couplings to both neighbours
                const int s = t * L0.p + L0.p - 1;
                double xs[BS];
#pragma unroll
                for (int c = 0; c < BS; ++c) xs[c] = nodep(L1, t)[c];
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double sl = 0.0, sr = 0.0;
#pragma unroll
                    for (int k = 0; k < BS; ++k) { sl += G[oCl + k * BS + c] * xs[k]; sr += G[oCr + k * BS + c] * xs[k]; }
                    c_last[c] = sl;                       // Cl' x_s: last node of run t (this lane)
                    xch[(size_t)(t + 1) * BS + c] = sr;   // Cr' x_s: first node of run t + 1
                    v0[s * BS + t + c] = xs[c];
                }
            }
            lds_barrier();
            if (t < L0.nruns) {
                const int lo = t * L0.p;
                const int hi = min(t * L0.p + L0.p - 1, L0.N);
                const int len = hi - lo;
                if (len > 0) {
                    double y[RMAX][BS];
#pragma unroll
                    for (int q = 0; q < RMAX; ++q)
#pragma unroll
                        for (int c = 0; c < BS; ++c) {
                            double v = (q == len - 1) ? c_last[c] : 0.0;
                            if (q == 0 && t >= 1) v += xch[(size_t)t * BS + c];
                            y[q][c] = v;
                        }
                    run_solve0(y, len);
#pragma unroll
                    for (int q = 0; q < RMAX; ++q) {
                        if (q < len) {
                            double* dst = v0 + (lo + q) * BS + t;
#pragma unroll
                            for (int c = 0; c < BS; ++c) dst[c] -= y[q][c];
                        }
                    }
                }
            }
            lds_barrier();
        }
        // ---- write z, p (INIT), the rest of the step's xt / kx, and accumulate r'z ----
        if (own_update && !kLateUpdate) upd_store(H1());
        if (own_update && kLateUpdate) { upd_load(H0()); upd_store(H0()); upd_load(H1()); upd_store(H1()); }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = t + u * kPrecThreads;
            if (idx < NB) {
                const double zz = v0[idx + pad(L0, idx / BS)];
                NTSP(a.z[cols[u]], zz);
                if (MODE == PREC_INIT) NTSP(a.p[cols[u]], zz);
                local += rv[u] * zz;
            }
        }
    }
    const double tot = block_sum_n<kPrecWaves>(local, red);
    if (t == 0) a.rz_out[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// vector update at the end of the PCG sweep (grid = vector blocks)
// ---------------------------------------------------------------------------
struct VecArgs {
    const int32_t* first_row;   // blocks of <= 256 consecutive vector entries of one problem (not the K row blocks:
    const int32_t* end_row;     //  a replicated K holds replica 0's rows only)
    const int32_t* blk_prob;
    const int32_t* done;
    const int32_t* prec_part_ptr;
    const int32_t* kblk_part_ptr;
    const double* rz_old;
    const double* pw_part;
    const double* p;
    const double* w;
    double* kx;
    double* xt;
    double* x;
    double alpha_relax;
    int apply_alpha;  // 0: xt already holds the final CG iterate
};

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
    if (row < a.end_row[b]) {
        const double xt = a.xt[row] + alpha * a.p[row];
        NTS(a.xt[row], xt);
        if (a.apply_alpha) a.kx[row] += alpha * a.w[row];
        a.x[row] = a.alpha_relax * xt + (1.0 - a.alpha_relax) * a.x[row];
    }
}

// ---------------------------------------------------------------------------
// cones: one cone per lane
// ---------------------------------------------------------------------------
struct ConeArgs {
    int xcd_chunk, n_blocks;  // see SpmvArgs::xcd_chunk (0: block i on workgroup i)
    UniRanges uni;
    const int32_t* A_ptr;
    const int32_t* A_col;
    const double* A_val;
    const int32_t* cone_row;
    const int32_t* cone_dim;
    const int32_t* cone_type;
    const int4* cone_meta;  // per cone: {row, dim, type, ptr[row]}, {ptr[row+1..row+4]} (clamped at row+dim)
    // small cones (<= kSmallCone rows of <= kConeRowNnz entries: every SCORE cone): the entries of A again, by cone
    // index -- two int4 of columns (missing entries repeat a valid column) and four double2 of values (missing: 0),
    // so that they are requested together with the cone record instead of after it
    const int4* cone_cols;
    const double2* cone_vals;
    int uniform_cones;      // > 0: one problem, block b holds cones [b * kConesPerBlock, ...) of uniform_cones
    const int32_t* block_first;
    const int32_t* block_prob;
    const int32_t* done;
    const double* rho;
    const double* b;
    const double* xt;   // gathered (xt for the iteration, x for residuals)
    const double* s_in; // k_cone's register path reads the old s, y here;
    const double* y_in; // the new ones go to s, y
    double* s;
    double* y;
    double* u;
    double alpha_relax;
    const double* invE;
    double* pres_part;  // 8 per block
    // fused end-of-PCG update: the gathered vector is xt + a * pfin
    int apply_alpha;
    const double* pfin;
    const double* pw_in;
    const double* rz_in;
    const int32_t* prec_part_ptr;
    const int32_t* kblk_part_ptr;
    double* step_out;   // per problem
    unsigned long long* tstamp;  // see KernelStamp
    int skip_large;     // k_cone leaves cones with more than kWaveCone rows to k_cone_wave
};

__device__ __forceinline__ double a_row_dot(const ConeArgs& a, int i, const double* __restrict__ v) {
    double acc = 0.0;
    for (int k = a.A_ptr[i]; k < a.A_ptr[i + 1]; ++k) acc += a.A_val[k] * v[a.A_col[k]];
    return acc;
}

constexpr int kSmallCone = 4;    // rows
constexpr int kConeRowNnz = 2;   // entries per row handled by the register path
constexpr int kWaveCone = 32;    // cones with more rows than this are projected by one wavefront each (k_cone_wave)

__global__ __launch_bounds__(kThreads) void k_cone(ConeArgs a) {
    KernelStamp stamp(a.tstamp);
    __shared__ double red[8];
    // (XCD-aware block order as in k_spmv: the cone blocks of one problem of a batch project through one L2)
    const int b = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.xcd_chunk > 0 && b >= a.n_blocks) return;
    const int prob = a.uni.on ? 0 : a.block_prob[b];
    // First trip (single problem; a batch member's starts once its problem is known): the frozen flag, the partial sums of the
    // last PCG step's r'z and p'w -- the first 256 of each by their own lanes, not as a loop of loads with a wait of its own in
    // front of everything --, the cone records, rho.  The flag is tested before the first dependent load; the partials are
    // reduced (two barriers) only after the cone's own loads are in flight.
    const int dnv = a.done[prob];
    double rz = 0.0, pw = 0.0;
    if (a.apply_alpha) {
        const int l0 = a.uni.on ? a.uni.l0 : a.prec_part_ptr[prob], l1 = a.uni.on ? a.uni.l1 : a.prec_part_ptr[prob + 1];
        const int k0 = a.uni.on ? a.uni.k0 : a.kblk_part_ptr[prob], k1 = a.uni.on ? a.uni.k1 : a.kblk_part_ptr[prob + 1];
        const int i0 = l0 + (int)threadIdx.x, j0 = k0 + (int)threadIdx.x;
        if (i0 < l1) rz = a.rz_in[i0];
        if (j0 < k1) pw = a.pw_in[j0];
        for (int i = i0 + kThreads; i < l1; i += kThreads) rz += a.rz_in[i];
        for (int i = j0 + kThreads; i < k1; i += kThreads) pw += a.pw_in[i];
    }
    double step = 0.0;  // step length of the last PCG step (its xt update is applied on the fly)
    auto finish_step = [&]() {
        if (a.apply_alpha) {
            block_sum2(rz, pw, red);
            step = pw > 0.0 ? rz / pw : 0.0;
            // every block of the problem computes the same value; the next right-hand-side kernel reads it
            if (threadIdx.x == 0) a.step_out[prob] = step;
        }
    };
    // (one problem: the block boundaries are arithmetic -- one dependent load less)
    const int c_end = a.uniform_cones > 0 ? min((b + 1) * kConesPerBlock, a.uniform_cones) : a.block_first[b + 1];
    const int c = (a.uniform_cones > 0 ? b * kConesPerBlock : a.block_first[b]) + threadIdx.x;
    // one 32-byte record per cone (first row, dimension, type, the row pointers of its first four
    // rows) instead of the chain cone_row -> A_ptr; loaded on a clamped index by every lane
    const int cl = min(c, c_end - 1);
    const int4 m0 = a.cone_meta[2 * cl], m1 = a.cone_meta[2 * cl + 1];
    const int4 pc0 = a.cone_cols[2 * cl], pc1 = a.cone_cols[2 * cl + 1];
    const double2 pv0 = a.cone_vals[4 * cl], pv1 = a.cone_vals[4 * cl + 1], pv2 = a.cone_vals[4 * cl + 2], pv3 = a.cone_vals[4 * cl + 3];
    const int row = m0.x, dim = m0.y, type = m0.z;
    const int ptrs[kSmallCone + 1] = {m0.w, m1.x, m1.y, m1.z, m1.w};
    const double rho = a.rho[prob], irho = 1.0 / rho, al = a.alpha_relax;
    if (dnv) return;  // (uniform over the block; nothing has been written)
    double t0 = 0.0, nz2 = 0.0, head, tail;
    // SCORE's cones have d + 1 = 3 or 4 rows with at most 2 entries each: every column/value,
    // then every gathered x are requested as batches of unconditional loads (clamped indices)
    // instead of a dependent chain per entry.
    bool small = (dim <= kSmallCone);
#pragma unroll
    for (int k = 0; k < kSmallCone; ++k) small = small && (ptrs[k + 1] - ptrs[k] <= kConeRowNnz);
    // (uniform decision per block would need a vote; lanes simply take their own path after the
    //  block-wide reduction below)
    double v[kSmallCone], wv[kSmallCone], yv[kSmallCone], bv[kSmallCone], sv[kSmallCone];
    int cc[kSmallCone][kConeRowNnz];
    double av[kSmallCone][kConeRowNnz], xv[kSmallCone][kConeRowNnz], pvv[kSmallCone][kConeRowNnz];
    if (small) {
        static_assert(kSmallCone == 4 && kConeRowNnz == 2, "packed cone entries: 4 rows x 2 entries");
        cc[0][0] = pc0.x; cc[0][1] = pc0.y; cc[1][0] = pc0.z; cc[1][1] = pc0.w;
        cc[2][0] = pc1.x; cc[2][1] = pc1.y; cc[3][0] = pc1.z; cc[3][1] = pc1.w;
        av[0][0] = pv0.x; av[0][1] = pv0.y; av[1][0] = pv1.x; av[1][1] = pv1.y;
        av[2][0] = pv2.x; av[2][1] = pv2.y; av[3][0] = pv3.x; av[3][1] = pv3.y;
#pragma unroll
        for (int k = 0; k < kSmallCone; ++k) {
            const int i = row + min(k, dim - 1);
            bv[k] = a.b[i];
            yv[k] = a.y_in[i];
            sv[k] = a.s_in[i];
        }
#pragma unroll
        for (int k = 0; k < kSmallCone; ++k)
#pragma unroll
            for (int e = 0; e < kConeRowNnz; ++e) {
                xv[k][e] = a.xt[cc[k][e]];
                pvv[k][e] = a.apply_alpha ? a.pfin[cc[k][e]] : 0.0;
            }
    }
    finish_step();
    if (c >= c_end) return;
    if (small) {
#pragma unroll
        for (int k = 0; k < kSmallCone; ++k) {
            double tt = 0.0;
#pragma unroll
            for (int e = 0; e < kConeRowNnz; ++e)  // (same predicate as before the packing: a missing entry adds an exact 0)
                tt += (ptrs[k] + e < ptrs[k + 1]) ? av[k][e] * (xv[k][e] + step * pvv[k][e]) : 0.0;
            v[k] = al * (bv[k] - tt) + (1.0 - al) * sv[k];
            wv[k] = v[k] - yv[k] * irho;
            if (k == 0) t0 = wv[k]; else if (k < dim) nz2 += wv[k] * wv[k];
        }
        soc_scales(type, t0, nz2, head, tail);
#pragma unroll
        for (int k = 0; k < kSmallCone; ++k)
            if (k < dim) {
                const int i = row + k;
                const double sn = (k == 0) ? head : tail * wv[k];
                const double yn = yv[k] + rho * (sn - v[k]);
                NTS(a.s[i], sn);
                NTS(a.y[i], yn);
                NTS(a.u[i], rho * (bv[k] - sn) - yn);
            }
        return;
    }
    if (a.skip_large && dim > kWaveCone) return;  // k_cone_wave projects it: one wavefront per cone
    for (int k = 0; k < dim; ++k) {
        const int i = row + k;
        double tt = a_row_dot(a, i, a.xt);
        if (a.apply_alpha) tt += step * a_row_dot(a, i, a.pfin);
        const double v_ = al * (a.b[i] - tt) + (1.0 - al) * a.s[i];
        const double w_ = v_ - a.y[i] * irho;
        a.u[i] = v_;   // stash v
        a.s[i] = w_;  // stash the point to project
        if (k == 0) t0 = w_; else nz2 += w_ * w_;
    }
    soc_scales(type, t0, nz2, head, tail);
    for (int k = 0; k < dim; ++k) {
        const int i = row + k;
        const double sn = (k == 0) ? head : tail * a.s[i];
        const double v_ = a.u[i];
        const double yn = a.y[i] + rho * (sn - v_);
        NTS(a.s[i], sn);
        NTS(a.y[i], yn);
        NTS(a.u[i], rho * (a.b[i] - sn) - yn);
    }
}

// Large cones (more than kWaveCone rows; none in a SCORE model, whose cones have d + 1 rows -- the C ABI takes any):
// one WAVEFRONT per cone.  The lanes stride over the rows -- A-row products, relaxation, the point to project -- the
// squared norm of the tail is reduced with wavefront shuffles, and the lanes apply the projection and the dual update to
// the rows they hold.  Four cones per 256-thread workgroup; `large` lists {cone, problem} pairs.
__global__ __launch_bounds__(kThreads) void k_cone_wave(ConeArgs a, const int2* __restrict__ large, int n_large) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (idx >= n_large) return;  // (whole wavefronts leave: no block-wide barrier below)
    const int2 lc = large[idx];
    const int prob = lc.y;
    if (a.done[prob]) return;
    double step = 0.0;
    if (a.apply_alpha) {  // the step length of the last PCG step, as k_cone forms it (fixed order over the partials)
        double rz = 0.0, pw = 0.0;
        for (int i = a.prec_part_ptr[prob] + lane; i < a.prec_part_ptr[prob + 1]; i += 64) rz += a.rz_in[i];
        for (int i = a.kblk_part_ptr[prob] + lane; i < a.kblk_part_ptr[prob + 1]; i += 64) pw += a.pw_in[i];
        rz = wave_sum(rz); pw = wave_sum(pw);
        rz = __shfl(rz, 0, 64); pw = __shfl(pw, 0, 64);
        step = pw > 0.0 ? rz / pw : 0.0;
    }
    const int row = a.cone_row[lc.x], dim = a.cone_dim[lc.x], type = a.cone_type[lc.x];
    const double rho = a.rho[prob], irho = 1.0 / rho, al = a.alpha_relax;
    double t0 = 0.0, nz2 = 0.0;
    for (int k = lane; k < dim; k += 64) {
        const int i = row + k;
        double tt = a_row_dot(a, i, a.xt);
        if (a.apply_alpha) tt += step * a_row_dot(a, i, a.pfin);
        const double v_ = al * (a.b[i] - tt) + (1.0 - al) * a.s[i];
        const double w_ = v_ - a.y[i] * irho;
        a.u[i] = v_;   // stash v (this lane reads it back below)
        a.s[i] = w_;   // stash the point to project
        if (k == 0) t0 = w_; else nz2 += w_ * w_;
    }
    nz2 = wave_sum(nz2);
    nz2 = __shfl(nz2, 0, 64);
    t0 = __shfl(t0, 0, 64);  // (lane 0 holds row 0)
    double head, tail;
    soc_scales(type, t0, nz2, head, tail);
    for (int k = lane; k < dim; k += 64) {
        const int i = row + k;
        const double sn = (k == 0) ? head : tail * a.s[i];
        const double v_ = a.u[i];
        const double yn = a.y[i] + rho * (sn - v_);
        NTS(a.s[i], sn);
        NTS(a.y[i], yn);
        NTS(a.u[i], rho * (a.b[i] - sn) - yn);
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
        NTS(a.u[i], rho * (a.b[i] - a.s[i]) - a.y[i]);
    }
}

// primal residual norms; a.xt points at x here
__global__ __launch_bounds__(kThreads) void k_pres(ConeArgs a) {
    __shared__ double red[8];
    const int b = blockIdx.x;
    const int prob = a.block_prob[b];
    if (a.done[prob]) return;
    const int c = a.block_first[b] + threadIdx.x;
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, sby = 0, sgap = 0, bad = 0;
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
            sgap += a.y[i] * (si - pr);  // s'y - y'r_p
        }
    }
    bad = block_sum(bad, red);
    m0 = block_max(m0, red); m1 = block_max(m1, red); m2 = block_max(m2, red);
    m3 = block_max(m3, red); m4 = block_max(m4, red); m5 = block_max(m5, red);
    sby = block_sum(sby, red);
    sgap = block_sum(sgap, red);
    if (threadIdx.x == 0) {
        double* o = a.pres_part + (size_t)b * kPartStride;
        const double nanv = bad > 0.0 ? __builtin_nan("") : 0.0;
        o[0] = m0 + nanv; o[1] = m1; o[2] = m2; o[3] = m3 + nanv; o[4] = m4; o[5] = m5; o[6] = sby; o[7] = sgap;
    }
}

// ---------------------------------------------------------------------------
// host <-> device words without the copy engine.  A hipMemcpyAsync between two kernels of a stream
// costs the copy kernel plus 10..50 us of cross-queue signalling (measured, rocprofv3 timeline);
// these one-workgroup kernels run in the stream's own queue instead.
//   k_push : up to 3 small device arrays -> host-mapped pinned memory, then a sequence number the
//            host spins on (everything earlier in the stream -- including partials that kernels wrote
//            straight into host-mapped memory -- has completed before this kernel starts)
//   k_fetch: host-mapped pinned memory -> a small device array (control words)
// ---------------------------------------------------------------------------
struct PushArgs {
    const unsigned long long* src[3];
    unsigned long long* dst[3];
    int n[3];                       // 8-byte words
    unsigned long long* flag;       // host-mapped
    unsigned long long seq;
};
__global__ __launch_bounds__(kThreads) void k_push(PushArgs a) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int i = threadIdx.x; i < a.n[k]; i += kThreads) a.dst[k][i] = a.src[k][i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(kThreads) void k_fetch(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n_words) {
    for (int i = threadIdx.x; i < n_words; i += kThreads) dst[i] = src[i];
}
// The same, queued BEFORE the host has written the words: the lead lane waits for the slot's flag (host-mapped memory, raised by
// the host after the payload), then the words are copied.  Whoever queues this must raise the flag without calling the runtime
// in between (HipBackend::prequeue_control).
__global__ __launch_bounds__(kThreads) void k_fetch_wait(const int32_t* src, int32_t* __restrict__ dst, int n_words,
                                                         const unsigned long long* flag, unsigned long long expect) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != expect) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_words; i += kThreads) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// K = K0 + rho[prob] * K1 on the fixed pattern (grid = K row blocks; one tile of <= kTileNnz entries each,
// or one long row)
__global__ __launch_bounds__(kThreads) void k_kval(CsrDev M, const double* __restrict__ K0, const double* __restrict__ K1,
                                                   const double* __restrict__ rho, double* __restrict__ val, const int32_t* skip) {
    const int4 meta = M.blk_meta[blockIdx.x];
    const int prob = M.blk_prob[blockIdx.x];
    if (skip && skip[prob]) return;
    const double r = rho[prob];
    for (int k = meta.z + (int)threadIdx.x; k < meta.w; k += kThreads) val[k] = K0[k] + r * K1[k];
}

// ---------------------------------------------------------------------------
// Ruiz equilibration passes on the device (ruiz_scale in score_host.hpp is the specification and the twin's path):
//   k_ruiz_cols   one WAVEFRONT per swept column j (a landmark's column holds thousands of entries): the lanes stride
//                 over row j of P (symmetric: = column j) and over column j of A through the position map,
//                 |v| D[other] resp. |v| E[row], wavefront max, d_j = 1 / sqrt(D_j * max)
//   k_ruiz_groups one lane per cone group: e_g from its rows (replicated problems: head row + first tail row)
//   k_ruiz_apply  D *= d (copied to the other replicas), E *= e
// Three launches per pass, nothing on the host between them.
// ---------------------------------------------------------------------------
struct RuizArgs {
    const int32_t* P_ptr; const int32_t* P_col; const double* P_val;
    const int32_t* A_ptr; const int32_t* A_col; const double* A_val;
    const int32_t* atp; const int32_t* atpos; const int32_t* arow; const int32_t* gstart;
    double* D; double* E; double* d; double* e;
    int64_t n_act, nr, ngroups;
    int rep;  // 1: plain problem (every column swept)
};
// what the passes start from, made on the device instead of uploaded: D = E = 1, the row of every entry of A
struct RuizInitArgs {
    double* D; double* E; int64_t n, m;
    const int32_t* A_ptr; int32_t* arow;
};
__global__ __launch_bounds__(kThreads) void k_ruiz_init(RuizInitArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < a.n) a.D[i] = 1.0;
    if (i < a.m) {
        a.E[i] = 1.0;
        for (int k = a.A_ptr[i]; k < a.A_ptr[i + 1]; ++k) a.arow[k] = (int32_t)i;
    }
}
__device__ __forceinline__ int64_t ruiz_col(const RuizArgs& a, int64_t act) {
    return (a.rep <= 1 || act < a.nr) ? act : act + (int64_t)(a.rep - 1) * a.nr;
}
__global__ __launch_bounds__(kThreads) void k_ruiz_cols(RuizArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t act = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (act >= a.n_act) return;
    const int64_t j = ruiz_col(a, act);
    double mx = 0.0;
    for (int k = a.P_ptr[j] + lane; k < a.P_ptr[j + 1]; k += 64) mx = fmax(mx, fabs(a.P_val[k]) * a.D[a.P_col[k]]);
    for (int k = a.atp[j] + lane; k < a.atp[j + 1]; k += 64) {
        const int32_t q = a.atpos[k];
        mx = fmax(mx, fabs(a.A_val[q]) * a.E[a.arow[q]]);
    }
    mx = wave_max(mx);
    if (lane == 0) {
        mx *= a.D[j];
        a.d[act] = mx > 1e-12 ? 1.0 / sqrt(mx) : 1.0;
    }
}
__global__ __launch_bounds__(kThreads) void k_ruiz_groups(RuizArgs a) {
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (g >= a.ngroups) return;
    const int r0 = a.gstart[g];
    const int r1 = a.rep > 1 ? min(r0 + 2, a.gstart[g + 1]) : a.gstart[g + 1];
    double mx = 0.0;
    for (int k = a.A_ptr[r0]; k < a.A_ptr[r1]; ++k) mx = fmax(mx, fabs(a.A_val[k]) * a.D[a.A_col[k]]);
    mx *= a.E[r0];
    a.e[g] = mx > 1e-12 ? 1.0 / sqrt(mx) : 1.0;
}
__global__ __launch_bounds__(kThreads) void k_ruiz_apply(RuizArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < a.n_act) {
        const int64_t j = ruiz_col(a, i);
        const double v = a.D[j] * a.d[i];
        a.D[j] = v;
        if (a.rep > 1 && i < a.nr)
            for (int q = 1; q < a.rep; ++q) a.D[j + q * a.nr] = v;
    }
    if (i < a.ngroups) {
        const double eg = a.e[i];
        for (int r = a.gstart[i]; r < a.gstart[i + 1]; ++r) a.E[r] *= eg;
    }
}

// launch-overhead probes (debug timing only)
// after a factorisation: keep the factors to float precision (in place, so every reader sees the same
// operator) and write the 4-byte copy the LDS-resident chain kernel streams
__global__ __launch_bounds__(kThreads) void k_fac_round(double* __restrict__ fac, float* __restrict__ fac32, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float f = (float)fac[i];
    fac32[i] = f;
    fac[i] = (double)f;
}

// k_fac_round for the chains of a work list only: block (x, w) rounds a slice of work item w's factor range.  What a
// factorisation launch skipped (frozen problems, problems whose active set has not moved) is not touched.
__global__ __launch_bounds__(kThreads) void k_fac_round_items(const PrecWork* __restrict__ work, const int64_t* __restrict__ range,
                                                              const int32_t* skip, double* __restrict__ fac, float* __restrict__ fac32) {
    const PrecWork wk = work[blockIdx.y];
    if (wk.kind != 0 || (skip && skip[wk.prob])) return;
    const int64_t b = range[2 * (size_t)wk.index], e = range[2 * (size_t)wk.index + 1];
    for (int64_t i = b + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < e; i += (int64_t)gridDim.x * kThreads) {
        const float f = (float)fac[i];
        fac32[i] = f;
        fac[i] = (double)f;
    }
}

// Lane-major copy of the coarse-level factors of every chain of the work list (HostSystem::deep_map): block (q, w) fills
// 16-byte packet q of work item w's chain for the 256 staging lanes -- what k_prec_pre<.., float, true> loads straight into
// registers.  Slot groups are padded to multiples of 4 (deep_padded_slot, score_host.hpp); b2 = block size squared.
__global__ __launch_bounds__(kThreads) void k_deep_pack(const PrecWork* __restrict__ work, const ChainDesc* __restrict__ chains,
                                                        const ChainLevelDesc* __restrict__ levels, const int32_t* __restrict__ map,
                                                        const float* __restrict__ fac32, float* __restrict__ deep, const int32_t* skip, int b2) {
    const PrecWork wk = work[blockIdx.y];
    if (wk.kind != 0 || (skip && skip[wk.prob])) return;
    const ChainDesc ch = chains[wk.index];
    if (ch.deep_map_off < 0) return;
    const int64_t base = levels[ch.level_begin].offR;  // the chain's first factor entry
    const int q = blockIdx.x, dt = threadIdx.x;
    const int pad = ((2 * b2 + 3) & ~3) - 2 * b2, p1 = 2 * b2 + pad;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int g = 4 * q + c;
        // padded position -> slot (or a padding position: zero)
        int s = -1;
        if (g < p1) s = g < 2 * b2 ? g : -1;
        else if (g < p1 + 8 * b2) s = g - pad;
        else s = (g - 2 * pad < 12 * b2) ? g - 2 * pad : -1;
        const int32_t idx = s >= 0 ? map[(size_t)ch.deep_map_off + (size_t)s * kThreads + dt] : -1;
        v[c] = idx >= 0 ? fac32[base + idx] : 0.0f;
    }
    reinterpret_cast<float4*>(deep + ch.deep_off)[(size_t)q * kThreads + dt] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void k_nop(int* sink) { if (sink && threadIdx.x == 9999) sink[0] = 1; }
__global__ void k_nop_load(const int32_t* a, const int32_t* b2, int* sink) {
    const int p = a[blockIdx.x % 7];
    if (b2[p & 1] == 12345 && sink) sink[0] = 1;
}

}  // namespace score
