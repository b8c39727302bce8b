// score_polish_device.hpp -- the Newton matrix's pattern and contribution lists, built on the device.
//
// What score_polish_host.hpp::build_polish computes row by row on the host -- the pattern of H = P + A_tail' B A_tail on the
// union of P's entries and the cone couplings, P on that pattern, and for every entry the list of (cone, block index,
// coefficient) contributions in the order the host meets them -- from the matrices the handle has just uploaded (G2 = [P | A'],
// A, the head flags): every row EXPANDS into records (key = row << 32 | column; P entries first, then the contributions in
// the host loop's order: entries of A' of the row, tail index b, entries of A's row), a stable sort by key groups them (every row sorted where it lies:
// k_row_rank_sort, score_setup_device.hpp),
// one scan numbers the entries (runs of equal keys) and the contributions, one scatter writes Hcol, P-on-pattern, cptr and the
// lists.  Equal to the host build entry by entry and contribution by contribution (the sort is stable, the records are laid
// out in the host loop's order), so k_hassemble sums the same terms in the same order:
// score_debug_get("polish_build_check"), tests/test_gpu_parity.py::test_device_built_newton_matrix_equals_the_host_build.
// Replaces, for this part of score_create, the host loops that were the largest share of its CPU time (13.5 ms of 29 on the
// headline problem, single thread; 18 of 64 ms per 8-trial Monte-Carlo handle) -- SURVEY 8 f2, the reference builds the same
// couplings term by term in /root/reference/score/utils/gurobi_utils.py:336-352 (cones) and :449-501 (range cost).
#pragma once

#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "score_polish_host.hpp"

namespace score {

struct HBuildArgs {
    int64_t n;        // rows of H
    int32_t T;        // tail dimension: every cone has T + 1 rows of A, cone c owns rows c (T + 1) ..
    const int32_t* g2_ptr;    // G2 = [P | A'] by rows: P entries [g2_ptr[i], g2_split[i]), A' entries [g2_split[i], g2_ptr[i + 1])
    const int32_t* g2_split;
    const int32_t* g2_col;    // (A' entries: n + row of A)
    const double* g2_val;
    const int32_t* A_ptr;
    const int32_t* A_col;
    const double* A_val;
    const int32_t* is_head;
    long long* rec_cnt;       // n + 1: records per row, then (exclusive scan) first record of every row
    int64_t rec_max;          // room in the record arrays
    unsigned long long* key;
    uint32_t* idx;
    int32_t* rcone;           // >= 0: cone of a contribution; -1: a P entry; -2: the unit diagonal of a head row
    int32_t* rab;
    double* rcoef;
    // rows by length (round 5): a row of a dozen entries is served by EIGHT lanes (a wavefront per row left 56 lanes idle on
    // all but the few landmark rows); rows of more than kLongRowEntries entries of G2 are listed (k_row_classify) and served
    // by a wavefront each in a second launch of the same kernel
    const int32_t* long_rows; const int32_t* n_long_rows;
};
constexpr int kLongRowEntries = 128;
// rows of [ptr[i], ptr[i + 1]) longer than kLongRowEntries -> list (unordered), *cnt
__global__ __launch_bounds__(256) void k_row_classify(const int32_t* __restrict__ ptr, int64_t n, int32_t* __restrict__ list, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (ptr[i + 1] - ptr[i] > kLongRowEntries) list[atomicAdd(cnt, 1)] = (int32_t)i;
}
// the row a group of G lanes serves: G < 64 -- row = group index, long rows are left to the second launch; G == 64 -- the
// w-th long row.  Returns -1 when there is nothing to do.
template <int G>
__device__ __forceinline__ int64_t group_row(const int32_t* __restrict__ ptr, int64_t n, const int32_t* __restrict__ long_rows,
                                             const int32_t* __restrict__ n_long_rows) {
    const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (G == 64) {
        if (!long_rows) return g < n ? g : -1;   // (no list: a wavefront per row, every row)
        return g < *n_long_rows ? (int64_t)long_rows[g] : -1;
    }
    if (g >= n) return -1;
    return (ptr[g + 1] - ptr[g] > kLongRowEntries) ? -1 : g;
}
template <int G>
__device__ __forceinline__ long long group_sum(long long v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}
template <int G>
__device__ __forceinline__ bool group_any(bool pred) {
    const unsigned long long b = __ballot(pred);
    if (G == 64) return b != 0;
    const int lane = threadIdx.x & 63;
    return ((b >> (lane & ~(G - 1))) & ((1ull << G) - 1ull)) != 0;
}

// One WAVEFRONT per row (a landmark's row holds thousands of entries of A' and expands into tens of thousands of records;
// a pose row a dozen): the lanes stride over the row's entries, a wave scan places every entry's records.
__device__ inline long long hb_wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// records of row i (host loop: score_polish_host.hpp, build_polish)
template <int G>
__global__ __launch_bounds__(256) void k_hb_count(HBuildArgs a) {
    const int lane = threadIdx.x & (G - 1);
    if (G == 64 && blockIdx.x == 0 && threadIdx.x == 0) a.rec_cnt[a.n] = 0;
    const int64_t i = group_row<G>(a.g2_ptr, a.n, a.long_rows, a.n_long_rows);
    if (i < 0) return;
    if (a.is_head[i]) { if (lane == 0) a.rec_cnt[i] = 1; return; }
    const int p0 = a.g2_ptr[i], sp = a.g2_split[i], p1 = a.g2_ptr[i + 1];
    long long c = 0;
    int diag = 0;
    for (int k = p0 + lane; k < sp; k += G) diag |= (a.g2_col[k] == (int32_t)i);
    const int D1 = a.T + 1;
    for (int t = sp + lane; t < p1; t += G) {
        const int r = a.g2_col[t] - (int32_t)a.n;
        const int r0 = r / D1 * D1;
        if (r - r0 - 1 >= 0) c += a.A_ptr[r0 + 1 + a.T] - a.A_ptr[r0 + 1];  // (head rows only hold the head column)
    }
    c = group_sum<G>(c);
    const bool has_diag = group_any<G>(diag != 0);
    if (lane == 0) a.rec_cnt[i] = (long long)(sp - p0) + (has_diag ? 0 : 1) + c;
}

template <int G>
__global__ __launch_bounds__(256) void k_hb_expand(HBuildArgs a) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t i = group_row<G>(a.g2_ptr, a.n, a.long_rows, a.n_long_rows);
    if (i < 0) return;
    long long base = a.rec_cnt[i];
    const unsigned long long hi = (unsigned long long)i << 32;
    auto put = [&](long long o, int32_t j, int32_t cone, int32_t ab, double coef) {
        a.key[o] = hi | (unsigned long long)(uint32_t)j;
        a.idx[o] = (uint32_t)o;
        a.rcone[o] = cone; a.rab[o] = ab; a.rcoef[o] = coef;
    };
    if (a.is_head[i]) { if (lane == 0) put(base, (int32_t)i, -2, 0, 1.0); return; }
    const int p0 = a.g2_ptr[i], sp = a.g2_split[i], p1 = a.g2_ptr[i + 1];
    int diag = 0;
    for (int k = p0 + lane; k < sp; k += G) {
        put(base + (k - p0), a.g2_col[k], -1, 0, a.g2_val[k]);
        diag |= (a.g2_col[k] == (int32_t)i);
    }
    base += sp - p0;
    if (!group_any<G>(diag != 0)) {
        if (lane == 0) put(base, (int32_t)i, -1, 0, 0.0);
        ++base;
    }
    const int D1 = a.T + 1;
    for (int t0 = sp; t0 < p1; t0 += G) {
        const int t = t0 + lane;
        int cone = 0, ta = -1, r0 = 0;
        double vi = 0.0;
        long long mine = 0;
        if (t < p1) {
            const int r = a.g2_col[t] - (int32_t)a.n;
            vi = a.g2_val[t];
            cone = r / D1; r0 = cone * D1; ta = r - r0 - 1;
            if (ta >= 0) mine = a.A_ptr[r0 + 1 + a.T] - a.A_ptr[r0 + 1];
        }
        // exclusive prefix of `mine` over the lanes of the group (entries of A' in order)
        long long incl = mine;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const long long up = __shfl_up(incl, o, G);
            if (lane >= o) incl += up;
        }
        long long o = base + incl - mine;
        if (ta >= 0)
            for (int b = 0; b < a.T; ++b) {
                const int rb = r0 + 1 + b;
                for (int kk = a.A_ptr[rb]; kk < a.A_ptr[rb + 1]; ++kk, ++o) put(o, a.A_col[kk], cone, ta * a.T + b, vi * a.A_val[kk]);
            }
        base += __shfl(incl, G - 1, G);
    }
}

// the unused tail of the record arrays sorts behind every row: key = n << 32
__global__ __launch_bounds__(256) void k_hb_pad(HBuildArgs a) {
    const int64_t s = a.rec_cnt[a.n] + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    a.key[s] = (unsigned long long)a.n << 32;
    a.idx[s] = 0;
}

struct HScatterArgs {
    int64_t n, rec_max;
    const unsigned long long* key;   // sorted
    const uint32_t* idx;             // sorted: position of the record before the sort
    const int32_t* rcone;
    const int32_t* rab;
    const double* rcoef;
    unsigned long long* flag;        // per sorted record: (starts an entry) << 32 | (is a contribution); then its inclusive scan
    int32_t* Hcol;
    int32_t* Hrow;
    double* Pon;
    int32_t* cptr;
    int32_t* ccone;
    int32_t* cab;
    double* ccoef;
    int32_t* Hptr;                   // n + 1
    long long* result;               // [0] entries, [1] contributions, [2] long entries
    int32_t* long_ent;               // entries with more than kLongContrib contributions (unordered; the host sorts them)
    int32_t long_max;
    double diag_reg;
};

__global__ __launch_bounds__(256) void k_hb_flags(HScatterArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    unsigned long long f = 0;
    if ((int64_t)(k >> 32) < a.n) {
        if (s == 0 || a.key[s - 1] != k) f |= 1ull << 32;
        if (a.rcone[a.idx[s]] >= 0) f |= 1ull;
    }
    a.flag[s] = f;
}

__global__ __launch_bounds__(256) void k_hb_scatter(HScatterArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    const int64_t i = (int64_t)(k >> 32);
    if (i >= a.n) return;
    const unsigned long long S = a.flag[s];  // inclusive
    const int64_t e = (int64_t)(S >> 32) - 1;
    const uint32_t r = a.idx[s];
    const int32_t cone = a.rcone[r];
    const int64_t con_incl = (int64_t)(S & 0xffffffffull), con = con_incl - (cone >= 0 ? 1 : 0);
    const int32_t j = (int32_t)(uint32_t)(k & 0xffffffffull);
    const bool head = s == 0 || a.key[s - 1] != k;
    if (head) {
        a.Hcol[e] = j;
        a.Hrow[e] = (int32_t)i;
        a.cptr[e] = (int32_t)con;
        // (a P entry, when the key has one, is the first record of its run: P entries precede the contributions of a row)
        a.Pon[e] = cone == -2 ? 1.0 : (cone == -1 ? a.rcoef[r] + (j == (int32_t)i ? a.diag_reg : 0.0) : 0.0);
    }
    if (cone >= 0) {
        a.ccone[con] = cone;
        a.cab[con] = a.rab[r];
        a.ccoef[con] = a.rcoef[r];
    }
    const bool last = s + 1 == a.rec_max || (int64_t)(a.key[s + 1] >> 32) >= a.n;
    if (last) {
        a.result[0] = e + 1;
        a.result[1] = con_incl;
        a.cptr[e + 1] = (int32_t)con_incl;
    }
}

// row pointers of H, long entries (launched over the record bound; the entry count is read on the device)
__global__ __launch_bounds__(256) void k_hb_rows(HScatterArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nnz = a.result[0];
    if (e >= nnz) return;
    const int32_t row = a.Hrow[e];
    if (e == 0 || a.Hrow[e - 1] != row) a.Hptr[row] = (int32_t)e;
    if (e == nnz - 1) a.Hptr[a.n] = (int32_t)nnz;
    if (a.cptr[e + 1] - a.cptr[e] > kLongContrib) {
        const unsigned long long slot = atomicAdd((unsigned long long*)&a.result[2], 1ull);
        if ((int64_t)slot < a.long_max) a.long_ent[slot] = (int32_t)e;
    }
}

// positions of the chain blocks and of the Jacobi diagonals in H (find_in_row on the device)
struct HPosArgs {
    const int32_t* Hptr;
    const int32_t* Hcol;
    const int32_t* node_col;
    const int32_t* prev_col;   // column of a node's chain predecessor, -1 for the first node of a chain, -2: not this node's blocks (no look-up)
    int64_t n_nodes;
    int32_t bs;
    int32_t* pos_diag;
    int32_t* pos_sub;
    const int32_t* diag_cols;
    int64_t n_diag;
    int32_t* diag_pos;
};
__device__ inline int32_t hb_find(const int32_t* ptr, const int32_t* col, int32_t row, int32_t c) {
    int lo = ptr[row], hi = ptr[row + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < c) lo = mid + 1; else hi = mid;
    }
    return (lo < ptr[row + 1] && col[lo] == c) ? lo : -1;
}
__global__ __launch_bounds__(256) void k_hb_positions(HPosArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b2 = a.bs * a.bs;
    if (t < a.n_nodes * b2) {
        const int64_t g = t / b2;
        const int ab = (int)(t - g * b2), ra = ab / a.bs, cb = ab - ra * a.bs;
        const int32_t col = a.node_col[g];
        const int32_t pc = a.prev_col[g];
        a.pos_diag[t] = pc > -2 ? hb_find(a.Hptr, a.Hcol, col + ra, col + cb) : -1;
        a.pos_sub[t] = pc >= 0 ? hb_find(a.Hptr, a.Hcol, col + ra, pc + cb) : -1;
    }
    if (t < a.n_diag) a.diag_pos[t] = hb_find(a.Hptr, a.Hcol, a.diag_cols[t], a.diag_cols[t]);  // (diag_cols: the ROW that holds the diagonal)
}

// The cone tables k_cone reads by cone index (ConeArgs::cone_meta, cone_cols, cone_vals): row, dimension, kind and the row
// pointers of the cone's first four rows; for small cones (<= kSmallCone rows of <= kConeRowNnz entries) the entries of A
// again, 8 slots per cone, unused slots pointing at a valid column with value 0.  From A on the device (was: a host loop
// over the cones and 6 MB of uploads).
struct ConeTabArgs {
    int64_t ncones;
    const int32_t* cone_row;
    const int32_t* cone_dim;
    const int32_t* cone_type;
    const int32_t* A_ptr;
    const int32_t* A_col;
    const double* A_val;
    int4* meta;       // 2 per cone
    int32_t* cols;    // 8 per cone
    double* vals;     // 8 per cone
};
__global__ __launch_bounds__(256) void k_cone_tables(ConeTabArgs a) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.ncones) return;
    const int row = a.cone_row[c], dim = a.cone_dim[c];
    int ptr[kSmallCone + 1];
#pragma unroll
    for (int k = 0; k <= kSmallCone; ++k) ptr[k] = a.A_ptr[row + (k < dim ? k : dim)];
    a.meta[2 * c] = make_int4(row, dim, a.cone_type[c], ptr[0]);
    a.meta[2 * c + 1] = make_int4(ptr[1], ptr[2], ptr[3], ptr[4]);
    bool small = dim <= kSmallCone;
    for (int k = 0; k < kSmallCone && k < dim && small; ++k) small = (ptr[k + 1] - ptr[k]) <= kConeRowNnz;
    const int p_end = a.A_ptr[row + dim];
    const int32_t safe = ptr[0] < p_end ? a.A_col[ptr[0]] : 0;  // any valid column
#pragma unroll
    for (int k = 0; k < kSmallCone; ++k)
#pragma unroll
        for (int e = 0; e < kConeRowNnz; ++e) {
            const int64_t o = 8 * c + k * kConeRowNnz + e;
            int32_t cc = safe;
            double vv = 0.0;
            if (small && k < dim && ptr[k] + e < ptr[k + 1]) {
                cc = a.A_col[ptr[k] + e];
                vv = a.A_val[ptr[k] + e];
            }
            a.cols[o] = cc;
            a.vals[o] = vv;
        }
}

// The equilibrated A, G1 = A' (stored rows) and G2 = [P | A'] (all rows) DERIVED on the device from what the device
// equilibration holds there anyway -- the raw P (replica 0's rows and the tail's when the problem is replicated), the raw A,
// the A' position map, the final scales D and E -- instead of filled on the host and uploaded (26 MB for the headline
// problem).  Same expressions in the same order as the host ((v * d_i) * d_j, (v * e_r) * d_j): bit-equal to the host arrays
// (score_debug_get "ag_device_check").  A replicated problem's replica rows take replica 0's raw values: only when
// check_replication found them bit-equal (HostSystem::rep_exact).
struct DeriveArgs {
    int64_t n, m, nnzA;
    int32_t rep;              // 1: plain
    int64_t nr;               // unknowns per replica (rep > 1)
    const int32_t* P_ptr; const int32_t* P_col; const double* P_val;   // raw (rows of replica 0 and of the tail valid)
    const int32_t* A_ptr; const int32_t* A_col; const double* A_val;   // raw
    const int32_t* atp; const int32_t* atpos; const int32_t* arow;
    const double* D; const double* E;
    int32_t* oA_col; double* oA_val;                                    // equilibrated A (pattern copied)
    const int32_t* g1_ptr; int32_t* g1_col; double* g1_val;
    const int32_t* g2_ptr; const int32_t* g2_split; int32_t* g2_col; double* g2_val;
};
__global__ __launch_bounds__(256) void k_derive_a(DeriveArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.nnzA) return;
    const int32_t c = a.A_col[k];
    a.oA_col[k] = c;
    a.oA_val[k] = (a.A_val[k] * a.E[a.arow[k]]) * a.D[c];
}
// 1 / D and 1 / E (the solver's unscaling vectors) from the scales on the device
__global__ __launch_bounds__(256) void k_derive_inv(const double* D, const double* E, double* iD, double* iE, int64_t n, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) iD[i] = 1.0 / D[i];
    if (i < m) iE[i] = 1.0 / E[i];
}
// a wavefront per row i of G2 (= column i of A): P part, then the entries of A' (also into G1 when the row is stored)
__global__ __launch_bounds__(256) void k_derive_g(DeriveArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n) return;
    const bool in_rep = a.rep > 1 && i < (int64_t)a.rep * a.nr;
    const int64_t i0 = in_rep ? i % a.nr : i;
    const int32_t shift = (int32_t)(i - i0);
    const bool stored = a.rep <= 1 || i < a.nr || i >= (int64_t)a.rep * a.nr;
    const double di = a.D[i];
    const int k0 = a.P_ptr[i0], np = a.P_ptr[i0 + 1] - k0, o2 = a.g2_ptr[i];
    for (int l = lane; l < np; l += 64) {
        const int32_t c = a.P_col[k0 + l] + shift;
        a.g2_col[o2 + l] = c;
        a.g2_val[o2 + l] = (a.P_val[k0 + l] * di) * a.D[c];
    }
    const int t0 = a.atp[i], nt = a.atp[i + 1] - t0, s2 = a.g2_split[i], s1 = stored ? a.g1_ptr[i] : 0;
    for (int l = lane; l < nt; l += 64) {
        const int32_t q = a.atpos[t0 + l];
        const int32_t c = (int32_t)a.n + a.arow[q];
        const double v = a.oA_val[q];
        a.g2_col[s2 + l] = c;
        a.g2_val[s2 + l] = v;
        if (stored) { a.g1_col[s1 + l] = c; a.g1_val[s1 + l] = v; }
    }
}

// upper bound of the records (exact but for the rows of P without a diagonal entry): P entries, one record per head
// row, one per row for a missing diagonal, (entries of a cone's tail rows)^2 contributions per cone
inline int64_t polish_record_bound(const HostSystem& H, int T, int64_t* contributions = nullptr, int64_t nnzP_known = -1) {
    int64_t con = 0;
    for (size_t k = 0; k < H.cone_row.size(); ++k) {
        const int r0 = H.cone_row[k];
        const int64_t L = H.A.ptr[r0 + 1 + T] - H.A.ptr[r0 + 1];
        con += L * L;
    }
    if (contributions) *contributions = con;
    int64_t nnzP = nnzP_known >= 0 ? nnzP_known : 0;
    for (int64_t i = 0; i < H.n_tot && nnzP_known < 0; ++i) nnzP += H.g2_split[(size_t)i] - H.G2.ptr[(size_t)i];
    return nnzP + H.n_tot + con;
}

}  // namespace score
