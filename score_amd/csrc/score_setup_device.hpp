// score_setup_device.hpp -- score_create's matrices built on the device.
//
// What score_host.hpp::build_system computes on the host for every handle -- the A' position map, the Ruiz equilibration,
// the equilibrated A, G1 = A' and G2 = [P | A'], and K = P + sigma I + rho A'A as K0 + rho K1 on the union pattern -- from the
// RAW program (P, q, A, b as the caller or the device assembler hands them over), for a single problem and for the lock-step
// batch of a Monte-Carlo handle alike.  The batch is ONE block-diagonal system in global indices: every kernel below sees n_tot
// rows and m_tot constraint rows and looks the problem of a row up in a small table (ProbTab) where the row-replicated structure
// (score_problem::rep_d / rep_n, HostSystem::rep) asks for it.  The host keeps what is not a sweep over matrix entries: sizes,
// cone and chain tables, and -- from the row pointers and K's columns, which come back in one transfer -- the tile tables and
// the band layout.
//
// The host loops stay as the specification (and as the CPU twin's path): every array made here equals the host's entry by
// entry, the value arrays bit for bit -- the records of a row are laid out in the host loop's order, the sort is stable, and an
// entry adds its records one after the other in that order (score_debug_get "setup_device_check",
// tests/test_gpu_parity.py::test_device_setup_equals_the_host_setup).
//
// Reference: the model these matrices come from is built term by term in /root/reference/score/utils/gurobi_utils.py:173-187
// (initialize_model), :336-352 (cones), :358-526 (objective); Gurobi's presolve / ordering (gurobi_utils.py:206-215,
// solve_score.py:76) is what this setup stands in for.
#pragma once

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "score_host.hpp"
#include "score_polish_device.hpp"

namespace score {

// The problems of a handle: unknown and constraint-row offsets, unknowns per replica.  Device arrays (a handle may hold
// dozens of problems).
struct ProbTab {
    const int32_t* xoff;  // count + 1
    const int32_t* roff;  // count + 1
    const int32_t* nr;    // count: unknowns per replica (rep > 1)
    int32_t count, rep;   // rep: replicas of every problem of the handle (1: plain problems)
};
// largest p with off[p] <= i (off[0] = 0 <= i < off[count])
__device__ __forceinline__ int tab_find(const int32_t* __restrict__ off, int count, int64_t i) {
    int lo = 0, hi = count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
struct RowInfo {
    int prob;
    int64_t i0;      // the row of replica 0 that holds this row's entries (the row itself for plain / tail rows)
    int32_t shift;   // column shift from replica 0 to this row's replica
    bool stored;     // K and G1 hold a row for this unknown (replica 0 and the tail)
};
__device__ __forceinline__ RowInfo row_info(const ProbTab& t, int64_t i) {
    RowInfo r;
    r.prob = t.count > 1 ? tab_find(t.xoff, t.count, i) : 0;
    const int64_t x0 = t.xoff[r.prob], local = i - x0;
    const int64_t nr = t.rep > 1 ? t.nr[r.prob] : 0;
    const bool in_rep = t.rep > 1 && local < (int64_t)t.rep * nr;
    const int64_t l0 = in_rep ? local % nr : local;
    r.i0 = x0 + l0;
    r.shift = (int32_t)(local - l0);
    r.stored = !in_rep || local < nr;
    return r;
}

// column indices of a batch's problems from local to global: entry k belongs to the problem whose entry range holds it
__global__ __launch_bounds__(256) void k_globalise(int32_t* __restrict__ col, const int32_t* __restrict__ ent_off, const int32_t* __restrict__ delta,
                                                   int count, int64_t total) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= total) return;
    const int p = tab_find(ent_off, count, k);
    col[k] += delta[p];
}
__global__ __launch_bounds__(256) void k_iota(uint32_t* __restrict__ v, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) v[k] = (uint32_t)k;
}
// out[j] = first position of the sorted keys that is >= j, j = 0..n (column pointers of the A' map from the sorted columns)
__global__ __launch_bounds__(256) void k_lower_bounds(const uint32_t* __restrict__ sorted, int64_t cnt, int64_t n, int32_t* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    int64_t lo = 0, hi = cnt;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted[mid] < j) lo = mid + 1; else hi = mid;
    }
    out[j] = (int32_t)lo;
}

// ---- the Ruiz passes over the block-diagonal batch (ruiz_scale, score_host.hpp, is the specification) ----
struct RzArgs {
    const int32_t* P_ptr; const int32_t* P_col; const double* P_val;   // raw, global; rows of replicas >= 1 are not read
    const int32_t* A_ptr; const int32_t* A_col; const double* A_val;   // raw, global
    const int32_t* atp; const uint32_t* atpos; const int32_t* arow;
    const int32_t* gstart;   // first row of every cone group (the handle's cone table); group g ends at gstart[g + 1] / m
    double* D; double* E; double* d; double* e;
    double* cmax;                    // n: per column the maximum of its A part (k_rz_colsA), zero between passes
    const uint32_t* acol_sorted;     // the columns of A's entries in A'-map order
    int64_t n, m, ngroups, nnzA;
    ProbTab tab;
};
// Column norms in two kernels.  The A part runs over the entries of A in COLUMN order (the A' map: a landmark's column holds
// thousands of entries, a pose column two or three): a lane per entry, |v| E[row], a segmented maximum over the lanes of a
// wavefront by column, and the last lane of every run folds its maximum into the column's cell with an integer atomic max on the
// bit pattern (non-negative doubles order like their bits; a maximum does not depend on the order: deterministic).  The P part
// and the scale itself: eight lanes per column (a row of P holds a dozen entries).  cmax is zeroed by k_rz_apply.
__device__ __forceinline__ void rz_groups_body(const RzArgs& a, int64_t g);
// (one launch: the first blocks_a blocks sweep A's entries, the rest the cone groups -- both read the scales of the pass before
//  and nothing of each other; four launches per pass were three quarters launch shell)
__global__ __launch_bounds__(256) void k_rz_colsA(RzArgs a, int blocks_a) {
    if ((int)blockIdx.x >= blocks_a) { rz_groups_body(a, (int64_t)((int)blockIdx.x - blocks_a) * 256 + threadIdx.x); return; }
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = k < a.nnzA;
    const uint32_t col = on ? a.acol_sorted[k] : 0xffffffffu;
    double v = 0.0;
    if (on) {
        const uint32_t q = a.atpos[k];
        v = fabs(a.A_val[q]) * a.E[a.arow[q]];
    }
    // segmented inclusive max-scan over the wavefront (runs of equal columns are contiguous)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(v, o, 64);
        const uint32_t cu = __shfl_up(col, o, 64);
        if (lane >= o && cu == col) v = fmax(v, up);
    }
    const uint32_t cn = __shfl_down(col, 1, 64);
    if (on && (lane == 63 || cn != col)) atomicMax((unsigned long long*)&a.cmax[col], (unsigned long long)__double_as_longlong(v));
}
__global__ __launch_bounds__(256) void k_rz_cols(RzArgs a) {
    const int sub = threadIdx.x & 7;
    const int64_t j = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (j >= a.n) return;
    const RowInfo ri = row_info(a.tab, j);
    if (!ri.stored) return;
    double mx = 0.0;
    for (int k = a.P_ptr[j] + sub; k < a.P_ptr[j + 1]; k += 8) mx = fmax(mx, fabs(a.P_val[k]) * a.D[a.P_col[k]]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 8));
    if (sub == 0) {
        mx = fmax(mx, a.cmax[j]);
        mx *= a.D[j];
        a.d[j] = mx > 1e-12 ? 1.0 / sqrt(mx) : 1.0;
    }
}
__device__ __forceinline__ void rz_groups_body(const RzArgs& a, int64_t g) {
    if (g >= a.ngroups) return;
    const int r0 = a.gstart[g];
    const int rend = g + 1 < a.ngroups ? a.gstart[g + 1] : (int)a.m;
    // (replicated: the head row and the first tail row stand for the whole cone)
    const int r1 = a.tab.rep > 1 ? min(r0 + 2, rend) : rend;
    double mx = 0.0;
    for (int k = a.A_ptr[r0]; k < a.A_ptr[r1]; ++k) mx = fmax(mx, fabs(a.A_val[k]) * a.D[a.A_col[k]]);
    mx *= a.E[r0];
    a.e[g] = mx > 1e-12 ? 1.0 / sqrt(mx) : 1.0;
}
__global__ __launch_bounds__(256) void k_rz_apply(RzArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) {
        a.cmax[i] = 0.0;
        const RowInfo ri = row_info(a.tab, i);
        if (ri.stored) {
            const double v = a.D[i] * a.d[i];
            a.D[i] = v;
            if (a.tab.rep > 1 && ri.i0 == i && i - a.tab.xoff[ri.prob] < a.tab.nr[ri.prob]) {
                const int64_t nr = a.tab.nr[ri.prob];
                for (int q = 1; q < a.tab.rep; ++q) a.D[i + q * nr] = v;
            }
        }
    }
    if (i < a.ngroups) {
        const double eg = a.e[i];
        const int rend = i + 1 < a.ngroups ? a.gstart[i + 1] : (int)a.m;
        for (int r = a.gstart[i]; r < rend; ++r) a.E[r] *= eg;
    }
}

// ---- G1 = A' (stored rows), G2 = [P | A'] (all rows), the equilibrated A, q, b, 1/D, 1/E ----
struct GDevArgs {
    int64_t n, m, nnzA;
    ProbTab tab;
    const int32_t* P_ptr; const int32_t* P_col; const double* P_val;   // raw
    const int32_t* A_col; const double* A_val;                          // raw
    const int32_t* atp; const uint32_t* atpos; const int32_t* arow;
    const double* D; const double* E;
    long long* len1; long long* len2;                                    // n + 1 each: row lengths, then (exclusive scan) row starts
    int32_t* g1_ptr; int32_t* g2_ptr; int32_t* g2_split;
    int32_t* oA_col; double* oA_val;
    int32_t* g1_col; double* g1_val; int32_t* g2_col; double* g2_val;
    const double* q_raw; const double* b_raw; double* q; double* b; double* invD; double* invE;
    const int32_t* long_rows; const int32_t* n_long_rows;   // rows of G2 beyond kLongRowEntries (k_row_classify): a wavefront each
};
__global__ __launch_bounds__(256) void k_g_lengths(GDevArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > a.n) return;
    if (i == a.n) { a.len1[i] = 0; a.len2[i] = 0; return; }
    const RowInfo ri = row_info(a.tab, i);
    const int np = a.P_ptr[ri.i0 + 1] - a.P_ptr[ri.i0], nt = a.atp[i + 1] - a.atp[i];
    a.len1[i] = ri.stored ? nt : 0;
    a.len2[i] = np + nt;
}
__global__ __launch_bounds__(256) void k_g_ptrs(GDevArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > a.n) return;
    a.g1_ptr[i] = (int32_t)a.len1[i];
    a.g2_ptr[i] = (int32_t)a.len2[i];
}
__global__ __launch_bounds__(256) void k_g_scale_a(GDevArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < a.nnzA) {
        const int32_t c = a.A_col[k];
        a.oA_col[k] = c;
        a.oA_val[k] = (a.A_val[k] * a.E[a.arow[k]]) * a.D[c];
    }
    if (k < 64) { a.oA_col[a.nnzA + k] = 0; a.oA_val[a.nnzA + k] = 0.0; }  // (the cone kernel's clamped loads land here)
    if (k < a.n) { const double d = a.D[k]; a.q[k] = a.q_raw[k] * d; a.invD[k] = 1.0 / d; }
    if (k < a.m) { const double e = a.E[k]; a.b[k] = a.b_raw[k] * e; a.invE[k] = 1.0 / e; }
}
// G lanes per row i of G2 (= column i of A): P part, then the entries of A' (also into G1 when the row is stored); eight
// lanes for the usual rows of a dozen entries, a wavefront for the listed long ones (landmark rows)
template <int G>
__global__ __launch_bounds__(256) void k_g_fill(GDevArgs a) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t i = group_row<G>(a.g2_ptr, a.n, a.long_rows, a.n_long_rows);
    if (i < 0) return;
    const RowInfo ri = row_info(a.tab, i);
    const double di = a.D[i];
    const int k0 = a.P_ptr[ri.i0], np = a.P_ptr[ri.i0 + 1] - k0, o2 = a.g2_ptr[i];
    for (int l = lane; l < np; l += G) {
        const int32_t c = a.P_col[k0 + l] + ri.shift;
        a.g2_col[o2 + l] = c;
        a.g2_val[o2 + l] = (a.P_val[k0 + l] * di) * a.D[c];
    }
    const int t0 = a.atp[i], nt = a.atp[i + 1] - t0, s2 = o2 + np, s1 = ri.stored ? a.g1_ptr[i] : 0;
    if (lane == 0) a.g2_split[i] = s2;
    for (int l = lane; l < nt; l += G) {
        const uint32_t q = a.atpos[t0 + l];
        const int32_t c = (int32_t)a.n + a.arow[q];
        const double v = a.oA_val[q];
        a.g2_col[s2 + l] = c;
        a.g2_val[s2 + l] = v;
        if (ri.stored) { a.g1_col[s1 + l] = c; a.g1_val[s1 + l] = v; }
    }
}
// per problem: |q|_inf and |b|_inf, unscaled and scaled (the scales of the residual tests).  gridDim.y = problem, gridDim.x
// workgroups share its entries and fold their maxima in with integer atomic maxima on the bit patterns (`out` zeroed first).
__global__ __launch_bounds__(256) void k_prob_norms(ProbTab tab, const double* __restrict__ q_raw, const double* __restrict__ q,
                                                    const double* __restrict__ b_raw, const double* __restrict__ b, double* __restrict__ out) {
    const int p = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[4] = {0, 0, 0, 0};
    for (int64_t i = tab.xoff[p] + t0; i < tab.xoff[p + 1]; i += stride) { v[0] = fmax(v[0], fabs(q_raw[i])); v[1] = fmax(v[1], fabs(q[i])); }
    for (int64_t r = tab.roff[p] + t0; r < tab.roff[p + 1]; r += stride) { v[2] = fmax(v[2], fabs(b_raw[r])); v[3] = fmax(v[3], fabs(b[r])); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[c] = fmax(v[c], __shfl_xor(v[c], o, 64));
        if ((threadIdx.x & 63) == 0 && v[c] > 0.0) atomicMax((unsigned long long*)&out[4 * p + c], (unsigned long long)__double_as_longlong(v[c]));
    }
}
// x = xhat * D, y = yhat * E, s = shat / E into a staging buffer [x | y | s] (score_solve's copy-out)
__global__ __launch_bounds__(256) void k_unscale(const double* __restrict__ xy, const double* __restrict__ s, const double* __restrict__ D,
                                                 const double* __restrict__ E, double* __restrict__ out, int64_t n, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = xy[i] * D[i];
    if (i < m) { out[n + i] = xy[n + i] * E[i]; out[n + m + i] = s[i] / E[i]; }
}

// ---- records -> CSR with summed values: the merge step shared by the builders of K (and of P, device assembler) ----
// A builder lays records out in the host loop's order: key = row << 32 | column (row = n: padding, sorts last), two values
// per record.  After the stable sort by key (HipBackend::sort_rows: row by row, see k_row_rank_sort below) an entry is a run of equal keys; it adds its records ONE AFTER THE OTHER in that
// order (what the host's merge does) -- a run of more than kLongRun records is handed to a wavefront (k_rec_long).
constexpr int kLongRun = 48;
struct RecArgs {
    int64_t n_rows, rec_max;
    const unsigned long long* key;   // sorted
    const uint32_t* idx;             // sorted: position of the record before the sort
    const double* v0; const double* v1;   // by position before the sort (v1 may be null)
    unsigned long long* flag;        // per sorted record: starts an entry; then its inclusive scan
    int32_t* row_cnt;                // n_rows + 1, zeroed: entries per row
    int32_t* col; double* o0; double* o1;   // per entry (o1 may be null)
    long long* result;               // [0] entries, [1] long runs
    int4* long_run;                  // {first sorted record, length, entry, 0}
    int32_t long_max;
    int32_t qcol;                    // >= 0: a record of this column is the row's entry of a dense vector (qout), not of the matrix
    double* qout;
};
__global__ __launch_bounds__(256) void k_rec_flags(RecArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    unsigned long long f = 0;
    if ((int64_t)(k >> 32) < a.n_rows && (int32_t)(uint32_t)(k & 0xffffffffull) != a.qcol && (s == 0 || a.key[s - 1] != k)) f = 1;
    a.flag[s] = f;
}
__global__ __launch_bounds__(256) void k_rec_merge(RecArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    const int64_t row = (int64_t)(k >> 32);
    if (row >= a.n_rows) return;
    if (s > 0 && a.key[s - 1] == k) return;  // (not the first record of its run)
    const int32_t j = (int32_t)(uint32_t)(k & 0xffffffffull);
    int64_t len = 1;
    while (s + len < a.rec_max && a.key[s + len] == k && len <= kLongRun) ++len;
    const bool is_q = a.qcol >= 0 && j == a.qcol;
    const int64_t e = is_q ? -1 : (int64_t)a.flag[s] - 1;
    if (!is_q) {
        a.col[e] = j;
        atomicAdd(&a.row_cnt[row], 1);
    }
    if (len > kLongRun) {
        const unsigned long long slot = atomicAdd((unsigned long long*)&a.result[1], 1ull);
        if ((int64_t)slot < a.long_max) a.long_run[slot] = make_int4((int)s, is_q ? -1 : 0, (int)(is_q ? row : e), 0);
        return;
    }
    double s0 = 0.0, s1 = 0.0;
    for (int64_t t = 0; t < len; ++t) {
        const uint32_t r = a.idx[s + t];
        s0 += a.v0[r];
        if (a.v1) s1 += a.v1[r];
    }
    if (is_q) a.qout[row] = s0;
    else {
        a.o0[e] = s0;
        if (a.o1) a.o1[e] = s1;
    }
}
// total number of entries: the scan's last value; the 64 entries behind the last one read as (column 0, value 0) (launched
// with 64 threads)
__global__ void k_rec_total(RecArgs a) {
    const long long tot = a.rec_max > 0 ? (long long)a.flag[a.rec_max - 1] : 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) a.result[0] = tot;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        a.col[tot + threadIdx.x] = 0;
        a.o0[tot + threadIdx.x] = 0.0;
        if (a.o1) a.o1[tot + threadIdx.x] = 0.0;
    }
}
// one wavefront per long run: 64 records are fetched together, lane 0's order of addition is the records' order
__global__ __launch_bounds__(256) void k_rec_long(RecArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_long = a.result[1] < (long long)a.long_max ? a.result[1] : (long long)a.long_max;
    if (w >= n_long) return;
    const int4 lr = a.long_run[w];
    const int64_t s = lr.x;
    const unsigned long long k = a.key[s];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t base = s;; base += 64) {
        const int64_t t = base + lane;
        const bool mine = t < a.rec_max && a.key[t] == k;
        double x0 = 0.0, x1 = 0.0;
        if (mine) {
            const uint32_t r = a.idx[t];
            x0 = a.v0[r];
            if (a.v1) x1 = a.v1[r];
        }
        const unsigned long long live = __ballot(mine);
        const int cnt = __popcll(live);  // (a run is contiguous: the live lanes are 0 .. cnt - 1)
        for (int l = 0; l < cnt; ++l) {
            s0 += __shfl(x0, l, 64);
            s1 += __shfl(x1, l, 64);
        }
        if (cnt < 64) break;
    }
    if (lane == 0) {
        if (lr.y < 0) a.qout[lr.z] = s0;
        else {
            a.o0[lr.z] = s0;
            if (a.o1) a.o1[lr.z] = s1;
        }
    }
}
__global__ __launch_bounds__(256) void k_cnt_to_ll(const int32_t* __restrict__ cnt, long long* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = cnt[i];
}
__global__ __launch_bounds__(256) void k_ll_to_i32(const long long* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}
__global__ __launch_bounds__(256) void k_rec_pad(unsigned long long* __restrict__ key, uint32_t* __restrict__ idx, const long long* __restrict__ used,
                                                 int64_t rec_max, int64_t n_rows) {
    const int64_t s = used[0] + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= rec_max) return;
    key[s] = (unsigned long long)n_rows << 32;
    idx[s] = 0;
}

// ---- records laid out row by row (the K and Newton-matrix builders: every row's records are contiguous, the rows in order) ----
// The stable sort by (row, column) is then a sort of every row's records by (column, position): position is unique, so whatever a
// sort does with ties the order is the stable one.  Round 5 sent every record through seven onesweep passes of a 50-bit key (a third
// of a create's kernel time, two runtime fills per pass).  Round 6: a row of at most kShortRow records is sorted by RANK -- a lane
// per record counts the row's records that come before its own (the row's keys are read by all of its lanes at once: broadcasts out
// of L1) and writes the record to its place; the few long rows (landmark rows: thousands of records) are listed and go through a
// segmented radix sort (rocprim: a block per row) with key = column << pbits | position, pbits = the bits of rec_max.
// (rocprim's segmented sort for ALL rows was tried first: its warp-sort kernel for small segments took 246 us for 218 k rows.)
constexpr int kShortRow = 128;
struct RowSortArgs {
    const unsigned long long* key;   // records, row << 32 | column (tail: row n_rows)
    const uint32_t* idx;             // position of a record before it was put into its row's slots (nullptr: where it lies)
    const long long* off;            // n_rows + 1 record offsets
    unsigned long long* key_out; uint32_t* idx_out;
    unsigned long long* sk;          // sort keys of the long rows' records (in: k_row_rank_sort writes, out: k_row_keys_back reads)
    int32_t* seg_b; int32_t* seg_e; int32_t* seg_n; int32_t seg_cap;   // the long rows as segments (unused entries stay [0, 0))
    int64_t rec_max, n_rows;
    int pbits;
};
__global__ __launch_bounds__(256) void k_row_rank_sort(RowSortArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    const int64_t row = (int64_t)(k >> 32);
    if (row >= a.n_rows) { a.key_out[s] = k; a.idx_out[s] = 0; return; }  // the padding tail
    const int64_t b = a.off[row], e = a.off[row + 1];
    const uint32_t mine = (uint32_t)k;
    const uint32_t me = a.idx ? a.idx[s] : (uint32_t)s;
    if (e - b > kShortRow) { a.sk[s] = ((unsigned long long)mine << a.pbits) | (unsigned long long)me; return; }
    const uint32_t* __restrict__ cols = reinterpret_cast<const uint32_t*>(a.key);  // (little-endian: the column is the low word)
    int rank = 0;
    for (int64_t j0 = b; j0 < e; j0 += 4) {
        uint32_t c[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = min(j0 + u, e - 1);
            c[u] = cols[2 * j];
            o[u] = a.idx ? a.idx[j] : (uint32_t)j;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rank += (j0 + u < e && (c[u] < mine || (c[u] == mine && o[u] < me))) ? 1 : 0;
    }
    a.key_out[b + rank] = k;
    a.idx_out[b + rank] = me;
}
// records in no particular order (the assembler's: measurement by measurement) into their rows' slots: count, scan, scatter -- the
// order inside a row is whatever the atomics make it; the sort by (column, original position) that follows does not depend on it
// (the padding records -- switched-off slots anywhere in the array, all of row n_rows -- take no part: a hundred thousand atomics on
//  ONE counter took 3 ms; the tail of the row-ordered arrays is padded afterwards, k_rec_pad)
__global__ __launch_bounds__(256) void k_rec_count(const unsigned long long* __restrict__ key, int64_t rec_max, int64_t n_rows, unsigned long long* __restrict__ cnt) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= rec_max) return;
    const int64_t row = (int64_t)(key[s] >> 32);
    if (row < n_rows) atomicAdd(&cnt[row], 1ull);
}
__global__ __launch_bounds__(256) void k_rec_scatter(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ idx, int64_t rec_max, int64_t n_rows,
                                                     const long long* __restrict__ off, unsigned int* __restrict__ cur,
                                                     unsigned long long* __restrict__ key_b, uint32_t* __restrict__ idx_b) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= rec_max) return;
    const unsigned long long k = key[s];
    const int64_t row = (int64_t)(k >> 32);
    if (row >= n_rows) return;
    const int64_t at = off[row] + (int64_t)atomicAdd(&cur[row], 1u);
    key_b[at] = k;
    idx_b[at] = idx[s];
}
__global__ __launch_bounds__(256) void k_rows_long_list(RowSortArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    const int64_t b = a.off[r], e = a.off[r + 1];
    if (e - b <= kShortRow) return;
    const int at = atomicAdd(a.seg_n, 1);
    if (at < a.seg_cap) { a.seg_b[at] = (int32_t)b; a.seg_e[at] = (int32_t)e; }
}
// ... and back, for the long rows: slot s keeps its row (a row's records stay in the row's slots)
__global__ __launch_bounds__(256) void k_row_keys_back(RowSortArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rec_max) return;
    const unsigned long long k = a.key[s];
    const int64_t row = (int64_t)(k >> 32);
    if (row >= a.n_rows || a.off[row + 1] - a.off[row] <= kShortRow) return;
    const unsigned long long v = a.sk[s];
    a.key_out[s] = (k & 0xffffffff00000000ull) | (v >> a.pbits);
    a.idx_out[s] = (uint32_t)(v & ((1ull << a.pbits) - 1ull));
}

// ---- K = P + sigma I + rho A'A as K0 + rho K1: the records of every stored row (append_problem, score_host.hpp) ----
struct KBuildArgs {
    int64_t n;
    ProbTab tab;
    double sigma;
    const int32_t* g2_ptr; const int32_t* g2_split; const int32_t* g2_col; const double* g2_val;  // equilibrated [P | A']
    const int32_t* A_ptr; const int32_t* A_col; const double* A_val;                                // equilibrated A, global
    long long* rec_cnt;      // n + 1: records per row, then (exclusive scan) first record of every row
    unsigned long long* key; uint32_t* idx; double* v0; double* v1;
    const int32_t* long_rows; const int32_t* n_long_rows;   // as GDevArgs
};
template <int G>
__global__ __launch_bounds__(256) void k_kb_count(KBuildArgs a) {
    const int lane = threadIdx.x & (G - 1);
    if (G == 64 && blockIdx.x == 0 && threadIdx.x == 0) a.rec_cnt[a.n] = 0;
    const int64_t i = group_row<G>(a.g2_ptr, a.n, a.long_rows, a.n_long_rows);
    if (i < 0) return;
    const RowInfo ri = row_info(a.tab, i);
    if (!ri.stored) { if (lane == 0) a.rec_cnt[i] = 0; return; }
    const int p0 = a.g2_ptr[i], sp = a.g2_split[i], p1 = a.g2_ptr[i + 1];
    long long c = 0;
    for (int t = sp + lane; t < p1; t += G) {
        const int r = a.g2_col[t] - (int32_t)a.n;
        c += a.A_ptr[r + 1] - a.A_ptr[r];
    }
    c = group_sum<G>(c);
    if (lane == 0) a.rec_cnt[i] = 1 + (long long)(sp - p0) + c;
}
template <int G>
__global__ __launch_bounds__(256) void k_kb_expand(KBuildArgs a) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t i = group_row<G>(a.g2_ptr, a.n, a.long_rows, a.n_long_rows);
    if (i < 0) return;
    const RowInfo ri = row_info(a.tab, i);
    if (!ri.stored) return;
    long long base = a.rec_cnt[i];
    const unsigned long long hi = (unsigned long long)i << 32;
    auto put = [&](long long o, int32_t j, double x0, double x1) {
        a.key[o] = hi | (unsigned long long)(uint32_t)j;
        a.idx[o] = (uint32_t)o;
        a.v0[o] = x0; a.v1[o] = x1;
    };
    if (lane == 0) put(base, (int32_t)i, a.sigma, 0.0);
    ++base;
    const int p0 = a.g2_ptr[i], sp = a.g2_split[i], p1 = a.g2_ptr[i + 1];
    for (int k = p0 + lane; k < sp; k += G) put(base + (k - p0), a.g2_col[k], a.g2_val[k], 0.0);
    base += sp - p0;
    for (int t0 = sp; t0 < p1; t0 += G) {
        const int t = t0 + lane;
        int r = 0;
        double av = 0.0;
        long long mine = 0;
        if (t < p1) {
            r = a.g2_col[t] - (int32_t)a.n;
            av = a.g2_val[t];
            mine = a.A_ptr[r + 1] - a.A_ptr[r];
        }
        long long incl = mine;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const long long up = __shfl_up(incl, o, G);
            if (lane >= o) incl += up;
        }
        long long o = base + incl - mine;
        if (t < p1)
            for (int kk = a.A_ptr[r]; kk < a.A_ptr[r + 1]; ++kk, ++o) put(o, a.A_col[kk], 0.0, av * a.A_val[kk]);
        base += __shfl(incl, G - 1, G);
    }
}

// ---- the structure the Newton polish needs per cone (polish_structure, score_polish_host.hpp), from the device matrices ----
struct PStructArgs {
    int64_t ncones, n;
    int32_t T;
    const int32_t* cone_row; const int32_t* cone_dim; const int32_t* cone_type;
    const int32_t* A_ptr; const int32_t* A_col; const double* A_val; const double* b; const double* q;
    const int32_t* g2_ptr; const int32_t* g2_split; const int32_t* g2_col; const double* g2_val;
    int32_t* head_col; int32_t* is_head; double* a_abs; double* ck; double* theta; double* xstar;
    int32_t* bad;      // [0]: cones that do not have the structure
};
__global__ __launch_bounds__(256) void k_polish_structure(PStructArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.ncones) return;
    bool ok = a.cone_type[k] == 1 && a.cone_dim[k] - 1 == a.T;
    const int r0 = a.cone_row[k];
    ok = ok && (a.A_ptr[r0 + 1] - a.A_ptr[r0] == 1);
    int32_t h = 0;
    double av = -1.0, phh = 1.0;
    if (ok) {
        h = a.A_col[a.A_ptr[r0]];
        av = a.A_val[a.A_ptr[r0]];
        ok = (av < 0.0) && a.b[r0] == 0.0;
        ok = ok && (a.g2_ptr[h + 1] - a.g2_split[h] == 1) && (a.g2_split[h] - a.g2_ptr[h] == 1) && a.g2_col[a.g2_ptr[h]] == h;
        if (ok) {
            phh = a.g2_val[a.g2_ptr[h]];
            ok = phh > 0.0;
        }
    }
    if (!ok) { atomicAdd(a.bad, 1); return; }
    a.head_col[k] = h;
    a.is_head[h] = 1;
    a.a_abs[k] = -av;
    a.ck[k] = phh / (av * av);
    const double xs = -a.q[h] / phh;
    a.xstar[k] = xs;
    a.theta[k] = -av * xs;
}
// sum over the cones of (entries of the cone's tail rows)^2: the contribution bound of the Newton matrix's records
__global__ __launch_bounds__(256) void k_cone_sq(const int32_t* __restrict__ cone_row, const int32_t* __restrict__ A_ptr, int64_t ncones, int T,
                                                 unsigned long long* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long v = 0;
    if (k < ncones) {
        const int r0 = cone_row[k];
        const unsigned long long L = (unsigned long long)(A_ptr[r0 + 1 + T] - A_ptr[r0 + 1]);
        v = L * L;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// ---------------------------------------------------------------------------------------------------------------------
// Model construction on the device: the factor graphs as flat arrays (score_graph) -> the raw program (P, q, A, b) in the
// handle's global numbering, as records for the merge above.  score_assemble.hpp::assemble_graph is the specification:
// every measurement writes the records assemble_graph's filling pass adds, in that order (slots the host skips -- a pinned
// pose, an exactly-zero entry of the measurement's matrix -- hold the padding key), so P and q come out bit-equal to the host
// assembler's.  Only replica 0's rows of P and the tail's are built (P = I_d (x) P_row + tail); q gets every replica's terms.
// Reference: /root/reference/score/utils/gurobi_utils.py:504-526 (relative-pose cost), :449-501 (range cost), :433-446
// (landmark priors), :336-352 (cones), :316-333 (pinned first pose: eliminated, its terms move to q and c0).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int32_t kQCol = 0x7fffffff;  // "column" of a record that belongs to q
struct GaProb {          // one problem of the handle
    int32_t xoff, roff;  // first unknown / first constraint row (global)
    int32_t Np, Nl, Nr;  // poses (pose 0 is the pinned one), landmarks, ranges
    int32_t n_rep;       // unknowns per replica
    int32_t rel_off, rng_off, pri_off, pin_off;   // first relative-pose measurement / range / prior / pinned-edge entry of the problem
    int32_t rec_rel, rec_rng, rec_pri;             // first P-record slot of its relative-pose measurements / ranges / priors
    int32_t recq_pin, recq_rng, recq_pri;          // first q-record slot of its pinned edges / ranges / priors
    int32_t a_off;                                  // first entry of its rows of A
};
struct GaArgs {
    int32_t d, relaxation, count;
    const GaProb* probs;
    const int32_t* rel_prob; const int32_t* rng_prob; const int32_t* pri_prob; const int32_t* pin_prob;  // measurement -> problem (sorted: tab_find over offsets)
    int64_t n_rel, n_rng, n_pri, n_pin;
    const int32_t* rel_base; const int32_t* rel_to; const double* rel_t; const double* rel_R; const double* rel_kappa; const double* rel_tau;
    const int32_t* rng_a; const int32_t* rng_b; const double* rng_dist; const double* rng_prec;
    const int32_t* pri_lm; const double* pri_t; const double* pri_prec;
    const int32_t* pin_edge;     // relative-pose measurements (global index) that touch a pinned pose
    unsigned long long* key; uint32_t* idx; double* val;     // records
    unsigned long long pad_key;                              // (n_tot << 32)
    const int32_t* A_ptr; int32_t* A_col; double* A_val; double* b;   // raw A (global), b
};
// solver column (local) of entry j of row k of pose p; -1 for the pinned pose
__device__ __forceinline__ int64_t ga_pcol(const GaProb& P, int d, int64_t p, int k, int j) {
    return p == 0 ? -1 : (int64_t)k * P.n_rep + (p - 1) * (d + 1) + j;
}
__device__ __forceinline__ int64_t ga_tcol(const GaProb& P, int d, int64_t v, int k) {
    if (v < P.Np) return ga_pcol(P, d, v, k, d);
    return (int64_t)k * P.n_rep + (int64_t)(P.Np - 1) * (d + 1) + (v - P.Np);
}
struct GaEdge {  // G, W, G'W, G'WG of one relative-pose measurement (assemble_graph's locals)
    double G[4][4], W[4], GtW[4][4], GtWG[4][4];
};
__device__ __forceinline__ void ga_edge_blocks(const GaArgs& a, int64_t e, GaEdge& E) {
#pragma clang fp contract(off)  // (the host assembler's arithmetic, operation by operation: no fused multiply-adds)
    const int d = a.d, D1 = d + 1;
    const double* tm = a.rel_t + e * d;
    const double* Rm = a.rel_R + e * d * d;
    const double kap = a.rel_kappa[e], tau = a.rel_tau[e];
    for (int c = 0; c < 4; ++c)
        for (int l = 0; l < 4; ++l) E.G[c][l] = 0.0;
    for (int c = 0; c < d; ++c) {
        for (int l = 0; l < d; ++l) E.G[c][l] = Rm[l * d + c];
        E.W[c] = tau;
    }
    for (int l = 0; l < d; ++l) E.G[d][l] = tm[l];
    E.G[d][d] = 1.0;
    E.W[d] = kap;
    for (int x = 0; x < D1; ++x)
        for (int y = 0; y < D1; ++y) {
            E.GtW[x][y] = E.G[y][x] * E.W[y];
            double s_ = 0;
            for (int c = 0; c < D1; ++c) s_ += E.G[c][x] * E.W[c] * E.G[c][y];
            E.GtWG[x][y] = s_;
        }
}
// P records of the relative-pose measurements: D1 + 3 D1^2 slots each, in assemble_graph's order (k = 0)
__global__ __launch_bounds__(256) void k_ga_rel(GaArgs a) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_rel) return;
    const GaProb P = a.probs[a.rel_prob[e]];
    const int d = a.d, D1 = d + 1, per = D1 + 3 * D1 * D1;
    const int64_t i = a.rel_base[e], j = a.rel_to[e];
    GaEdge E;
    ga_edge_blocks(a, e, E);
    int64_t o = (int64_t)P.rec_rel + (e - P.rel_off) * per;
    const int64_t ci = ga_pcol(P, d, i, 0, 0), cj = ga_pcol(P, d, j, 0, 0);
    auto put = [&](bool on, int64_t r, int64_t c, double v) {
        a.key[o] = on ? (((unsigned long long)(P.xoff + r)) << 32 | (unsigned long long)(uint32_t)(P.xoff + c)) : a.pad_key;
        a.idx[o] = (uint32_t)o;
        a.val[o] = v;
        ++o;
    };
    for (int x = 0; x < D1; ++x) put(cj >= 0, cj + x, cj + x, 2.0 * E.W[x]);
    for (int x = 0; x < D1; ++x)
        for (int y = 0; y < D1; ++y) {
            const double v = -2.0 * E.GtW[x][y];
            const bool on = cj >= 0 && ci >= 0 && E.G[y][x] != 0.0;
            put(on, ci + x, cj + y, v);
            put(on, cj + y, ci + x, v);
        }
    for (int x = 0; x < D1; ++x)
        for (int y = 0; y < D1; ++y) put(ci >= 0, ci + x, ci + y, 2.0 * E.GtWG[x][y]);
}
// q records of the measurements that touch the pinned pose: d * D1 slots each (replica k: D1 entries)
__global__ __launch_bounds__(256) void k_ga_pin(GaArgs a) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_pin) return;
    const GaProb P = a.probs[a.pin_prob[t]];
    const int d = a.d, D1 = d + 1;
    const int64_t e = a.pin_edge[t];
    const int64_t i = a.rel_base[e], j = a.rel_to[e];
    GaEdge E;
    ga_edge_blocks(a, e, E);
    int64_t o = (int64_t)P.recq_pin + (t - P.pin_off) * d * D1;
    auto putq = [&](bool on, int64_t r, double v) {
        a.key[o] = on ? (((unsigned long long)(P.xoff + r)) << 32 | (unsigned long long)(uint32_t)kQCol) : a.pad_key;
        a.idx[o] = (uint32_t)o;
        a.val[o] = v;
        ++o;
    };
    for (int k = 0; k < d; ++k) {
        const int64_t ci = ga_pcol(P, d, i, k, 0), cj = ga_pcol(P, d, j, k, 0);
        double ui[4] = {0, 0, 0, 0}, uj[4] = {0, 0, 0, 0};  // the pinned pose is [I | 0]: its row k is e_k
        ui[k] = 1.0; uj[k] = 1.0;
        for (int x = 0; x < D1; ++x) {
            if (cj >= 0 && ci < 0) {  // u_i fixed
                double gu = 0;
                for (int l = 0; l < D1; ++l) gu += E.G[x][l] * ui[l];
                putq(true, cj + x, -2.0 * E.W[x] * gu);
            } else if (ci >= 0 && cj < 0) {  // u_j fixed
                double s_ = 0;
                for (int c = 0; c < D1; ++c) s_ += E.GtW[x][c] * uj[c];
                putq(true, ci + x, -2.0 * s_);
            } else {
                putq(false, 0, 0.0);
            }
        }
    }
}
// ranges: P records (SOCP: the distance variable's diagonal; QCQP: 9 couplings of replica 0), q record (SOCP), the cone's
// rows of A and b
__global__ __launch_bounds__(256) void k_ga_rng(GaArgs a) {
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rng) return;
    const GaProb P = a.probs[a.rng_prob[r]];
    const int d = a.d, D1 = d + 1;
    const int64_t lr = r - P.rng_off;
    const double w = a.rng_prec[r], dist = a.rng_dist[r];
    const int64_t va = a.rng_a[r], vb = a.rng_b[r];
    const int64_t rng_base = (int64_t)d * P.n_rep, rq_base = (int64_t)(P.Np - 1) * D1 + P.Nl;
    const int64_t row0 = (int64_t)P.roff + lr * D1;
    if (a.relaxation == 0) {
        const int64_t c = rng_base + lr;
        const int64_t o = (int64_t)P.rec_rng + lr, oq = (int64_t)P.recq_rng + lr;
        a.key[o] = ((unsigned long long)(P.xoff + c)) << 32 | (unsigned long long)(uint32_t)(P.xoff + c);
        a.idx[o] = (uint32_t)o; a.val[o] = 2.0 * w;
        a.key[oq] = ((unsigned long long)(P.xoff + c)) << 32 | (unsigned long long)(uint32_t)kQCol;
        a.idx[oq] = (uint32_t)oq; a.val[oq] = -2.0 * w * dist;
        // (d_ij, t_a - t_b) in SOC: A = -[e_d ; e_ta - e_tb], b = 0
        int k0 = a.A_ptr[row0];
        a.A_col[k0] = (int32_t)(P.xoff + c); a.A_val[k0] = -1.0;
        a.b[row0] = 0.0;
        for (int k = 0; k < d; ++k) {
            int64_t ca = ga_tcol(P, d, va, k), cb = ga_tcol(P, d, vb, k);
            double xa = -1.0, xb = 1.0;
            if (ca >= 0 && cb >= 0 && cb < ca) { const int64_t t = ca; ca = cb; cb = t; const double u = xa; xa = xb; xb = u; }
            int kk = a.A_ptr[row0 + 1 + k];
            if (ca >= 0) { a.A_col[kk] = (int32_t)(P.xoff + ca); a.A_val[kk] = xa; ++kk; }
            if (cb >= 0) { a.A_col[kk] = (int32_t)(P.xoff + cb); a.A_val[kk] = xb; }
            a.b[row0 + 1 + k] = 0.0;
        }
    } else {
        // w || t_a - t_b - dist r ||^2: replica 0's couplings; (1, r_ij) in SOC: A = -[0 ; I], b = (1, 0 .. 0)
        int64_t o = (int64_t)P.rec_rng + lr * 9;
        const int64_t cs[3] = {ga_tcol(P, d, va, 0), ga_tcol(P, d, vb, 0), rq_base + lr};
        const double cf[3] = {1.0, -1.0, -dist};
        for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) {
                const bool on = cs[x] >= 0 && cs[y] >= 0;
                a.key[o] = on ? (((unsigned long long)(P.xoff + cs[x])) << 32 | (unsigned long long)(uint32_t)(P.xoff + cs[y])) : a.pad_key;
                a.idx[o] = (uint32_t)o; a.val[o] = 2.0 * w * cf[x] * cf[y];
                ++o;
            }
        a.b[row0] = 1.0;
        for (int k = 0; k < d; ++k) {
            const int kk = a.A_ptr[row0 + 1 + k];
            a.A_col[kk] = (int32_t)(P.xoff + (int64_t)k * P.n_rep + rq_base + lr); a.A_val[kk] = -1.0;
            a.b[row0 + 1 + k] = 0.0;
        }
    }
}
// landmark priors: one P record (replica 0), d q records
__global__ __launch_bounds__(256) void k_ga_pri(GaArgs a) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_pri) return;
    const GaProb P = a.probs[a.pri_prob[e]];
    const int d = a.d, D1 = d + 1;
    const int64_t le = e - P.pri_off, l = a.pri_lm[e];
    const double w = a.pri_prec[e];
    const int64_t lm_base = (int64_t)(P.Np - 1) * D1;
    const int64_t o = (int64_t)P.rec_pri + le;
    const int64_t c0 = lm_base + l;
    a.key[o] = ((unsigned long long)(P.xoff + c0)) << 32 | (unsigned long long)(uint32_t)(P.xoff + c0);
    a.idx[o] = (uint32_t)o; a.val[o] = 2.0 * w;
    for (int k = 0; k < d; ++k) {
        const int64_t oq = (int64_t)P.recq_pri + le * d + k;
        const int64_t c = (int64_t)k * P.n_rep + lm_base + l;
        a.key[oq] = ((unsigned long long)(P.xoff + c)) << 32 | (unsigned long long)(uint32_t)kQCol;
        a.idx[oq] = (uint32_t)oq; a.val[oq] = -2.0 * w * a.pri_t[e * d + k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The estimate in the reference's own shapes, straight from the device-resident solution (read_estimates_host,
// score_assemble.hpp, is the specification): one thread per pose / landmark / range.  Replaces
// VariableCollection.get_variable_values, /root/reference/score/utils/gurobi_utils.py:114-136.
// ---------------------------------------------------------------------------------------------------------------------
struct EstArgs {
    int32_t d, relaxation, count, qcqp_dirs;
    const EstProb* probs;
    const int32_t* pose_off; const int32_t* lm_off; const int32_t* rng_off;   // count + 1 each
    int64_t n_pose, n_lm, n_rng;
    const double* x; const double* D;     // the equilibrated solution and the column scales: x = xhat * D
    const int32_t* rng_a; const int32_t* rng_b; const double* rng_dist;
    double* poses; double* relaxed; double* lms; double* rng; int32_t* degenerate;
};
__global__ __launch_bounds__(256) void k_read_estimates(EstArgs a) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int d = a.d, D1 = d + 1;
    auto xv = [&](const EstProb& P, int64_t local) { const int64_t c = P.xoff + local; return a.x[c] * a.D[c]; };
    if (i < a.n_pose) {
        const EstProb P = a.probs[a.count > 1 ? tab_find(a.pose_off, a.count, i) : 0];
        const int64_t lp = i - P.pose_off;
        double blk[3][4], m[9], r[9];
        for (int k = 0; k < d; ++k)
            for (int j = 0; j < D1; ++j) blk[k][j] = lp == 0 ? (j == k ? 1.0 : 0.0) : xv(P, (int64_t)k * P.n_rep + (lp - 1) * D1 + j);
        for (int k = 0; k < d; ++k)
            for (int j = 0; j < d; ++j) m[k * d + j] = blk[k][j];
        int32_t bad = 0;
        if (d == 2) round_so2(m, r, &bad); else round_so3(m, r, &bad);
        a.degenerate[i] = bad;
        for (int k = 0; k < d; ++k)
            for (int j = 0; j < D1; ++j) a.relaxed[i * d * D1 + k * D1 + j] = blk[k][j];
        double* T = a.poses + i * D1 * D1;
        for (int k = 0; k < d; ++k) {
            for (int j = 0; j < d; ++j) T[k * D1 + j] = r[k * d + j];
            T[k * D1 + d] = blk[k][d];
        }
        for (int j = 0; j < d; ++j) T[d * D1 + j] = 0.0;
        T[d * D1 + d] = 1.0;
    }
    if (i < a.n_lm) {
        const EstProb P = a.probs[a.count > 1 ? tab_find(a.lm_off, a.count, i) : 0];
        const int64_t l = i - P.lm_off, lm0 = (int64_t)(P.Np - 1) * D1;
        for (int k = 0; k < d; ++k) a.lms[i * d + k] = xv(P, (int64_t)k * P.n_rep + lm0 + l);
    }
    if (i < a.n_rng) {
        const EstProb P = a.probs[a.count > 1 ? tab_find(a.rng_off, a.count, i) : 0];
        const int64_t r_ = i - P.rng_off, lm0 = (int64_t)(P.Np - 1) * D1;
        if (a.relaxation != 0) {
            for (int k = 0; k < d; ++k) a.rng[i * d + k] = xv(P, (int64_t)k * P.n_rep + lm0 + P.Nl + r_);
        } else if (!a.qcqp_dirs) {
            a.rng[i] = xv(P, (int64_t)d * P.n_rep + r_);
        } else {
            auto tvar = [&](int64_t v, int k) {
                if (v < P.Np) return v == 0 ? 0.0 : xv(P, (int64_t)k * P.n_rep + (v - 1) * D1 + d);
                return xv(P, (int64_t)k * P.n_rep + lm0 + (v - P.Np));
            };
            double dl[3], nn = 0.0;
            for (int k = 0; k < d; ++k) { dl[k] = tvar(a.rng_a[i], k) - tvar(a.rng_b[i], k); nn += dl[k] * dl[k]; }
            const double den = fmax(sqrt(nn), a.rng_dist[i]);
            const bool idle = !(a.rng_dist[i] > 0.0);  // (measured distance 0: r = 0 on every path, see read_estimates_host)
            for (int k = 0; k < d; ++k) a.rng[i * d + k] = (den > 0.0 && !idle) ? dl[k] / den : 0.0;
        }
    }
}

}  // namespace score
