// score_join.hpp -- the second level of the chain preconditioner for chains of more than kSegMaxNodes nodes (score_host.hpp:
// JoinChain).  A long chain's block-tridiagonal matrix T, with its nodes ordered [segments | separators], is
//
//     T = [ T_S  C ]     T_S = diag(T_1 .. T_k)   the segments: ordinary chains, factored by k_factor, applied by k_prec_pre
//         [ C'   D ]     D   = diag(D_1 .. D_k-1) the separators' diagonal blocks; C couples separator j to the last node of
//                                                 segment j (block A_j) and to the first node of segment j + 1 (block B_j)
//
// and T z = r is solved exactly by  y = T_S^-1 r_S  (the chain kernel),  Sigma z_b = r_b - C' y  with the Schur complement
// Sigma = D - C' T_S^-1 C  (block tridiagonal, k - 1 nodes),  z_S = y - W z_b  with the spikes W = T_S^-1 C.  The spikes are
// 2 BS vectors over the unknowns (left separator's BS columns, right separator's BS columns; every segment of every long chain
// at once), made after each factorisation by 2 BS applications of the chain kernel to the coupling columns (k_join_rhs);
// k_join_schur forms Sigma from the spikes' end blocks and factors it (block Thomas); k_join_solve + k_join_apply run after every
// chain-kernel launch: separators' right-hand sides and the small solve (one wavefront per long chain), then the correction of
// every segment and the r'z partial sum of its work item restated with the corrected z.
//
// The separators are Jacobi columns of the chain kernel with inverse diagonal zero: the PCG step's vector updates (r, xt, kx)
// happen there, z and their share of r'z come from here.  With this the preconditioner is the exact chain solve the streaming
// kernel k_prec computes for long chains -- same PCG iteration counts -- at the cost of the LDS-resident kernel plus one short
// launch.  Structure: /root/reference/score/utils/gurobi_utils.py:380-404, :504-526 (odometry couples pose i to i + 1 only).
#pragma once

#include "score_kernels.hpp"

namespace score {

constexpr int kJoinThreads = 256;
constexpr int kJoinMaxSeps = 128;  // separators of one long chain the join kernel keeps in LDS (chains of up to 129 segments: 132 k nodes)
static_assert(kJoinMaxSeps == kJoinMaxSepsHost, "build_system keeps longer chains whole (score_host.hpp)");

struct JoinArgs {
    const JoinChain* jc;
    const JoinItem* items;
    const ChainDesc* chains;
    const int32_t* node_col;
    const int32_t* sep_col;
    const int32_t* sep_nodes;  // pseudo-node tables (2 per separator): [b, first of the segment after]; sep_prev: [last of the segment before, b]
    const int32_t* sep_prev;
    const int32_t* done;
    int use_owner;            // the matrix blocks of a join chain are its owner's (K of a replicated problem)
    // spikes and Schur data
    double* W;                // 2 BS vectors of n_tot
    int64_t n_tot;
    double* data;             // per separator: A, B, Pinv, Lo, Up (5 BS^2)
    // matrix values through the pseudo-node position tables (2 per separator: [b | prev = last of the segment before],
    // [first of the segment after | prev = b])
    const double* val;
    const int32_t* pos_diag;
    const int32_t* pos_sub;
    // k_join_rhs
    double* rhs;
    int column;
    // k_prec_join
    const double* r;
    double* z;
    double* p;                // INIT: receives z
    double* rz_out;
    double* zb;               // the separators' solution (k_join_solve -> k_join_apply): BS per separator
    // several vectors in one launch (grid.y; score_link.hpp's refresh): r, z, p advance by y * vec_stride, zb by y * zb_stride,
    // rz_out by y * rz_stride (the chain kernel's grid)
    int n_vec;
    long long vec_stride, zb_stride, rz_stride;
    // k_join_dinv
    const int32_t* sep_diag;
    double* dinv;
    int n_sep_entries;
};

template <int BS>
__device__ __forceinline__ int join_col(const ChainDesc& ch, const int32_t* __restrict__ node_col, int node) {
    return ch.col_stride ? ch.col0 + node * ch.col_stride : node_col[ch.node_begin + node];
}

// dinv of the separators' Jacobi entries := 0 (after every factorisation: k_factor's Jacobi items have just rewritten them)
__global__ __launch_bounds__(kJoinThreads) void k_join_dinv(JoinArgs a) {
    const int i = blockIdx.x * kJoinThreads + threadIdx.x;
    if (i < a.n_sep_entries && a.sep_diag[i] >= 0) a.dinv[a.sep_diag[i]] = 0.0;
}

// Right-hand side of spike `column` (0 .. BS-1: the left separator's columns, BS .. 2BS-1: the right one's): zero except in
// the first / last node's rows of every segment.  One thread per join item and block row.
template <int BS>
__global__ __launch_bounds__(kJoinThreads) void k_join_rhs(JoinArgs a, int n_items) {
    const int i = (blockIdx.x * kJoinThreads + threadIdx.x) / BS, row = (blockIdx.x * kJoinThreads + threadIdx.x) % BS;
    if (i >= n_items) return;
    const JoinItem it = a.items[i];
    const JoinChain jc = a.jc[it.jc];
    const JoinChain own = a.use_owner ? a.jc[jc.owner] : jc;
    const ChainDesc ch = a.chains[it.chain];
    const int first = join_col<BS>(ch, a.node_col, 0), last = join_col<BS>(ch, a.node_col, ch.N - 1);
    const int c = a.column;
    double v_first = 0.0, v_last = 0.0;
    if (c < BS && it.seg >= 1) {  // K[first + row, b_left + c]: pseudo-node 2 sp + 1 of the left separator
        const int sp = own.sep_begin + it.seg - 1;
        const int pos = a.pos_sub[(size_t)(2 * sp + 1) * BS * BS + row * BS + c];
        v_first = pos >= 0 ? a.val[pos] : 0.0;
    }
    if (c >= BS && it.seg + 1 < jc.n_seg) {  // K[last + row, b_right + (c - BS)] = K[b_right + (c - BS), last + row]: pseudo-node 2 sp
        const int sp = own.sep_begin + it.seg;
        const int pos = a.pos_sub[(size_t)(2 * sp) * BS * BS + (c - BS) * BS + row];
        v_last = pos >= 0 ? a.val[pos] : 0.0;
    }
    if (first == last) a.rhs[first + row] = v_first + v_last;
    else { a.rhs[first + row] = v_first; a.rhs[last + row] = v_last; }
}

template <int BS>
__device__ __forceinline__ void small_inverse(const double (&M)[BS * BS], double (&I)[BS * BS]) {  // Gauss-Jordan, SPD input
    double A[BS][2 * BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) { A[i][j] = M[i * BS + j]; A[i][BS + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < BS; ++k) {
        const double piv = 1.0 / A[k][k];
#pragma unroll
        for (int j = 0; j < 2 * BS; ++j) A[k][j] *= piv;
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            if (i == k) continue;
            const double f = A[i][k];
#pragma unroll
            for (int j = 0; j < 2 * BS; ++j) A[i][j] -= f * A[k][j];
        }
    }
#pragma unroll
    for (int i = 0; i < BS; ++i)
#pragma unroll
        for (int j = 0; j < BS; ++j) I[i * BS + j] = A[i][BS + j];
}

// Sigma and its block-Thomas factors, one thread per join chain (owners only when use_owner)
template <int BS>
__global__ __launch_bounds__(64) void k_join_schur(JoinArgs a, int n_jc) {
    constexpr int B2 = BS * BS;
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jc) return;
    const JoinChain jc = a.jc[j];
    if (a.use_owner && jc.owner != j) return;
    auto val = [&](int pos) { return pos >= 0 ? a.val[pos] : 0.0; };
    auto Wv = [&](int vec, int col) { return a.W[(size_t)vec * a.n_tot + col]; };
    double Pinv_prev[B2], Up_prev[B2];
    for (int s = 0; s + 1 < jc.n_seg; ++s) {
        const int sp = jc.sep_begin + s;
        double* out = a.data + (size_t)sp * 5 * B2;
        const ChainDesc cl = a.chains[jc.first_chain + s], cr = a.chains[jc.first_chain + s + 1];
        const int last = join_col<BS>(cl, a.node_col, cl.N - 1), first = join_col<BS>(cr, a.node_col, 0);
        const int last_r = join_col<BS>(cr, a.node_col, cr.N - 1);
        double A[B2], B[B2], S[B2], Up[B2];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                S[r * BS + c] = val(a.pos_diag[(size_t)(2 * sp) * B2 + r * BS + c]);
                A[r * BS + c] = val(a.pos_sub[(size_t)(2 * sp) * B2 + r * BS + c]);      // K[b + r, last + c]
                B[r * BS + c] = val(a.pos_sub[(size_t)(2 * sp + 1) * B2 + c * BS + r]);  // K[first + c, b + r]
            }
        // Sigma_ss = D - A Mr - B Ml,  Mr[x][c] = W_{BS+c}[last + x] (right spike of the left segment at its last node),
        //                               Ml[x][c] = W_c[first + x]      (left spike of the right segment at its first node)
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = S[r * BS + c];
#pragma unroll
                for (int x = 0; x < BS; ++x) acc -= A[r * BS + x] * Wv(BS + c, last + x) + B[r * BS + x] * Wv(c, first + x);
                S[r * BS + c] = acc;
            }
        // Sigma_{s, s+1} = -B Mx,  Mx[x][c] = W_{BS+c}[first + x] (right spike of the right segment at its FIRST node)
        const bool has_next = s + 2 < jc.n_seg;
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                double acc = 0.0;
                if (has_next) {
#pragma unroll
                    for (int x = 0; x < BS; ++x) acc -= B[r * BS + x] * Wv(BS + c, first + x);
                }
                Up[r * BS + c] = acc;
            }
        (void)last_r;
        // block Thomas: pivot = Sigma_ss - Lo Up_prev with Lo = Up_prev' Pinv_prev
        double Lo[B2];
#pragma unroll
        for (int e = 0; e < B2; ++e) Lo[e] = 0.0;
        if (s > 0) {
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int x = 0; x < BS; ++x) acc += Up_prev[x * BS + r] * Pinv_prev[x * BS + c];
                    Lo[r * BS + c] = acc;
                }
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int c = 0; c < BS; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int x = 0; x < BS; ++x) acc += Lo[r * BS + x] * Up_prev[x * BS + c];
                    S[r * BS + c] -= acc;
                }
        }
        double Pinv[B2];
        small_inverse<BS>(S, Pinv);
#pragma unroll
        for (int e = 0; e < B2; ++e) {
            out[e] = A[e]; out[B2 + e] = B[e]; out[2 * B2 + e] = Pinv[e]; out[3 * B2 + e] = Lo[e]; out[4 * B2 + e] = Up[e];
            Pinv_prev[e] = Pinv[e]; Up_prev[e] = Up[e];
        }
    }
}

// After the chain kernel (z = y on the segments, z = 0 on the separators).  Two launches, because the first reads y at the
// ends of every segment and the second overwrites it: k_join_solve -- one wavefront per long chain: the separators' right-hand
// sides r_b - A y[last of the left segment] - B y[first of the right segment], the block-Thomas solve, z_b into `zb`;
// k_join_apply -- one workgroup per segment: z -= W_left z_b[seg - 1] + W_right z_b[seg], the separator to its right, and the
// r'z partial sum of the segment's work item restated with the corrected z.
__device__ __forceinline__ void join_select_vector(JoinArgs& a) {
    if (a.n_vec > 1) {
        const long long off = (long long)blockIdx.y * a.vec_stride;
        a.r += off; a.z += off; a.zb += (long long)blockIdx.y * a.zb_stride;
        if (a.p) a.p += off;
        if (a.rz_out) a.rz_out += (long long)blockIdx.y * a.rz_stride;
    }
}
template <int BS>
__global__ __launch_bounds__(64) void k_join_solve(JoinArgs a) {
    join_select_vector(a);
    constexpr int B2 = BS * BS;
    __shared__ double g[kJoinMaxSeps * BS];
    __shared__ double F[kJoinMaxSeps * 3 * B2];  // Pinv, Lo, Up of every separator: the sequential solve reads LDS only
    const JoinChain jc = a.jc[blockIdx.x];
    if (a.done[jc.prob]) return;  // frozen problem (or a PCG whose gate has fired): the chain kernel wrote nothing either
    const int sep0 = a.use_owner ? a.jc[jc.owner].sep_begin : jc.sep_begin;  // the matrix blocks
    const int t = threadIdx.x;
    const int n_sep = jc.n_seg - 1;
    for (int e = t; e < n_sep * 3 * B2; e += 64) {
        const int s = e / (3 * B2);
        F[e] = a.data[(size_t)(sep0 + s) * 5 * B2 + 2 * B2 + (e - s * 3 * B2)];
    }
    for (int s = t; s < n_sep; s += 64) {
        const double* __restrict__ D = a.data + (size_t)(sep0 + s) * 5 * B2;
        // (the pseudo-node tables of the position look-up hold the columns: [b | last of the left segment], [first of the right | b])
        const int sp = jc.sep_begin + s;
        const int b = a.sep_nodes[2 * sp], last = a.sep_prev[2 * sp], first = a.sep_nodes[2 * sp + 1];
        double yl[BS], yr[BS];
#pragma unroll
        for (int c = 0; c < BS; ++c) { yl[c] = a.z[last + c]; yr[c] = a.z[first + c]; }
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            double acc = a.r[b + r];
#pragma unroll
            for (int c = 0; c < BS; ++c) acc -= D[r * BS + c] * yl[c] + D[B2 + r * BS + c] * yr[c];
            g[s * BS + r] = acc;
        }
    }
    __syncthreads();
    if (t == 0) {
        for (int s = 1; s < n_sep; ++s) {
            const double* Lo = F + s * 3 * B2 + B2;
            double v[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                double acc = g[s * BS + r];
#pragma unroll
                for (int c = 0; c < BS; ++c) acc -= Lo[r * BS + c] * g[(s - 1) * BS + c];
                v[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < BS; ++r) g[s * BS + r] = v[r];
        }
        for (int s = n_sep - 1; s >= 0; --s) {
            const double* Pinv = F + s * 3 * B2;
            const double* Up = Pinv + 2 * B2;
            double v[BS], x[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                double acc = g[s * BS + r];
                if (s + 1 < n_sep) {
#pragma unroll
                    for (int c = 0; c < BS; ++c) acc -= Up[r * BS + c] * g[(s + 1) * BS + c];
                }
                v[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int c = 0; c < BS; ++c) acc += Pinv[r * BS + c] * v[c];
                x[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < BS; ++r) g[s * BS + r] = x[r];
        }
    }
    __syncthreads();
    for (int e = t; e < n_sep * BS; e += 64) a.zb[(size_t)jc.sep_begin * BS + e] = g[e];
}

template <int BS, int MODE>
__global__ __launch_bounds__(kJoinThreads) void k_join_apply(JoinArgs a) {
    join_select_vector(a);
    __shared__ double red[16];
    const JoinItem it = a.items[blockIdx.x];
    const JoinChain jc = a.jc[it.jc];
    if (a.done[jc.prob]) return;
    const int t = threadIdx.x;
    const int n_sep = jc.n_seg - 1;
    const double* __restrict__ zb = a.zb + (size_t)jc.sep_begin * BS;
    double zl[BS], zr[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) {
        zl[c] = it.seg >= 1 ? zb[(it.seg - 1) * BS + c] : 0.0;
        zr[c] = it.seg < n_sep ? zb[it.seg * BS + c] : 0.0;
    }
    const ChainDesc ch = a.chains[it.chain];
    const int NB = ch.N * BS;
    double local = 0.0;
    // four entries per lane and trip, every load of the trip requested before the first use (clamped addresses, predicated
    // stores): a segment is one workgroup's work, and 200 workgroups on 256 CUs hide no latency by themselves
    constexpr int U = 4;
    for (int base = t; base < NB; base += kJoinThreads * U) {
        int col[U];
        double zv[U], rv[U], wl_[U][BS], wr_[U][BS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(base + u * kJoinThreads, NB - 1);
            const int node = e / BS;
            col[u] = join_col<BS>(ch, a.node_col, node) + (e - node * BS);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            zv[u] = a.z[col[u]];
            rv[u] = a.r[col[u]];
#pragma unroll
            for (int c = 0; c < BS; ++c) { wl_[u][c] = a.W[(size_t)c * a.n_tot + col[u]]; wr_[u][c] = a.W[(size_t)(BS + c) * a.n_tot + col[u]]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * kJoinThreads < NB) {
                double zz = zv[u];
#pragma unroll
                for (int c = 0; c < BS; ++c) zz -= wl_[u][c] * zl[c] + wr_[u][c] * zr[c];
                a.z[col[u]] = zz;
                if (MODE == PREC_INIT) a.p[col[u]] = zz;
                local += rv[u] * zz;
            }
        }
    }
    if (it.seg < n_sep && t < BS) {  // the separator to the right of this segment
        const int b = a.sep_col[jc.sep_begin + it.seg];
        const double zz = zb[it.seg * BS + t];
        a.z[b + t] = zz;
        if (MODE == PREC_INIT) a.p[b + t] = zz;
        local += a.r[b + t] * zz;
    }
    const double tot = block_sum_n<kJoinThreads / 64>(local, red);
    if (t == 0) a.rz_out[it.work] = tot;
}

}  // namespace score
