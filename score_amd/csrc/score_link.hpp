// score_link.hpp -- loop closures inside the Newton preconditioner (round 6).
//
// The chain preconditioner T (score_kernels.hpp: k_prec_pre; score_join.hpp for chains beyond 1023 nodes) is the exact inverse
// of the per-robot block-tridiagonal part of the Newton matrix H: what ODOMETRY couples (gurobi_utils.py:380-404).  A loop
// closure (:407-430) is the same relative-pose term (:504-526) between two poses that are NOT neighbours in a chain -- as stiff
// as an odometry edge and outside T.  Measured on the oracle's Newton systems (scratch prototype): a 2 x 397-pose
// graph with two loop closures takes 377 PCG iterations with T alone and 97 with the loop-closure blocks inside the preconditioner,
// 2 x 2570 poses with three 495 -> 106.
//
// With U the columns of the unknowns of the linked nodes (a selection: n_u <= kLinkMaxU per problem) and G = the entries of H
// between linked nodes (the blocks T does not hold), the preconditioner becomes the exact inverse of T + U G U':
//
//     (T + U G U')^-1 r = y - Z t,    y = T^-1 r,   Z = T^-1 U,   t = Q y[U],   Q = (I + G Z[U,:])^-1 G          (Woodbury)
//
// Z e_u is nonzero on u's own chain only, so one application of the chain kernel to a sum of unit vectors -- one per chain --
// yields one column for every chain at once: `rounds` = the most unknowns any chain carries (3 per linked pose and matrix row in
// 2-D).  After every factorisation of H's chains (HipBackend::launch_factor -> link_refresh): the rounds (k_link_rhs + the chain
// kernel, second level included), then k_link_cap: G from H's values, S = I + G Z[U,:], Q = S^-1 G by Gauss-Jordan with partial
// pivoting in LDS (one workgroup per problem).  After every application of the chain kernel to the Newton set (launch_prec ->
// link_apply): k_link_solve (t = Q y[U], one workgroup per problem) and k_link_apply (one workgroup per affected chain:
// z -= Z t, and the chain's r'z partial sum restated with the corrected z) -- two launches because the first reads what the
// second overwrites, as in score_join.hpp.  Graphs without loop closures never come here (n_items == 0: no launch, no buffer).
//
// Two sets of these buffers (HipBackend::link_*[2]): one for the ADMM loop's K (refreshed with every penalty change), one
// for the Newton matrix H (refreshed with every k_factor).  SCORE_NO_LINKS=1 switches the correction off for both.
#pragma once

#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "score_host.hpp"
#if defined(__HIPCC__)
#include "score_join.hpp"
#endif

namespace score {

constexpr int kLinkMaxU = 96;       // unknowns of one independent group (make_link_plan): the system k_link_cap solves in LDS
constexpr int kLinkMaxRounds = 48;  // unknowns of one (whole) chain: right-hand sides of the chain kernel per refresh
constexpr int kLinkMaxNodes = 256;  // linked nodes of one problem
constexpr int kLinkThreads = 256;

struct LinkItem {   // one affected chain (a segment of a long chain: every segment of it) of a problem
    int32_t prob, chain, work, sep_col;  // work: its slot of the r'z partial sums; sep_col: separator to its right (-1: none)
    int32_t u[kLinkMaxRounds];           // per round: the unknown (index into the handle's unknown list) whose column this chain carries, -1 none
};
struct LinkProb {  // one independent group of a problem's unknowns (make_link_plan): a system of its own
    int32_t prob, u_begin, n_u, q_off;   // q_off: first entry of the group's n_u x n_u tables (mask / positions / Q')
    int32_t item_begin, item_count;      // the group's affected chains in LinkPlan::items
};

// ---- host: which chain nodes do relative-pose terms couple outside the chains? (pairs of global first columns) ----
inline void find_link_pairs_graphs(const HostSystem& h, const score_graph* graphs, std::vector<int32_t>& out) {
    for (int p = 0; p < h.count; ++p) {
        const score_graph& g = graphs[p];
        const int d = g.dim, D1 = d + 1;
        std::vector<int64_t> first((size_t)g.n_chains + 1, 0);
        for (int c = 0; c < g.n_chains; ++c) first[(size_t)c + 1] = first[(size_t)c] + g.chain_len[c];
        const int64_t Np = first[(size_t)g.n_chains];
        const int64_t n_rep = (Np - 1) * D1 + g.n_landmarks + (g.relaxation == 0 ? 0 : g.n_rng);  // (graph_skeleton's layout)
        auto chain_of = [&](int64_t v) { return (int)(std::upper_bound(first.begin(), first.end(), v) - first.begin()) - 1; };
        for (int64_t e = 0; e < g.n_rel; ++e) {
            const int64_t i = g.rel_base[e], j = g.rel_to[e];
            if (i <= 0 || j <= 0 || i >= Np || j >= Np || i == j) continue;  // (the pinned pose is no unknown)
            if ((i - j == 1 || j - i == 1) && chain_of(i) == chain_of(j)) continue;  // odometry: inside the chain
            for (int k = 0; k < d; ++k) {
                out.push_back((int32_t)(h.xoff[p] + k * n_rep + (std::min(i, j) - 1) * D1));
                out.push_back((int32_t)(h.xoff[p] + k * n_rep + (std::max(i, j) - 1) * D1));
            }
        }
    }
}
// The same from P's pattern (handles made from arrays): the row of a node's FIRST unknown -- a rotation entry, which only
// relative-pose terms touch (range terms couple translations, also after the head-form rewrite) -- holds a column of a chain
// node that is neither the node itself nor its neighbour in the chain.
inline void find_link_pairs_P(const HostSystem& h, const score_problem* probs, std::vector<int32_t>& out) {
    for (int p = 0; p < h.count; ++p) {
        const score_problem& pr = probs[p];
        if (!pr.P_rowptr || !pr.P_col || pr.n_chains <= 0 || !pr.node_first_col) continue;
        const int bs = pr.block_size, nn = pr.chain_ptr[pr.n_chains];
        std::vector<int32_t> node_of((size_t)pr.n, -1), chain_of((size_t)nn, 0);
        for (int c = 0; c < pr.n_chains; ++c)
            for (int j = pr.chain_ptr[c]; j < pr.chain_ptr[c + 1]; ++j) {
                chain_of[(size_t)j] = c;
                for (int a = 0; a < bs; ++a) node_of[(size_t)pr.node_first_col[j] + a] = j;
            }
        int32_t last_a = -1, last_b = -1;
        for (int j = 0; j < nn; ++j) {
            const int32_t r = pr.node_first_col[j];
            for (int32_t k = pr.P_rowptr[r]; k < pr.P_rowptr[r + 1]; ++k) {
                const int32_t j2 = node_of[(size_t)pr.P_col[k]];
                if (j2 <= j) continue;
                if (j2 == j + 1 && chain_of[(size_t)j2] == chain_of[(size_t)j]) continue;
                const int32_t a = (int32_t)(h.xoff[p] + r), b = (int32_t)(h.xoff[p] + pr.node_first_col[j2]);
                if (a == last_a && b == last_b) continue;  // (the other entries of the same block)
                out.push_back(a); out.push_back(b);
                last_a = a; last_b = b;
            }
        }
    }
}

// What the backend uploads: unknowns, rounds, items, masks -- built from the pairs and the Newton set's chains.
struct LinkPlan {
    std::vector<LinkProb> probs;
    std::vector<LinkItem> items;
    std::vector<int32_t> ucol, uround, usuper;  // per unknown: global column, its round, its (whole) chain
    std::vector<uint8_t> mask;                  // per problem n_u x n_u: 1 = (a, b) lie in linked nodes (an entry of G)
    std::vector<int32_t> pair_cols;             // the node pairs inside the preconditioner (global first columns, two per pair)
    int rounds = 0, max_items = 0;              // max_items: the most chains any group touches
    int pairs_total = 0, pairs_used = 0;
    bool empty() const { return items.empty(); }
};
inline void make_link_plan(const HostSystem& h, const std::vector<int32_t>& pairs_in, LinkPlan& L) {
    L = LinkPlan();
    if (pairs_in.empty()) return;
    // one order whatever found the pairs (the measurement list or P's pattern): sorted by columns, duplicates dropped -- the
    // unknowns' order decides the order of every sum below, and both paths must give the same bits
    std::vector<int32_t> pairs;
    {
        std::vector<std::pair<int32_t, int32_t>> pp;
        for (size_t k = 0; k + 1 < pairs_in.size(); k += 2) pp.push_back({std::min(pairs_in[k], pairs_in[k + 1]), std::max(pairs_in[k], pairs_in[k + 1])});
        std::sort(pp.begin(), pp.end());
        pp.erase(std::unique(pp.begin(), pp.end()), pp.end());
        for (const auto& q : pp) { pairs.push_back(q.first); pairs.push_back(q.second); }
    }
    L.pairs_total = (int)(pairs.size() / 2);
    const int bs = h.bs;
    const std::vector<ChainDesc>& chains = h.chainsH;
    // column -> (chain, node); whole-chain id of every chain (a long chain's segments share their join chain's)
    std::unordered_map<int32_t, std::pair<int32_t, int32_t>> where;
    where.reserve(pairs.size());
    {
        std::unordered_map<int32_t, int> wanted;
        for (int32_t c : pairs) wanted[c] = 1;
        for (size_t ci = 0; ci < chains.size(); ++ci)
            for (int i = 0; i < chains[ci].N; ++i) {
                const int32_t col = h.node_col[(size_t)chains[ci].node_begin + i];
                if (wanted.count(col)) where[col] = {(int32_t)ci, i};
            }
    }
    std::vector<int32_t> super(chains.size());
    for (size_t ci = 0; ci < chains.size(); ++ci) super[ci] = (int32_t)ci;
    for (const JoinItem& it : h.join_items) super[(size_t)it.chain] = h.join_chains[(size_t)it.jc].first_chain;
    std::vector<int32_t> work_of(chains.size(), -1);
    for (size_t w = 0; w < h.prec_work.size(); ++w)
        if (h.prec_work[w].kind == 0) work_of[(size_t)h.prec_work[w].index] = (int32_t)w;
    // per problem: nodes in order of appearance, under the caps
    struct Node { int32_t col, chain, sup, u0; };
    std::vector<std::vector<Node>> nodes((size_t)h.count);
    std::vector<std::vector<std::pair<int, int>>> used((size_t)h.count);  // node-index pairs per problem
    std::unordered_map<int32_t, int> rounds_of;                            // whole chain -> unknowns so far
    // ALL of a problem's pairs or none: a partial correction buys next to nothing (measured: 4 x 1000 poses with 20 loop closures,
    // 8 of them inside: 135 PCG iterations per Newton step against 151 without any, at 2.5 x the cost per iteration) -- a problem
    // whose pairs exceed the caps keeps the chain preconditioner alone
    std::vector<char> over((size_t)h.count, 0);
    for (size_t k = 0; k + 1 < pairs.size(); k += 2) {
        const auto fa = where.find(pairs[k]), fb = where.find(pairs[k + 1]);
        if (fa == where.end() || fb == where.end()) continue;  // (an endpoint that is a separator of a long chain: left out)
        const int p = chains[(size_t)fa->second.first].prob;
        if (p != chains[(size_t)fb->second.first].prob || over[(size_t)p]) continue;
        auto& N = nodes[(size_t)p];
        auto find = [&](int32_t col) { for (size_t i = 0; i < N.size(); ++i) if (N[i].col == col) return (int)i; return -1; };
        int ia = find(pairs[k]), ib = find(pairs[k + 1]);
        const int32_t sa = super[(size_t)fa->second.first], sb = super[(size_t)fb->second.first];
        const int add_a = ia < 0 ? bs : 0, add_b = ib < 0 ? bs : 0;
        if ((int)N.size() + 2 > kLinkMaxNodes || rounds_of[sa] + add_a + (sa == sb ? add_b : 0) > kLinkMaxRounds ||
            rounds_of[sb] + add_b + (sa == sb ? add_a : 0) > kLinkMaxRounds) {
            over[(size_t)p] = 1;
            L.pairs_used -= (int)used[(size_t)p].size();
            N.clear(); used[(size_t)p].clear();
            continue;
        }
        if (ia < 0) { ia = (int)N.size(); N.push_back(Node{pairs[k], fa->second.first, sa, 0}); rounds_of[sa] += bs; }
        if (ib < 0) { ib = (int)N.size(); N.push_back(Node{pairs[k + 1], fb->second.first, sb, 0}); rounds_of[sb] += bs; }
        used[(size_t)p].push_back({ia, ib});
        ++L.pairs_used;
    }
    std::unordered_map<int32_t, int> next_round;  // whole chain -> next free round
    for (int p = 0; p < h.count; ++p) {
        auto& N = nodes[(size_t)p];
        if (N.empty()) continue;
        // The unknowns fall into independent groups: two nodes belong together when they sit on the same (whole) chain -- Z
        // couples them -- or are a linked pair -- G does.  A loop closure couples row k of one pose with row k of another and the
        // rows' chains are separate, so a 2-D problem splits into at least two groups (3-D: three), more when the loop closures
        // sit on different robots.  Every group is a system of its own for k_link_cap / k_link_solve (a LinkProb): the
        // Gauss-Jordan steps, which are a chain of LDS round trips each, shrink with the largest group, not with the problem.
        std::vector<int> root(N.size());
        for (size_t i = 0; i < N.size(); ++i) root[i] = (int)i;
        auto find_root = [&](int i) { while (root[(size_t)i] != i) i = root[(size_t)i] = root[(size_t)root[(size_t)i]]; return i; };
        for (size_t i = 0; i < N.size(); ++i)
            for (size_t j = i + 1; j < N.size(); ++j)
                if (N[i].sup == N[j].sup) root[(size_t)find_root((int)j)] = find_root((int)i);
        for (const auto& pr : used[(size_t)p]) root[(size_t)find_root(pr.second)] = find_root(pr.first);
        std::vector<int> group_roots;
        for (size_t i = 0; i < N.size(); ++i) {
            const int r = find_root((int)i);
            if (std::find(group_roots.begin(), group_roots.end(), r) == group_roots.end()) group_roots.push_back(r);
        }
        {   // (a group beyond what k_link_cap holds in LDS: the problem keeps the chain preconditioner alone)
            bool too_big = false;
            for (int gr : group_roots) {
                int cnt = 0;
                for (size_t i = 0; i < N.size(); ++i) cnt += find_root((int)i) == gr ? bs : 0;
                too_big = too_big || cnt > kLinkMaxU;
            }
            if (too_big) { L.pairs_used -= (int)used[(size_t)p].size(); continue; }
        }
        for (int gr : group_roots) {
            LinkProb P{};
            P.prob = p; P.u_begin = (int32_t)L.ucol.size(); P.q_off = (int32_t)L.mask.size();
            for (size_t i = 0; i < N.size(); ++i) {
                if (find_root((int)i) != gr) continue;
                Node& nd = N[i];
                nd.u0 = (int32_t)L.ucol.size();
                for (int a = 0; a < bs; ++a) {
                    L.ucol.push_back(nd.col + a);
                    L.uround.push_back(next_round[nd.sup]++);
                    L.usuper.push_back(nd.sup);
                }
            }
            P.n_u = (int32_t)L.ucol.size() - P.u_begin;
            L.mask.resize(L.mask.size() + (size_t)P.n_u * P.n_u, 0);
            for (const auto& pr : used[(size_t)p]) {
                if (find_root(pr.first) != gr) continue;
                for (int a = 0; a < bs; ++a)
                    for (int b = 0; b < bs; ++b) {
                        const int ua = N[(size_t)pr.first].u0 - P.u_begin + a, ub = N[(size_t)pr.second].u0 - P.u_begin + b;
                        L.mask[(size_t)P.q_off + (size_t)ua * P.n_u + ub] = 1;
                        L.mask[(size_t)P.q_off + (size_t)ub * P.n_u + ua] = 1;
                    }
            }
            // the group's affected chains: every segment of every whole chain that carries one of its unknowns
            P.item_begin = (int32_t)L.items.size();
            std::vector<int32_t> sups;
            for (size_t i = 0; i < N.size(); ++i)
                if (find_root((int)i) == gr && std::find(sups.begin(), sups.end(), N[i].sup) == sups.end()) sups.push_back(N[i].sup);
            for (int32_t sc : sups) {
                std::vector<std::pair<int32_t, int32_t>> segs;  // (chain, separator to the right or -1)
                bool joined = false;
                for (const JoinItem& it : h.join_items)
                    if (h.join_chains[(size_t)it.jc].first_chain == sc) {
                        const JoinChain& jc = h.join_chains[(size_t)it.jc];
                        segs.push_back({it.chain, it.seg + 1 < jc.n_seg ? h.join_sep_col[(size_t)jc.sep_begin + it.seg] : -1});
                        joined = true;
                    }
                if (!joined) segs.push_back({sc, -1});
                for (const auto& sg : segs) {
                    LinkItem it{};
                    it.prob = p; it.chain = sg.first; it.work = work_of[(size_t)sg.first]; it.sep_col = sg.second;
                    for (int r = 0; r < kLinkMaxRounds; ++r) it.u[r] = -1;
                    for (size_t u = (size_t)P.u_begin; u < (size_t)P.u_begin + P.n_u; ++u)
                        if (L.usuper[u] == sc) it.u[L.uround[u]] = (int32_t)u;
                    L.items.push_back(it);
                }
            }
            P.item_count = (int32_t)L.items.size() - P.item_begin;
            L.probs.push_back(P);
        }
        for (const auto& pr : used[(size_t)p]) { L.pair_cols.push_back(N[(size_t)pr.first].col); L.pair_cols.push_back(N[(size_t)pr.second].col); }
    }
    for (int32_t r : L.uround) L.rounds = std::max(L.rounds, r + 1);
    for (const LinkProb& P : L.probs) L.max_items = std::max(L.max_items, (int)P.item_count);
}

// The column whose row of a row-replicated K holds the entries of `col` (K stores replica 0's rows: a column of replica k maps to
// its replica-0 sibling; tail columns and general problems map to themselves).  `shift` receives the distance.
inline int32_t link_owner_col(const HostSystem& h, int p, int32_t col, int32_t* shift = nullptr) {
    int32_t sh = 0;
    if (h.rep > 1) {
        const int64_t local = (int64_t)col - h.xoff[(size_t)p], nr = h.rep_n[(size_t)p];
        if (nr > 0 && local < (int64_t)h.rep * nr) sh = (int32_t)((local / nr) * nr);
    }
    if (shift) *shift = sh;
    return col - sh;
}

#if defined(__HIPCC__)
struct LinkArgs {
    const LinkProb* probs;
    const LinkItem* items;
    const int32_t* ucol;
    const int32_t* uround;
    const int32_t* usuper;
    const uint8_t* mask;
    const int32_t* pcol;     // per unknown: the column its row / column is looked up under (K of a replicated problem stores replica
    const int32_t* pshift;   //   0's rows: link_owner_col; otherwise the unknown's own column) and the distance to it
    int32_t* pos;            // per group n_u x n_u: position of M[u_a, u_b] in the matrix's values (-1: no entry of G)
    double* Qt;              // per problem n_u x n_u: Q transposed (thread a of k_link_solve reads Qt[b n_u + a])
    double* t;               // per unknown: t = Q y[U]
    double* Zr;              // rounds x n_tot: round r's applications of the chain kernel
    double* rhs;             // rounds x n_tot, zero except for a 1 at every unknown, in its round's vector
    int64_t n_tot;
    int rounds, n_u_total;
    const int32_t* Hptr;
    const int32_t* Hcol;
    const double* Hval;
    const ChainDesc* chains;
    const int32_t* node_col;
    const int32_t* done;
    const double* r;
    double* z;
    double* p;
    double* rz_out;
    int32_t* status;         // per problem with links: 0 fine, 1 = singular capacitance matrix (Q := 0: the chain preconditioner alone)
};

__global__ __launch_bounds__(kLinkThreads) void k_link_positions(LinkArgs a, int n_probs) {
    for (int q = 0; q < n_probs; ++q) {
        const LinkProb P = a.probs[q];
        for (int e = blockIdx.x * kLinkThreads + threadIdx.x; e < P.n_u * P.n_u; e += gridDim.x * kLinkThreads) {
            const int ua = e / P.n_u, ub = e - ua * P.n_u;
            // (a loop closure couples row k with row k: both unknowns in the same replica)
            const bool same = a.pshift[P.u_begin + ua] == a.pshift[P.u_begin + ub];
            a.pos[P.q_off + e] = (a.mask[P.q_off + e] && same) ? hb_find(a.Hptr, a.Hcol, a.pcol[P.u_begin + ua], a.pcol[P.u_begin + ub]) : -1;
        }
    }
}

// the right-hand sides, once: vector r (n_tot entries each, zero-filled) holds a 1 at every unknown of round r
__global__ __launch_bounds__(kLinkThreads) void k_link_rhs(LinkArgs a) {
    const int i = blockIdx.x * kLinkThreads + threadIdx.x;
    if (i < a.n_u_total) a.rhs[(size_t)a.uround[i] * a.n_tot + a.ucol[i]] = 1.0;
}

// Q = (I + G Z[U,:])^-1 G, one workgroup of 256 per problem; dynamic LDS: G and S, n_u x n_u doubles each.
// Phase 1 (all four wavefronts): G from H's values; the nonzeros of every row of G listed (a row holds the unknowns of the
// node's link partners: a handful); S = I + G Z[U,:] with one lane per entry and the entry's few columns of Z requested together
// (the first builds walked all n_u columns with two dependent global loads per nonzero: 40-160 us of the kernel's 120-280).
// Phase 2: Gauss-Jordan with partial pivoting, rows never moved or scaled on the way (the pivot of step k is the largest entry
// of column k among the rows not used yet; a used row keeps its pivot as the only entry of its column), a lane owning columns of
// [S | G] and walking down the rows eight at a time.  ONEWAVE (n_u <= 48): wavefronts 1-3 leave after phase 1 and phase 2 runs
// on one wavefront with no block barrier at all; otherwise four wavefronts and four barriers per step.
constexpr int kLinkRowNnz = 16;
template <bool ONEWAVE>
__global__ __launch_bounds__(256) void k_link_cap(LinkArgs a) {
    constexpr int NT1 = 256, NT = ONEWAVE ? 64 : 256;
    extern __shared__ __attribute__((aligned(16))) double link_lds[];
    __shared__ int piv_of[kLinkMaxU];      // step k -> its pivot row
    __shared__ char used[kLinkMaxU];       // rows that have been a pivot
    __shared__ double fcol[kLinkMaxU];
    __shared__ int s_ucol[kLinkMaxU], s_round[kLinkMaxU], s_super[kLinkMaxU], s_cnt[kLinkMaxU];
    __shared__ unsigned char s_rowc[kLinkMaxU * kLinkRowNnz];
    __shared__ double wv[4];
    __shared__ int wi[4];
    __shared__ int bad;
    static_assert(kLinkMaxU <= 128, "the pivot search looks at 128 rows");
    const LinkProb P = a.probs[blockIdx.x];
    const int n = P.n_u, t = threadIdx.x;
    double* G = link_lds;
    double* S = link_lds + n * n;
    // ---- phase 1 ----
    for (int e = t; e < n * n; e += NT1) {
        const int pos = a.pos[P.q_off + e];
        G[e] = pos >= 0 ? a.Hval[pos] : 0.0;
    }
    if (t < n) { s_ucol[t] = a.ucol[P.u_begin + t]; s_round[t] = a.uround[P.u_begin + t]; s_super[t] = a.usuper[P.u_begin + t]; used[t] = 0; }
    if (t == 0) bad = 0;
    __syncthreads();
    if (t < n) {  // the nonzero columns of row t of G (an entry of the mask that holds a zero value is skipped as well)
        int cnt = 0;
        for (int c = 0; c < n; ++c)
            if (G[t * n + c] != 0.0) { if (cnt < kLinkRowNnz) s_rowc[t * kLinkRowNnz + cnt] = (unsigned char)c; ++cnt; }
        s_cnt[t] = cnt;
    }
    __syncthreads();
    for (int e = t; e < n * n; e += NT1) {
        const int ra = e / n, cb = e - ra * n;
        // S[ra][cb] = delta + sum_c G[ra][c] Z_cb[u_c],  Z_cb = column of unknown cb: round(cb)'s vector, nonzero on cb's chain only
        const int sup_b = s_super[cb], cnt = s_cnt[ra];
        const double* __restrict__ Zb = a.Zr + (size_t)s_round[cb] * a.n_tot;
        double acc = ra == cb ? 1.0 : 0.0;
        if (cnt <= kLinkRowNnz) {
            for (int j0 = 0; j0 < cnt; j0 += 4) {
                double zv[4], gv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = s_rowc[ra * kLinkRowNnz + min(j0 + q, cnt - 1)];
                    gv[q] = (j0 + q < cnt && s_super[c] == sup_b) ? G[ra * n + c] : 0.0;
                    zv[q] = Zb[s_ucol[c]];  // (a valid address whatever the weight)
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += gv[q] * zv[q];
            }
        } else {  // (a node with more link partners than the list holds: the plain walk)
            for (int c = 0; c < n; ++c) {
                const double g = G[ra * n + c];
                if (g != 0.0 && s_super[c] == sup_b) acc += g * Zb[s_ucol[c]];
            }
        }
        S[e] = acc;
    }
    __syncthreads();
    if (ONEWAVE && t >= 64) return;
    // ---- phase 2 ----
    auto sync = [] {
        if (ONEWAVE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
        else __syncthreads();
    };
    for (int k = 0; k < n; ++k) {
        // pivot: the largest |S[i][k]| among the rows not used yet (ties: the smallest row)
        {
            double v = -1.0;
            int idx = 0x7fffffff;
            for (int i = t; i < n; i += NT) {
                const double x = used[i] ? -1.0 : fabs(S[i * n + k]);
                if (x > v || (x == v && i < idx)) { v = x; idx = i; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_down(v, off);
                const int oi = __shfl_down(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            if ((t & 63) == 0) { wv[t >> 6] = v; wi[t >> 6] = idx; }
        }
        sync();
        if (t == 0) {
            double v = wv[0];
            int idx = wi[0];
            for (int w = 1; w < NT / 64; ++w)
                if (wv[w] > v || (wv[w] == v && wi[w] < idx)) { v = wv[w]; idx = wi[w]; }
            if (!(v > 1e-300)) bad = 1;
            piv_of[k] = idx; used[idx] = 1;
        }
        sync();
        if (bad) break;  // (uniform)
        const int pr = piv_of[k];
        const double inv = 1.0 / S[pr * n + k];
        for (int i = t; i < n; i += NT) fcol[i] = (i == pr) ? 0.0 : S[i * n + k] * inv;
        sync();
        // every other row: row_i -= (S[i][k] / pivot) * row_pr; a lane owns columns of [S | G] and walks down the rows eight at a
        // time (every read requested before the first write)
        for (int c = t; c < 2 * n; c += NT) {
            double* M = c < n ? S : G;
            const int cc = c < n ? c : c - n;
            const double pv = M[pr * n + cc];
            if (pv != 0.0) {
                constexpr int kRows = 8;
                for (int i0 = 0; i0 < n; i0 += kRows) {
                    double mv[kRows], fv[kRows];
#pragma unroll
                    for (int q = 0; q < kRows; ++q) {
                        const int i = min(i0 + q, n - 1);
                        mv[q] = M[i * n + cc]; fv[q] = fcol[i];
                    }
#pragma unroll
                    for (int q = 0; q < kRows; ++q)
                        if (i0 + q < n) M[(i0 + q) * n + cc] = mv[q] - fv[q] * pv;  // (fcol[pr] = 0: the pivot row stays)
                }
            }
        }
        sync();
    }
    const bool singular = bad != 0;
    // row piv_of[k] now holds d_k e_k' in S: unknown k's row of Q is that row of G over d_k
    for (int e = t; e < n * n; e += NT) {
        const int k = e / n, cb = e - k * n;
        double q = 0.0;
        if (!singular) { const int pr = piv_of[k]; q = G[pr * n + cb] / S[pr * n + k]; }
        a.Qt[P.q_off + cb * n + k] = q;
    }
    if (t == 0) a.status[blockIdx.x] = singular ? 1 : 0;
}

// t = Q y[U] (y = what the chain kernel left in z), one workgroup per problem with links
__global__ __launch_bounds__(128) void k_link_solve(LinkArgs a) {
    __shared__ double v[kLinkMaxU];
    const LinkProb P = a.probs[blockIdx.x];
    if (a.done[P.prob]) return;  // frozen problem / a PCG whose gate has fired: the chain kernel wrote nothing either
    const int t = threadIdx.x;
    if (t < P.n_u) v[t] = a.z[a.ucol[P.u_begin + t]];
    __syncthreads();
    if (t < P.n_u) {
        const double* __restrict__ Q = a.Qt + P.q_off;
        double acc = 0.0;
        for (int b = 0; b < P.n_u; ++b) acc += Q[b * P.n_u + t] * v[b];
        a.t[P.u_begin + t] = acc;
    }
}

// z -= Z t on one affected chain (and the separator to its right); its r'z partial sum restated with the corrected z.
// The rounds a chain carries are compacted first; an entry then requests kLinkBatch columns of Z at a time before it uses any
// (a chain is ONE workgroup's work: a loop of dependent loads over 24 rounds took 100 us -- measured on the first build).
// `tw`: the weights t by unknown -- global memory (k_link_apply) or the group's LDS copy minus `u_base` (k_link_group).
constexpr int kLinkApplyThreads = 512;
constexpr int kLinkBatch = 8;
struct LinkApplyLds {
    double red[16];
    double ts[kLinkMaxRounds + kLinkBatch];
    int rr[kLinkMaxRounds + kLinkBatch];
    int n_act;
};
// (`it` stays in global memory: a copy of the 212-byte record indexed by the lane number would live in scratch)
template <int BS, int MODE>
__device__ __forceinline__ void link_apply_chain(const LinkArgs& a, const LinkItem& it, const double* tw, int u_base, LinkApplyLds& L) {
    static_assert(kLinkMaxRounds <= 64, "the rounds of a chain are compacted by one wavefront");
    const int t = threadIdx.x;
    if (t < 64) {  // the rounds this chain carries, in order, by ballot: every round's weight requested at once
        const int u = t < a.rounds ? it.u[min(t, kLinkMaxRounds - 1)] : -1;
        const double w = u >= 0 ? tw[max(u, u_base) - u_base] : 0.0;
        const unsigned long long m = __ballot(u >= 0);
        const int k = __popcll(m & ((1ull << t) - 1ull)), n = __popcll(m);
        if (u >= 0) { L.rr[k] = t; L.ts[k] = w; }
        if (t < kLinkBatch) { L.rr[n + t] = m ? 63 - __clzll((long long)m) : 0; L.ts[n + t] = 0.0; }  // (padding: a valid column of Z, weight 0)
        if (t == 0) L.n_act = n;
    }
    __syncthreads();
    const int na = L.n_act;
    const ChainDesc ch = a.chains[it.chain];
    const int NB = ch.N * BS, NE = NB + (it.sep_col >= 0 ? BS : 0);
    const double* __restrict__ Zr = a.Zr;
    double local = 0.0;
    // kLinkEnt entries per lane and trip (a chain of 1023 nodes of 3 unknowns: one trip), every load of a batch of rounds
    // requested before the first use
    constexpr int kLinkEnt = 6;
    for (int e0 = t; e0 < NE; e0 += kLinkApplyThreads * kLinkEnt) {
        int col[kLinkEnt];
        double zz[kLinkEnt], rv[kLinkEnt];
#pragma unroll
        for (int j = 0; j < kLinkEnt; ++j) {
            const int e = min(e0 + j * kLinkApplyThreads, NE - 1);
            if (e < NB) {
                const int node = e / BS;
                col[j] = join_col<BS>(ch, a.node_col, node) + (e - node * BS);
            } else col[j] = it.sep_col + (e - NB);
        }
#pragma unroll
        for (int j = 0; j < kLinkEnt; ++j) { zz[j] = a.z[col[j]]; rv[j] = a.r[col[j]]; }
        for (int k0 = 0; k0 < na; k0 += kLinkBatch) {
            double zv[kLinkEnt][kLinkBatch];
#pragma unroll
            for (int j = 0; j < kLinkEnt; ++j)
#pragma unroll
                for (int q = 0; q < kLinkBatch; ++q) zv[j][q] = Zr[(size_t)L.rr[k0 + q] * a.n_tot + col[j]];
#pragma unroll
            for (int j = 0; j < kLinkEnt; ++j)
#pragma unroll
                for (int q = 0; q < kLinkBatch; ++q) zz[j] -= zv[j][q] * L.ts[k0 + q];
        }
#pragma unroll
        for (int j = 0; j < kLinkEnt; ++j) {
            if (e0 + j * kLinkApplyThreads < NE) {
                a.z[col[j]] = zz[j];
                if (MODE == PREC_INIT) a.p[col[j]] = zz[j];
                local += rv[j] * zz[j];
            }
        }
    }
    const double tot = block_sum_n<kLinkApplyThreads / 64>(local, L.red);
    if (t == 0) a.rz_out[it.work] = tot;
}
template <int BS, int MODE>
__global__ __launch_bounds__(kLinkApplyThreads) void k_link_apply(LinkArgs a) {
    __shared__ LinkApplyLds L;
    const LinkItem& it = a.items[blockIdx.x];
    if (a.done[it.prob]) return;
    link_apply_chain<BS, MODE>(a, it, a.t, 0, L);
}
// k_link_solve and k_link_apply in ONE launch, a workgroup per group: t = Q y[U] of the group, then its chains one after the
// other -- nobody else touches a group's chains, so nothing is read that another workgroup overwrites.  Taken when no group
// has more than kLinkGroupItems chains (HipBackend::link_apply): one launch shell (~5 us) less per application.  (Measured: a
// chain costs the workgroup ~10 us; with the three segments of a 2570-pose chain in a row the fused launch took 29-42 us against
// 12 + 5 for the two launches -- hence two chains at most; 2 x 400 poses with 2 loop closures: 4.16 -> 3.87 ms per solve.)
constexpr int kLinkGroupItems = 2;
template <int BS, int MODE>
__global__ __launch_bounds__(kLinkApplyThreads) void k_link_group(LinkArgs a) {
    __shared__ LinkApplyLds L;
    __shared__ double v[kLinkMaxU], tg[kLinkMaxU];
    const LinkProb P = a.probs[blockIdx.x];
    if (a.done[P.prob]) return;  // frozen problem / a PCG whose gate has fired: the chain kernel wrote nothing either
    const int t = threadIdx.x;
    if (t < P.n_u) v[t] = a.z[a.ucol[P.u_begin + t]];
    __syncthreads();
    if (t < P.n_u) {
        const double* __restrict__ Q = a.Qt + P.q_off;
        double acc = 0.0;
        for (int b = 0; b < P.n_u; ++b) acc += Q[b * P.n_u + t] * v[b];
        tg[t] = acc;
    }
    __syncthreads();
    for (int i = 0; i < P.item_count; ++i) {
        const LinkItem& it = a.items[P.item_begin + i];
        link_apply_chain<BS, MODE>(a, it, tg, P.u_begin, L);
        __syncthreads();
    }
}
#endif  // __HIPCC__

}  // namespace score
