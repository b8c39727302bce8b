// score_band.hpp -- band + remainder view of a matrix whose rows follow the pose chains.
//
// The reference's cost couples every pose with its chain neighbours only (relative-pose terms,
// score/utils/gurobi_utils.py:504-526; odometry is a chain, :380-404), so with the unknowns of a chain
// stored node by node the KKT operator K = P + sigma I + rho A'A -- and the Newton matrix of the polish --
// is block-tridiagonal on the chain rows (bs x bs blocks, bs = d + 1) plus a sparse remainder: the range
// couplings between a pose and a landmark / another robot's pose (:449-501) and the landmark rows.
// On the headline problem 89.5 % of K's nonzeros are band entries.
//
// This header lays such a matrix out for the SpMV kernels (k_spmv_band, score_kernels.hpp):
//   * band tiles   rows of consecutive chain nodes.  Per row NP slot PAIRS, NO column indices: pair j of a row of
//                  class c (= row within its node) multiplies the two vector entries at node_first_col +
//                  off[c][j] and + 1 (one 16-byte load); the window offsets actually used are collected per class
//                  and covered by pairs (2-D SCORE rows: 4 pairs for 7-8 of the 9 window positions; 3-D rows: 6 for
//                  9-11 of 12).  Values are stored pair-major per tile (NP x 256 x 2 doubles), so a wavefront's
//                  loads are 1 KiB contiguous.  What a band row holds outside its window (range couplings) forms
//                  the tile's REMAINDER: (column, value) pairs, row by row, padded to a multiple of 64, with a
//                  (first, count) word per row.
//   * diag tiles   rows whose only entry is their diagonal (the SOCP distance variables): one value per row.
//   * CSR tiles    everything else (landmark rows), exactly the row blocks of the CSR-stream kernel, read
//                  from the source arrays.
// The value array V = [band tiles | diag tiles | remainder entries] is a second copy of the matrix values,
// filled on the device through `dst` (source entry -> position in V); rows served by CSR tiles keep -1.
// Results are deterministic: a row adds its band slots in slot order, then its remainder entries in CSR
// order.  The reference has no counterpart (its solve is Gurobi's, score/solve_score.py:76).
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "score_host.hpp"

namespace score {

constexpr int kBandMaxS = 12;        // slots per band row: 3 * bs for bs <= 4
constexpr int kBandLanes = 256;      // lanes (= row slots) of a band tile
constexpr int kBandRemMax = 512;     // remainder entries per band tile (2 per lane); a tile ends early where its rows hold more
constexpr int kBandDiagRows = 1024;  // rows per diag tile (4 per lane, one trip)
constexpr int kBandCsrNnz = 512;     // nonzeros per CSR tile of a band view (2 per lane)

enum { BAND_KIND_CSR = 0, BAND_KIND_BAND = 1, BAND_KIND_DIAG = 2 };

// rows [r0, r1): consecutive chain nodes of bs rows each (one or more chains back to back)
struct BandRun { int64_t r0, r1; int32_t prob; int32_t rs; };

struct BandLayout {
    bool on = false;
    int bs = 0, S = 0;              // S = 2 * pairs: value slots per band row
    int32_t offw[kBandMaxS] = {0};  // pair j: window offset (relative to the node's first column) of its first slot, class c in byte c
    // unified tile list, problem-major: CSR tiles (the long rows: slowest first), band tiles, diag tiles of every problem
    //   meta  band {r0, r1, first value in V, first remainder entry (relative to rem0)}
    //         diag {r0, r1, first value in V, 0}      CSR {r0, r1, first nonzero, end nonzero} in the source arrays
    //   meta2 band {remainder entries (multiple of 64), run begin, run end, ordinal | kind << 28}; others {0, 0, 0, kind << 28}
    std::vector<int32_t> meta, meta2;     // 4 per tile
    std::vector<int32_t> lng;             // 4 per tile: split long rows among the CSR tiles (RowBlocks::lfirst .. lid; first block in THIS list)
    int n_long = 0, n_long_slots = 0;
    std::vector<int32_t> prob, rs;        // per tile; rs: replica stride of replicated rows (0 = plain)
    std::vector<int32_t> part_ptr;        // count + 1: tile range of each problem
    std::vector<int32_t> rem_col;         // per remainder entry
    std::vector<int32_t> rowseg;          // per band tile x kBandLanes: (first << 16) | count
    std::vector<int32_t> dst;             // per source entry: position in V, -1 = served from the source arrays
    int64_t v_size = 0, rem0 = 0;         // V: [band | diag | remainder); rem0 = first remainder entry
    int n_band = 0, n_diag = 0, n_csr = 0;
    std::vector<double> bytes;            // per problem: algorithmic bytes of one application (matrix stream + p + w)
    int nb() const { return (int)prob.size(); }
};

// The chains (regular column stride == bs) of every problem, merged into runs of back-to-back nodes.
// `use[ci]` selects the chains whose rows the matrix holds (replicated K: the owners).
inline std::vector<BandRun> band_runs(const std::vector<ChainDesc>& chains, const std::vector<char>& use, int bs, int rep,
                                      const std::vector<int64_t>& rep_n, bool replicated) {
    std::vector<BandRun> cand;
    for (size_t ci = 0; ci < chains.size(); ++ci) {
        if (!use[ci]) continue;
        const ChainDesc& ch = chains[ci];
        if (ch.N < 1 || (ch.N >= 2 && ch.col_stride != bs)) continue;
        const int32_t rs = (replicated && rep > 1) ? (int32_t)rep_n[(size_t)ch.prob] : 0;
        cand.push_back(BandRun{(int64_t)ch.col0, (int64_t)ch.col0 + (int64_t)ch.N * bs, ch.prob, rs});
    }
    std::sort(cand.begin(), cand.end(), [](const BandRun& a, const BandRun& b) { return a.prob != b.prob ? a.prob < b.prob : a.r0 < b.r0; });
    std::vector<BandRun> runs;
    for (const BandRun& c : cand) {
        if (!runs.empty() && runs.back().prob == c.prob && runs.back().r1 == c.r0) runs.back().r1 = c.r1;
        else runs.push_back(c);
    }
    return runs;
}

// M: the source matrix (pattern only is read).  segs: the row ranges M holds, problem by problem (plain rows, or the
// rows of replica 0 with their replica stride) -- what make_rowblocks would tile.  runs: band_runs(), every run inside
// one segment.  Returns a layout with on == false when the matrix does not fit (no runs, a node with more than
// kBandRemMax remainder entries, indices beyond 28 / 31 bits).
inline BandLayout build_band_layout(const Csr& M, const std::vector<RowSegment>& segs, const std::vector<BandRun>& runs, int bs, int count,
                                    int csr_tile_nnz = kBandCsrNnz) {
    BandLayout L;
    L.bs = bs;
    if (runs.empty() || bs < 1 || bs > kMaxBs) return L;
    PhaseTimer bt_(trace_on("band"));
    const int32_t* ptr = M.ptr.data();
    const int32_t* col = M.col.data();
    // ---- pass 1: window offsets in use, per class ----
    const int W = 3 * bs;  // window positions: offsets -bs .. 2 bs - 1
    std::vector<uint32_t> masks(kMaxBs, 0u);
    std::vector<int64_t> node0((size_t)runs.size() + 1, 0);  // node numbering over all runs
    for (size_t r = 0; r < runs.size(); ++r) node0[r + 1] = node0[r] + (runs[r].r1 - runs[r].r0) / bs;
    const int64_t n_nodes = node0.back();
    std::vector<int32_t> node_rem((size_t)n_nodes, 0);  // remainder entries per node
    // (one parallel sweep over the nodes of all runs: a batch is many short runs)
    parallel_ranges(n_nodes, 1024, [&](int, int64_t g0, int64_t g1) {
        uint32_t mk[kMaxBs] = {0, 0, 0, 0};
        size_t r = (size_t)(std::upper_bound(node0.begin(), node0.end(), g0) - node0.begin()) - 1;
        for (int64_t g = g0; g < g1; ++g) {
            while (g >= node0[r + 1]) ++r;
            const BandRun& R = runs[r];
            const int64_t nb = R.r0 + (g - node0[r]) * bs;
            int32_t rem = 0;
            for (int c = 0; c < bs; ++c)
                for (int k = ptr[nb + c]; k < ptr[nb + c + 1]; ++k) {
                    const int64_t w = (int64_t)col[k] - nb;
                    if (w >= -bs && w < 2 * bs && col[k] >= R.r0 && col[k] < R.r1) mk[c] |= 1u << (int)(w + bs);
                    else ++rem;
                }
            node_rem[(size_t)g] = rem;
        }
        for (int c = 0; c < bs; ++c) __atomic_fetch_or(&masks[c], mk[c], __ATOMIC_RELAXED);
    });
    bt_.mark("band: pass 1");
    uint32_t cls_mask[kMaxBs] = {0, 0, 0, 0};
    for (size_t r = 0; r < runs.size(); ++r)
        if (runs[r].r1 - runs[r].r0 < 2) return L;  // (the pair loads clamp to [run begin, run end - 2])
    for (int c = 0; c < bs; ++c) cls_mask[c] = masks[c];
    // cover the used offsets of every class by pairs (w, w + 1), greedily from the left; classes with fewer pairs are
    // padded with their own diagonal (slots that stay zero)
    int8_t pair_off[kMaxBs][kBandMaxS / 2];
    int8_t slot_of[kMaxBs][kBandMaxS + 1];  // window position -> slot
    int npairs = 1;
    for (int c = 0; c < kMaxBs; ++c) {
        for (int j = 0; j < kBandMaxS / 2; ++j) pair_off[c][j] = (int8_t)std::min(c, bs - 1);
        for (int w = 0; w <= kBandMaxS; ++w) slot_of[c][w] = -1;
    }
    for (int c = 0; c < bs; ++c) {
        cls_mask[c] |= 1u << (c + bs);  // (the diagonal is always present)
        int j = 0;
        for (int w = 0; w < W; ++w) {
            if (!(cls_mask[c] & (1u << w)) || slot_of[c][w] >= 0) continue;
            pair_off[c][j] = (int8_t)(w - bs);
            slot_of[c][w] = (int8_t)(2 * j);
            slot_of[c][w + 1] = (int8_t)(2 * j + 1);
            ++j;
        }
        npairs = std::max(npairs, j);
    }
    npairs = std::max(npairs, 4);
    L.S = 2 * npairs;
    if (L.S > kBandMaxS) return L;
    for (int j = 0; j < kBandMaxS / 2; ++j) {
        uint32_t wv = 0;
        for (int c = 0; c < kMaxBs; ++c) wv |= (uint32_t)(uint8_t)pair_off[c][j] << (8 * c);
        L.offw[j] = (int32_t)wv;
    }
    // ---- tiles: greedy over the nodes of a run (<= tile_rows rows, <= kBandRemMax remainder entries) ----
    const int tile_nodes = kBandLanes / bs;
    struct BT { int64_t r0, r1; int32_t prob, rs, rem, run; int64_t lo, hi; };
    std::vector<BT> bts;
    for (size_t r = 0; r < runs.size(); ++r) {
        const BandRun& R = runs[r];
        const int64_t nn = (R.r1 - R.r0) / bs;
        int64_t j = 0;
        while (j < nn) {
            int64_t j1 = j;
            int32_t rem = 0;
            while (j1 < nn && j1 - j < tile_nodes) {
                const int32_t nr = node_rem[(size_t)(node0[r] + j1)];
                if (nr > kBandRemMax) return L;  // one node beyond a tile's remainder capacity: the CSR kernels serve this matrix
                if (rem + nr > kBandRemMax) break;
                rem += nr;
                ++j1;
            }
            bts.push_back(BT{R.r0 + j * bs, R.r0 + j1 * bs, R.prob, R.rs, rem, (int32_t)r, R.r0, R.r1});
            j = j1;
        }
    }
    bt_.mark("band: tiles");
    // ---- rows outside the runs: diag tiles (rows holding only their diagonal) and CSR tiles ----
    struct Piece { int64_t r0, r1; int32_t prob, rs; bool diag; };
    std::vector<Piece> pieces;
    {
        size_t ri = 0;
        for (const RowSegment& sg : segs) {
            int64_t r = sg.begin;
            auto emit_gap = [&](int64_t g0, int64_t g1) {  // rows [g0, g1) outside every run: split into diag / general stretches
                int64_t a = g0;
                while (a < g1) {
                    const bool dg = (ptr[a + 1] - ptr[a] == 1) && col[ptr[a]] == (int32_t)a;
                    int64_t b = a + 1;
                    while (b < g1 && (((ptr[b + 1] - ptr[b] == 1) && col[ptr[b]] == (int32_t)b) == dg)) ++b;
                    // (short diagonal stretches stay with their CSR neighbours: a tile per handful of rows costs more than it saves)
                    pieces.push_back(Piece{a, b, sg.prob, sg.rs, dg && (b - a) >= 64 && sg.rs == 0});  // (diag tiles: plain rows only)
                    a = b;
                }
            };
            while (ri < runs.size() && runs[ri].prob == sg.prob && runs[ri].r0 >= sg.begin && runs[ri].r1 <= sg.end) {
                if (runs[ri].r0 > r) emit_gap(r, runs[ri].r0);
                r = runs[ri].r1;
                ++ri;
            }
            if (r < sg.end) emit_gap(r, sg.end);
        }
        if (ri != runs.size()) return L;  // a run outside the stored rows: not a layout this view describes
    }
    // merge neighbouring non-diag pieces of one segment kind
    std::vector<RowSegment> csr_segs;
    for (const Piece& pc : pieces)
        if (!pc.diag) {
            if (!csr_segs.empty() && csr_segs.back().end == pc.r0 && csr_segs.back().prob == pc.prob && csr_segs.back().rs == pc.rs) csr_segs.back().end = pc.r1;
            else csr_segs.push_back(RowSegment{pc.r0, pc.r1, pc.prob, pc.rs});
        }
    bt_.mark("band: pieces");
    const RowBlocks rbc = make_rowblocks(M, csr_segs, count, csr_tile_nnz);
    bt_.mark("band: csr row blocks");
    L.n_long = rbc.n_long; L.n_long_slots = rbc.n_long_slots;
    // ---- V layout ----
    L.n_band = (int)bts.size();
    const int64_t band_doubles = (int64_t)L.n_band * L.S * kBandLanes;
    int64_t diag_doubles = 0;
    for (const Piece& pc : pieces)
        if (pc.diag) diag_doubles += pc.r1 - pc.r0;
    std::vector<int64_t> rem_off((size_t)L.n_band + 1, 0);
    for (int b = 0; b < L.n_band; ++b) rem_off[(size_t)b + 1] = rem_off[(size_t)b] + ((bts[(size_t)b].rem + 63) & ~63);
    L.rem0 = band_doubles + diag_doubles;
    L.v_size = L.rem0 + rem_off.back();
    if (L.v_size >= ((int64_t)1 << 31) || L.n_band >= (1 << 28) || M.nrows >= ((int64_t)1 << 30)) return L;
    L.rem_col.assign((size_t)rem_off.back(), 0);
    L.rowseg.assign((size_t)L.n_band * kBandLanes, 0);
    L.dst.assign(M.col.size(), -1);
    bt_.mark("band: assign");
    // ---- fill the band tiles ----
    parallel_ranges(L.n_band, 8, [&](int, int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            const BT& T = bts[(size_t)b];
            const int64_t vb = b * L.S * kBandLanes;
            int32_t k_rem = 0;
            for (int64_t row = T.r0; row < T.r1; ++row) {
                const int t = (int)(row - T.r0);
                const int c = t % bs;
                const int64_t nb = row - c;
                const int32_t first = k_rem;
                for (int k = ptr[row]; k < ptr[row + 1]; ++k) {
                    const int64_t w = (int64_t)col[k] - nb;
                    if (w >= -bs && w < 2 * bs && col[k] >= T.lo && col[k] < T.hi) {
                        const int sl = slot_of[c][w + bs];
                        L.dst[(size_t)k] = (int32_t)(vb + (int64_t)(sl >> 1) * 2 * kBandLanes + 2 * t + (sl & 1));
                    } else {
                        L.rem_col[(size_t)(rem_off[(size_t)b] + k_rem)] = col[k];
                        L.dst[(size_t)k] = (int32_t)(L.rem0 + rem_off[(size_t)b] + k_rem);
                        ++k_rem;
                    }
                }
                L.rowseg[(size_t)b * kBandLanes + t] = (first << 16) | (k_rem - first);
            }
            for (int64_t k = rem_off[(size_t)b] + k_rem; k < rem_off[(size_t)b + 1]; ++k) L.rem_col[(size_t)k] = (int32_t)T.r0;  // (value 0)
        }
    });
    bt_.mark("band: fill");
    // ---- the unified tile list, problem by problem ----
    L.part_ptr.assign((size_t)count + 1, 0);
    L.bytes.assign((size_t)count, 0.0);
    size_t bi = 0, pi = 0;
    int ci = 0;
    int64_t diag_at = band_doubles;
    auto push = [&](int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t b0, int32_t b1, int32_t b2, int32_t b3, int32_t prob, int32_t rs) {
        L.meta.insert(L.meta.end(), {a0, a1, a2, a3});
        L.meta2.insert(L.meta2.end(), {b0, b1, b2, b3});
        L.lng.insert(L.lng.end(), {0, 0, 0, 0});
        L.prob.push_back(prob);
        L.rs.push_back(rs);
    };
    for (int p = 0; p < count; ++p) {
        double by = 0.0;
        for (; ci < rbc.nb() && rbc.prob[(size_t)ci] == p; ++ci) {
            const int32_t r0 = rbc.first_row[(size_t)ci], r1 = rbc.end_row[(size_t)ci];
            const bool seg = rbc.kbeg[(size_t)ci] >= 0;
            const int32_t k0 = seg ? rbc.kbeg[(size_t)ci] : ptr[r0], k1 = seg ? rbc.kend[(size_t)ci] : ptr[r1];
            push(r0, r1, k0, k1, 0, 0, 0, BAND_KIND_CSR << 28, p, rbc.rs[(size_t)ci]);
            if (seg) {  // (first block of the row: rbc numbers its blocks from 0; here the CSR tiles of a problem follow part_ptr[p])
                int32_t* lg = &L.lng[L.lng.size() - 4];
                lg[0] = (int32_t)L.prob.size() - 1 - (ci - rbc.lfirst[(size_t)ci]); lg[1] = rbc.lseg[(size_t)ci]; lg[2] = rbc.lbase[(size_t)ci]; lg[3] = rbc.lid[(size_t)ci];
            }
            by += 12.0 * (k1 - k0) + 4.0 * (r1 - r0 + 1) + 32.0;
            ++L.n_csr;
        }
        for (; bi < bts.size() && bts[bi].prob == p; ++bi) {
            const BT& T = bts[bi];
            const int32_t padded = (int32_t)(rem_off[bi + 1] - rem_off[bi]);
            push((int32_t)T.r0, (int32_t)T.r1, (int32_t)((int64_t)bi * L.S * kBandLanes), (int32_t)rem_off[bi], padded, (int32_t)T.lo, (int32_t)T.hi,
                 (int32_t)bi | (BAND_KIND_BAND << 28), p, T.rs);
            by += 8.0 * L.S * kBandLanes + 12.0 * padded + 4.0 * kBandLanes + 32.0;
        }
        for (; pi < pieces.size() && pieces[pi].prob <= p; ++pi) {
            const Piece& pc = pieces[pi];
            if (!pc.diag || pc.prob != p) continue;
            for (int64_t r = pc.r0; r < pc.r1; r += kBandDiagRows) {
                const int64_t r1 = std::min<int64_t>(r + kBandDiagRows, pc.r1);
                push((int32_t)r, (int32_t)r1, (int32_t)diag_at, 0, 0, 0, 0, BAND_KIND_DIAG << 28, p, pc.rs);
                for (int64_t row = r; row < r1; ++row) L.dst[(size_t)ptr[row]] = (int32_t)(diag_at + (row - r));
                diag_at += r1 - r;
                by += 8.0 * (r1 - r) + 32.0;
                ++L.n_diag;
            }
        }
        L.part_ptr[(size_t)p + 1] = (int32_t)L.prob.size();
        L.bytes[(size_t)p] = by;
    }
    bt_.mark("band: tile list");
    L.on = true;
    return L;
}

// Host application of a layout (the specification of k_spmv_band; the CPU twin checks the layout with it):
// y[row + q rs] = (M x)[row + q rs] for every stored row, values taken from V (band / diag / remainder) or from
// the source arrays (CSR tiles).  V must have been filled through dst.
inline void band_apply_host(const BandLayout& L, const Csr& M, const double* V, const double* srcval, int NR, const double* x, double* y) {
    for (int b = 0; b < L.nb(); ++b) {
        const int32_t* m = &L.meta[(size_t)4 * b];
        const int32_t* m2 = &L.meta2[(size_t)4 * b];
        const int kind = (int)((uint32_t)m2[3] >> 28);
        const int rs = L.rs[(size_t)b];
        const int nrep = rs > 0 ? NR : 1;
        for (int row = m[0]; row < m[1]; ++row)
            for (int q = 0; q < nrep; ++q) {
                double acc = 0.0;
                const int t = row - m[0];
                if (kind == BAND_KIND_BAND) {
                    const int c = t % L.bs, nb = row - c;
                    for (int j = 0; j < L.S / 2; ++j) {  // one pair: the two entries at a base clamped to the run, see band_tile
                        const int o = (int)(int8_t)(((uint32_t)L.offw[j] >> (8 * c)) & 0xff);
                        const int base = nb + o, cb = std::min(std::max(base, m2[1]), m2[2] - 2), d = base - cb;
                        const double lx = x[cb + (size_t)q * rs], ly = x[cb + 1 + (size_t)q * rs];
                        const double x0 = d > 0 ? ly : lx, x1 = d < 0 ? lx : ly;
                        acc += V[(size_t)m[2] + (size_t)j * 2 * kBandLanes + 2 * t] * x0;
                        acc += V[(size_t)m[2] + (size_t)j * 2 * kBandLanes + 2 * t + 1] * x1;
                    }
                    const int ord = m2[3] & 0x0fffffff;
                    const int seg = L.rowseg[(size_t)ord * kBandLanes + t];
                    for (int k = seg >> 16; k < (seg >> 16) + (seg & 0xffff); ++k)
                        acc += V[(size_t)L.rem0 + m[3] + k] * x[L.rem_col[(size_t)m[3] + k] + (size_t)q * rs];
                } else if (kind == BAND_KIND_DIAG) {
                    acc = V[(size_t)m[2] + t] * x[row + (size_t)q * rs];
                } else {  // (a segment of a split long row adds its part)
                    const bool whole = m[2] == M.ptr[row] || m[1] - m[0] > 1;
                    for (int k = (m[1] - m[0] > 1 ? M.ptr[row] : m[2]); k < (m[1] - m[0] > 1 ? M.ptr[row + 1] : m[3]); ++k) acc += srcval[k] * x[M.col[k] + (size_t)q * rs];
                    if (!whole) acc += y[row + (size_t)q * rs];
                }
                y[row + (size_t)q * rs] = acc;
            }
    }
}

}  // namespace score
