// score_split.hpp -- host-side plans of the split chain kernel (k_prec_wave, score_prec_wave.hpp).
//
// One 512-thread workgroup per chain leaves most of the machine idle on a single problem (40 chains
// on 256 CUs) and pulls a whole chain's factor (~250 KB) through one CU.  The nested-dissection
// factor already contains the decomposition that fixes this: the nodes of a chain's LAST level (1-3
// separators, 256 poses apart for radix 4) cut the chain into 2-4 parts, and every lower level of the
// factorisation is local to a part -- a run never crosses a higher-level separator.  The only coupling
// between parts is the reduced right-hand side of those top separators (a sum of per-level
// contributions from the two adjacent parts) and their solution.  So a part can be solved by its own
// workgroup -- here a single wavefront: a part has at most 64 runs per level, every phase fits one
// wave and needs no block barrier -- with ONE in-kernel exchange of 3 x (d+1) doubles per part.
// The factor data (`fac`, produced on the device by k_factor) are used exactly as they are.
//
// A plan describes one (chain length, part) combination: the part's node / run ranges on every level,
// where the coarse-level factor blocks it needs sit in `fac` (a gather list, relative to the chain's
// first factor double) and how its dynamic LDS is laid out.
#pragma once

#include <cstdint>
#include <vector>

#include "score_host.hpp"

namespace score {

constexpr int kSplitMaxLevels = 6;  // local levels of a part: <= 5 for chains of <= 1023 nodes
constexpr int kSplitMaxParts = 4;
constexpr int kSplitSlotDoubles = 16;  // exchange slot per part: cL | cR | r_top (3 x bs <= 9 doubles)

struct SplitLevel {
    int32_t n;       // nodes of the part on this level
    int32_t nr;      // runs: n / 4 + 1 (the last one may be empty)
    int32_t g0;      // first node (index on the chain's level)
    int32_t jr0;     // first run (index on the chain's level)
    int32_t P;       // node positions per run in the R array (3; N_last on a chain's last level)
    int32_t offR, offS, offB;  // staged factor offsets (levels >= 1): R[(sb*P + q)*nr + j'], S[sb*(nr+1) + s' + 1], B[sb*n + i']
    int32_t voff;    // LDS offset of the level's vector; node i' (halo -1 .. n) at voff + (i'+4)*bs + ((i'+4) >> 2)
    int32_t pad_;
};

struct SplitPlan {
    int32_t n_levels;            // local levels L >= 1
    int32_t nparts, part;
    int32_t has_left, has_right; // a top separator on that side
    int32_t n_top;               // nodes of the chain's last level (0: the part is the whole chain)
    int32_t top_off;             // staged offset of the last level's run blocks (Lf, Dinv; P = n_top, one run)
    int32_t n_stage, stage_begin;
    int32_t xch_off;             // LDS: couplings handed from a level-0 separator to the run on its right (nr0 + 1 nodes)
    int32_t misc_off;            // LDS: accL | accR | x_top (kSplitMaxParts - 1 nodes) | exchange copy
    int32_t lfac_off;            // LDS: staged factors
    int32_t lds_doubles;
    int32_t pad_[3];
    SplitLevel lv[kSplitMaxLevels + 1];  // [n_levels] is the virtual level above the local top (halo only)
    // level-0 register tile: entry k of lane t comes from fac[chain base + row_base[k] + min(min(t, nr0 - 1), row_cmax[k])]
    // (row_cmax < 0: zero).  k: run blocks Lf / Dinv (6 b2), separator blocks Cl / Cr (2 b2), Cr of the
    // separator on the part's left (b2).  Regular enough to need no gather list.
    int32_t row_base[81];
    int32_t row_cmax[81];
    int32_t pad2_[2];
};

struct SplitItem {       // per work item of the split launch (chain parts only; Jacobi items keep PrecWork)
    int32_t chain;       // chain id
    int32_t plan;        // plan id
    int32_t slot;        // exchange slot of the chain
    int32_t part;
};

struct SplitSystem {
    bool active = false;
    std::vector<SplitPlan> plans;
    std::vector<int32_t> stage_rel;      // gather list: offsets relative to the chain's first factor double (-1: zero)
    std::vector<PrecWork> work;          // kind 0: chain part (index = entry in items), kind 1: Jacobi block
    std::vector<SplitItem> items;
    std::vector<int32_t> part_ptr;       // count + 1: work items per problem
    int n_slots = 0;                     // chains with more than one part
    size_t max_lds_doubles = 0;
};

// plan of part q of a chain with the given (chain-relative) level descriptors
inline SplitPlan make_split_plan(const std::vector<ChainLevelDesc>& lv, int bs, int nparts, int q, std::vector<int32_t>& stage_rel) {
    const int b2 = bs * bs;
    const int nl = (int)lv.size();
    SplitPlan P{};
    P.nparts = nparts; P.part = q;
    P.has_left = (nparts > 1 && q > 0) ? 1 : 0;
    P.has_right = (nparts > 1 && q < nparts - 1) ? 1 : 0;
    P.n_top = nparts > 1 ? lv[nl - 1].N : 0;
    const int64_t base = lv[0].offR;  // chain's first factor double
    // local levels
    int L = 0;
    int lds = 16;  // [0, 16): scratch
    const int gl_levels = nparts > 1 ? nl - 1 : nl;  // chain levels that hold part-local work
    for (int l = 0; l < gl_levels; ++l) {
        const ChainLevelDesc& G = lv[l];
        const bool last = (G.p == 0);
        SplitLevel& S = P.lv[L];
        int64_t g0 = 0, end = G.N;
        if (nparts > 1) {
            int64_t span = 1;
            for (int k = 0; k < nl - 1 - l; ++k) span *= 4;
            g0 = (int64_t)q * span;
            end = std::min<int64_t>(G.N, (int64_t)(q + 1) * span - 1);
        }
        if (end <= g0) break;  // the (short, last) part has nothing on this level
        S.n = (int32_t)(end - g0);
        S.g0 = (int32_t)g0;
        S.jr0 = (int32_t)(g0 / 4);
        S.P = last ? G.N : 3;
        S.nr = last ? 1 : S.n / 4 + 1;
        S.voff = lds;
        lds += (S.n + 6) * bs + (S.n + 6) / 4 + 2;
        ++L;
        if (S.n <= 3) break;  // a single run: the part's top level
    }
    P.n_levels = L;
    {   // virtual level above the local top: halo only
        SplitLevel& S = P.lv[L];
        S.n = 0; S.nr = 0; S.voff = lds;
        lds += 6 * bs + 3;
    }
    P.xch_off = lds; lds += (P.lv[0].nr + 2) * bs;
    P.misc_off = lds; lds += 2 * bs + (kSplitMaxParts - 1) * bs + kSplitMaxParts * kSplitSlotDoubles + 8;
    P.lfac_off = lds;
    // level-0 register tile (see SplitPlan::row_base)
    P.stage_begin = (int32_t)stage_rel.size();
    {
        const ChainLevelDesc& G = lv[0];
        const SplitLevel& S = P.lv[0];
        for (int k = 0; k < 81; ++k) { P.row_base[k] = 0; P.row_cmax[k] = -1; }
        for (int k = 0; k < 9 * b2 && k < 81; ++k) {
            if (k < 6 * b2) {
                const int slot = k / (3 * b2), qe = k % (3 * b2), qpos = qe / b2, e = qe % b2;
                const int qq = std::min(qpos, G.P - 1);
                P.row_base[k] = (int32_t)(G.offR - base + ((int64_t)(slot * b2 + e) * G.P + qq) * G.nruns + S.jr0);
                P.row_cmax[k] = S.nr - 1;
            } else if (k < 8 * b2) {
                const int slot = (k - 6 * b2) / b2, e = (k - 6 * b2) % b2;
                if (G.nsep > S.jr0) {
                    P.row_base[k] = (int32_t)(G.offS - base + (int64_t)(slot * b2 + e) * G.nsep + S.jr0);
                    P.row_cmax[k] = std::min(S.nr - 1, G.nsep - 1 - S.jr0);
                }
            } else {
                const int e = k - 8 * b2;
                if (G.nsep > 0 && S.jr0 >= 1) {
                    P.row_base[k] = (int32_t)(G.offS - base + (int64_t)(b2 + e) * G.nsep + (S.jr0 - 1));
                    P.row_cmax[k] = 0;
                }
            }
        }
    }
    // staged factors (gather list): levels >= 1 and the chain's last level (top system)
    int off = 0;
    for (int l = 1; l < L; ++l) {
        const ChainLevelDesc& G = lv[l];
        SplitLevel& S = P.lv[l];
        S.offR = off;
        for (int sb = 0; sb < 2 * b2; ++sb)
            for (int qq = 0; qq < S.P; ++qq)
                for (int j = 0; j < S.nr; ++j) {
                    const int64_t gj = S.jr0 + j;
                    const bool ok = gj < G.nruns;
                    stage_rel.push_back(ok ? (int32_t)(G.offR - base + ((int64_t)sb * G.P + qq) * G.nruns + gj) : -1);
                    ++off;
                }
        S.offS = off;
        for (int sb = 0; sb < 2 * b2; ++sb)
            for (int s = 0; s <= S.nr; ++s) {  // local separator s - 1  <->  chain separator jr0 - 1 + s
                const int64_t gs = (int64_t)S.jr0 - 1 + s;
                const bool ok = gs >= 0 && gs < G.nsep;
                stage_rel.push_back(ok ? (int32_t)(G.offS - base + (int64_t)sb * G.nsep + gs) : -1);
                ++off;
            }
        S.offB = off;
        const bool last = (G.p == 0);
        for (int sb = 0; sb < 2 * b2; ++sb)
            for (int i = 0; i < S.n; ++i) {
                // (a chain's last level has no spikes: never read)
                stage_rel.push_back(last ? -1 : (int32_t)(G.offB - base + (int64_t)sb * G.N + S.g0 + i));
                ++off;
            }
    }
    P.top_off = off;
    if (nparts > 1) {
        const ChainLevelDesc& G = lv[nl - 1];
        for (int sb = 0; sb < 2 * b2; ++sb)
            for (int qq = 0; qq < G.N; ++qq) {
                stage_rel.push_back((int32_t)(G.offR - base + ((int64_t)sb * G.P + qq) * G.nruns));
                ++off;
            }
    }
    P.n_stage = off;
    lds += off + 81 * 64 + 2;
    P.lds_doubles = lds;
    return P;
}

// Decide whether the split launch pays for this system and build its work list.  Conditions: block
// size 3 (2-D poses), radix 4, every chain within the lane budget of one wave per part (<= 64 runs
// per level and part: chains of <= 1023 nodes), and the whole launch resident at once (<= max_groups
// workgroups): parts of a chain wait for each other inside the kernel.
inline void build_split_system(const HostSystem& H, int max_groups, SplitSystem& S) {
    S = SplitSystem();
    if (H.bs != 3 || H.radix != 4 || H.chains.empty()) return;
    struct Key { int N; int first_plan; int nparts; };
    std::vector<Key> keys;
    size_t total = 0;
    for (const auto& ch : H.chains) {
        if (ch.n_levels > 5 || ch.col_stride != H.bs) return;  // contiguous chains only
        const ChainLevelDesc& last = H.levels[ch.level_begin + ch.n_levels - 1];
        int nparts = 1;
        if (ch.n_levels >= 3 && ch.N >= 256) nparts = last.N + 1;
        if (nparts > kSplitMaxParts) return;
        if (nparts == 1 && ch.N > 255) return;  // one wave per part: <= 64 runs on level 0
        total += nparts;
    }
    total += H.prec_work.size() - H.chains.size();  // Jacobi items
    if ((int)total > max_groups) return;
    S.part_ptr.assign(H.count + 1, 0);
    size_t wi = 0;
    for (int p = 0; p < H.count; ++p) {
        for (; wi < H.prec_work.size() && H.prec_work[wi].prob == p; ++wi) {
            const PrecWork& w = H.prec_work[wi];
            if (w.kind != 0) { S.work.push_back(w); continue; }
            const ChainDesc& ch = H.chains[w.index];
            const Key* key = nullptr;
            for (const auto& k : keys) if (k.N == ch.N) key = &k;
            if (!key) {
                std::vector<ChainLevelDesc> rel(H.levels.begin() + ch.level_begin, H.levels.begin() + ch.level_begin + ch.n_levels);
                const ChainLevelDesc& last = rel.back();
                const int nparts = (ch.n_levels >= 3 && ch.N >= 256) ? last.N + 1 : 1;
                Key k{ch.N, (int)S.plans.size(), nparts};
                for (int q = 0; q < nparts; ++q) {
                    S.plans.push_back(make_split_plan(rel, H.bs, nparts, q, S.stage_rel));
                    S.max_lds_doubles = std::max<size_t>(S.max_lds_doubles, (size_t)S.plans.back().lds_doubles);
                }
                keys.push_back(k);
                key = &keys.back();
            }
            const int slot = key->nparts > 1 ? S.n_slots++ : -1;
            for (int q = 0; q < key->nparts; ++q) {
                S.work.push_back(PrecWork{0, (int32_t)S.items.size(), 0, p});
                S.items.push_back(SplitItem{w.index, key->first_plan + q, slot, q});
            }
        }
        S.part_ptr[p + 1] = (int32_t)S.work.size();
    }
    S.active = true;
}

}  // namespace score
