// score_host.hpp -- host-side setup for the MI355X SCORE conic solver.
//
// Everything the device needs that is computed ONCE per problem (or once per
// penalty update) lives here, in plain C++17 with no HIP dependency:
//   * Ruiz equilibration of [[P, A'], [A, 0]] with one scale per cone,
//   * the KKT operator K = P + sigma*I + rho*A'A on a fixed sparsity pattern
//     (values K0 + rho*K1), and the two "wide" operators
//       G1 = [ 0 | A']  applied to [xt ; u]   (A'u of the KKT right-hand side; K xt is
//                                              carried incrementally, see kx)
//       G2 = [ P | A']  applied to [x  ; y]   (dual residual),
//   * the row-block tiling the CSR-stream SpMV kernel consumes,
//   * the multi-level (radix-p nested-dissection) factorisation of the
//     per-chain block-tridiagonal preconditioner.
// The reference has no counterpart for any of this: its solve happens inside
// Gurobi (score/solve_score.py:76).  The problem it receives is the one
// score/utils/gurobi_utils.py:173-187 builds.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>

#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <vector>

#include "../../include/score_hip.h"

namespace score {

constexpr int kRowsPerBlock = 256;  // rows (= threads) per SpMV workgroup
constexpr int kTileNnz = 2048;      // products staged in LDS per workgroup (16 KiB); 2048 measured best of 1536..4608
constexpr int kLongRow = 48;        // rows longer than this get a workgroup of their own
constexpr int kLongSeg = 512;       // ... and beyond this many nonzeros several: segments of <= kLongSeg (2 per lane), whose sums the
                                    // last segment to finish adds in segment order (spmv_tile).  A landmark seen by 2000 ranges was the
                                    // last workgroup of every K product to leave (7.5 us against 3.8 us for the median tile)
constexpr int kLongSegMax = 256;    // segments per row at most (longer rows: longer segments)
constexpr int kLongVals = 4;        // sums a segment publishes (up to 3 right-hand sides, or P and A' parts)
constexpr int kConesPerBlock = 256;
constexpr int kMaxBs = 4;

// Diagnostics on stderr / stdout, ONE switch: SCORE_TRACE=<comma-separated list> of
//   cache (every raw hipMalloc / hipHostMalloc of the block cache), host (queueing and wait times of a handle, printed when it
//   goes), stamps (per-workgroup timeline of score_time_iteration), band (phases of the band-layout builder), assemble (phases
//   of the host assembler), headform (which check of the QCQP rewrite declined a program).  Nothing here changes a result.
inline bool trace_on(const char* what) {
    static const std::string list = [] { const char* e = std::getenv("SCORE_TRACE"); return std::string(e ? e : ""); }();
    if (list.empty()) return false;
    const std::string w(what);
    size_t at = 0;
    while (at <= list.size()) {
        const size_t end = std::min(list.find(',', at), list.size());
        if (list.compare(at, end - at, w) == 0 || list.compare(at, end - at, "all") == 0) return true;
        at = end + 1;
    }
    return false;
}

// Setup work (equilibration, K = P + sigma I + rho A'A, chain factorisations) is split into
// contiguous index ranges over a few host threads.  Every range computes exactly what the serial
// loop would, so results do not depend on the thread count.
inline int host_threads() {
    static const int n = [] {
        unsigned h = std::thread::hardware_concurrency();
        cpu_set_t set;  // the CPUs this process may actually run on (containers, taskset)
        if (sched_getaffinity(0, sizeof(set), &set) == 0) h = std::min<unsigned>(h, (unsigned)CPU_COUNT(&set));
        // ... and the CPU time it may use: a cgroup bandwidth limit (cpu.max "quota period") stalls EVERY thread
        // of the group for the rest of the period once the quota is spent, so more runnable threads than
        // quota / period CPUs make setup slower, not faster (the MI355X boxes: 256 CPUs visible, 16 granted)
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                h = std::min<unsigned>(h, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            std::fclose(f);
        } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long long quota = 0, period = 100000;
            const bool ok = std::fscanf(g, "%lld", &quota) == 1;
            std::fclose(g);
            if (FILE* pf = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(pf, "%lld", &period) != 1) period = 100000;
                std::fclose(pf);
            }
            if (ok && quota > 0 && period > 0) h = std::min<unsigned>(h, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        // several ranks on one node (torch.distributed.run / bench.py --gpus N export LOCAL_WORLD_SIZE) share the host:
        // each takes its share of what the node grants
        if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
            const int ranks = std::atoi(e);
            if (ranks > 1) h = std::max(1u, h / (unsigned)ranks);
        }
        if (const char* e = std::getenv("SCORE_HOST_THREADS")) h = (unsigned)std::max(1, std::atoi(e));
        return (int)std::min(16u, std::max(1u, h));
    }();
    return n;
}

// ---- how host threads wait (the drivers for the device: score_hip.hip; the team workers for their next job: below) ----
struct HostWaitStats {
    std::atomic<long long> spin_ns{0}, sleep_ns{0}, waits{0}, sleeps{0};
    std::atomic<int> active_solves{0};
};
inline HostWaitStats& wait_stats() { static HostWaitStats s; return s; }
constexpr long kEconomySleepNs = 25000;
constexpr int kEconomyDepth = 6;
// Policy (SCORE_WAIT_POLICY = auto | spin | economy; default auto): economy when the waiting drivers would take more than half
// of the CPUs this rank may use (host_threads(): affinity, cgroup quota, LOCAL_WORLD_SIZE) -- measured on one MI355X box with 16
// CPUs granted, 64 fresh graphs per sweep on 4 driver threads: spinning 2 232-2 267 problems/s at 2.3-2.5 ms of CPU per problem,
// economy 2 073-2 098 at 1.36-1.53 (profiles/r06_economy_ab.txt): a lone rank with idle CPUs keeps the 7 %, eight ranks on those 16
// CPUs (two each) cannot afford 4 spinning drivers per rank and sleep.
inline int wait_policy() {  // 0 auto, 1 spin, 2 economy
    static const int v = [] {
        const char* e = std::getenv("SCORE_WAIT_POLICY");
        if (!e) return 0;
        const std::string s(e);
        return s == "spin" ? 1 : s == "economy" ? 2 : 0;
    }();
    return v;
}
inline bool economy_waits() {
    const int pol = wait_policy();
    if (pol) return pol == 2;
    return 2 * wait_stats().active_solves.load(std::memory_order_relaxed) > host_threads();
}

// The thread budget.  Handles are created concurrently (the lock-step groups of solve_score_batch, one host thread
// each; the polish structures of a handle on a thread of their own; model construction on a pool): every such
// builder opens a BuildScope for its duration, and every parallel region takes, at the moment it starts, what is
// free -- host_threads() minus the builders' own threads minus the extra threads other regions are running -- and at
// least its caller.  A builder in a serial phase leaves its share to the others; the machine's thread budget is not
// multiplied by the number of callers.  A region opened from inside a part runs serially.  Where the number of parts
// is planned ahead (parallel_parts) the region is started with that number (the `parts` argument).
inline std::atomic<int>& active_builders() {
    static std::atomic<int> n{0};
    return n;
}
inline std::atomic<int>& extra_threads_busy() {
    static std::atomic<int> n{0};
    return n;
}
inline bool& tl_in_parallel_region();
struct BuildScope {
    // (a builder running as a part of somebody's parallel region -- score_assemble_batch: one graph per part -- is
    //  already counted among that region's threads)
    bool counted;
    BuildScope() : counted(!tl_in_parallel_region()) { if (counted) active_builders().fetch_add(1, std::memory_order_acq_rel); }
    ~BuildScope() { if (counted) active_builders().fetch_sub(1, std::memory_order_acq_rel); }
    BuildScope(const BuildScope&) = delete;
    BuildScope& operator=(const BuildScope&) = delete;
};

// Teams of host threads that outlive the call: starting and joining 16 std::threads costs 0.3-0.4 ms
// on the MI355X host, and score_create alone has ~17 parallel phases.  A parallel region borrows a whole
// team for its duration (TeamLease); handles created concurrently from different host threads (the
// lock-step groups of solve_score_batch) each get a team of their own, up to kMaxTeams -- beyond that a
// region starts its own threads.  Part 0 runs on the calling thread; every part of a region runs
// concurrently (TeamBarrier relies on that).  Teams are leaked at exit on purpose and rebuilt lazily in
// a forked child (their threads do not exist there).
// Hand-over of a job: its fields go into the slot of its generation's parity before the generation counter is published
// (release); a worker that sees a new generation g (acquire) reads slot g & 1 and checks that the counter still says g --
// a worker that takes no part in a job may lag behind, but the slot of g is only rewritten for g + 2, after g + 1 has
// been published.  Workers SPIN for the next job for team_spin_us() before they sleep on the condition
// variable, and the caller spins for the last part before it sleeps: the parallel phases of one setup follow each other
// within tens of microseconds, and waking 15 sleepers through one mutex cost 0.1-0.2 ms per phase -- more than many of
// the phases themselves (a headline score_create is ~29 ms of single-thread work and took 10 ms on 16 threads).
// Round 6, under economy waits (several drivers / ranks share the CPUs): a worker spins for 20 us, and only when it TOOK PART in
// the job before -- notify_all wakes the whole team, and a region of four parts left eleven workers spinning for 120 us each:
// the team workers were half of the host CPU of a fresh-graph sweep (profiles/r06_economy_ab.txt: 1.30 -> 1.07 ms per problem).
// With CPUs to spare everybody keeps the 120 us (cold workers asleep cost a headline score_create 0.3 ms).
inline int team_spin_us(bool warm = true) { return economy_waits() ? (warm ? 20 : 0) : 120; }
struct HostTeam {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    struct Slot {
        std::atomic<const std::function<void(int)>*> job{nullptr};
        std::atomic<int> parts{0};
    } slot[2];
    std::atomic<uint64_t> gen{0};
    std::atomic<int> remaining{0};
    explicit HostTeam(int n_workers) {
        for (int w = 0; w < n_workers; ++w) workers.emplace_back([this, w] { loop(w + 1); });
    }
    static bool spin_until(const std::function<bool()>& ready, int limit) {
        if (limit <= 0) return ready();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            if (ready()) return true;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((i & 127) == 127 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > limit)
                return ready();
        }
    }
    void loop(int part) {
        uint64_t seen = 0;
        bool warm = false;  // took part in the job before: the next phase of the same setup follows within tens of microseconds
        for (;;) {
            if (!spin_until([&] { return gen.load(std::memory_order_acquire) != seen; }, team_spin_us(warm))) {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
            }
            warm = false;
            const uint64_t g = gen.load(std::memory_order_acquire);
            const std::function<void(int)>* j = slot[g & 1].job.load(std::memory_order_relaxed);
            const int parts = slot[g & 1].parts.load(std::memory_order_relaxed);
            // (a seqlock read: the slot loads above must not sink below the re-check -- an acquire LOAD alone does not order
            //  the earlier relaxed loads before it; the fence does)
            std::atomic_thread_fence(std::memory_order_acquire);
            if (gen.load(std::memory_order_relaxed) != g) continue;  // (published again meanwhile: read it afresh)
            seen = g;
            if (part >= parts) continue;
            (*j)(part);
            warm = true;
            if (remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m);
                cv_done.notify_one();
            }
        }
    }
    void run(int parts, const std::function<void(int)>& f) {  // caller holds the lease; parts - 1 <= workers.size()
        const uint64_t g = gen.load(std::memory_order_relaxed) + 1;  // (one caller at a time: the lease)
        slot[g & 1].job.store(&f, std::memory_order_relaxed);
        slot[g & 1].parts.store(parts, std::memory_order_relaxed);
        remaining.store(parts - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m);  // (sleepers check the generation under this mutex)
            gen.store(g, std::memory_order_release);
        }
        cv_work.notify_all();
        f(0);
        if (!spin_until([&] { return remaining.load(std::memory_order_acquire) == 0; }, team_spin_us())) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return remaining.load(std::memory_order_acquire) == 0; });
        }
    }
};
constexpr int kMaxTeams = 8;
struct TeamPool {
    std::mutex m;
    std::vector<HostTeam*> idle;
    int made = 0;
};
inline std::atomic<TeamPool*>& team_pool_slot() {
    static std::atomic<TeamPool*> slot{nullptr};
    return slot;
}
inline std::mutex& team_pool_make_mutex() {
    static std::mutex make;
    return make;
}
inline TeamPool* team_pool() {
    std::mutex& make = team_pool_make_mutex();
    static std::once_flag fork_hook;
    TeamPool* t = team_pool_slot().load(std::memory_order_acquire);
    if (t) return t;
    std::lock_guard<std::mutex> lk(make);
    t = team_pool_slot().load(std::memory_order_acquire);
    if (!t) {
        // a forked child has none of the parent's threads: no team, nobody inside a build or a parallel region
        std::call_once(fork_hook, [] {
            // (the creation mutex is taken across the fork, so the child never inherits it locked by a thread it does not have)
            pthread_atfork([] { team_pool_make_mutex().lock(); }, [] { team_pool_make_mutex().unlock(); }, [] {
                team_pool_slot().store(nullptr);
                active_builders().store(0);
                extra_threads_busy().store(0);
                team_pool_make_mutex().unlock();
            });
        });
        t = new TeamPool();
        team_pool_slot().store(t, std::memory_order_release);
    }
    return t;
}
// a team for the duration of one parallel region (team == nullptr: none free, the region starts threads)
struct TeamLease {
    TeamPool* pool = nullptr;
    HostTeam* team = nullptr;
    TeamLease() {
        if (host_threads() <= 1) return;
        pool = team_pool();
        std::lock_guard<std::mutex> lk(pool->m);
        if (!pool->idle.empty()) { team = pool->idle.back(); pool->idle.pop_back(); }
        else if (pool->made < kMaxTeams) { ++pool->made; team = new HostTeam(host_threads() - 1); }
    }
    ~TeamLease() {
        if (!team) return;
        std::lock_guard<std::mutex> lk(pool->m);
        pool->idle.push_back(team);
    }
    TeamLease(const TeamLease&) = delete;
    TeamLease& operator=(const TeamLease&) = delete;
};

inline bool& tl_in_parallel_region() {
    static thread_local bool v = false;
    return v;
}
inline int region_width() {
    if (tl_in_parallel_region()) return 1;
    const int callers = std::max(1, active_builders().load(std::memory_order_acquire));
    const int free_extra = host_threads() - callers - extra_threads_busy().load(std::memory_order_acquire);
    int width = 1 + std::max(0, free_extra);
    // many builders at once (the lock-step groups of a Monte-Carlo sweep start together): nobody takes more than its share --
    // the first to open a region used to take every free thread and left the others serial (8 groups' model construction:
    // 17-22 ms per group, the last one done 24-28 ms into the sweep)
    if (callers >= 4) width = std::min(width, std::max(1, (host_threads() + callers - 1) / callers));
    return width;
}

// what the surviving members of a TeamBarrier region throw when another member failed (never the root cause)
struct TeamAborted : std::runtime_error {
    TeamAborted() : std::runtime_error("parallel setup aborted: another part failed") {}
};

// fn(part, begin, end) for part = 0..T-1 over the boundaries `bound` (T + 1 entries), all parts concurrently
template <class F>
inline void run_parts(const std::vector<int64_t>& bound, F&& fn) {
    const int T = (int)bound.size() - 1;
    if (T <= 1) { fn(0, bound[0], bound[(size_t)T]); return; }
    std::vector<std::exception_ptr> err((size_t)T);
    bool& in_region = tl_in_parallel_region();  // a part that opens a region of its own never touches the gate
    const bool nested = in_region;
    const std::function<void(int)> body = [&](int t) {
        bool& mine = tl_in_parallel_region();  // (this thread's flag: parts run on other threads)
        const bool was = mine;
        mine = true;
        try { fn(t, bound[(size_t)t], bound[(size_t)t + 1]); } catch (...) { err[(size_t)t] = std::current_exception(); }
        mine = was;
    };
    struct Busy {  // the extra threads of this region, for the budget of the others
        int n;
        explicit Busy(int k) : n(k) { extra_threads_busy().fetch_add(n, std::memory_order_acq_rel); }
        ~Busy() { extra_threads_busy().fetch_sub(n, std::memory_order_acq_rel); }
    } busy(T - 1);
    bool ran = false;
    if (!nested) {
        TeamLease lease;
        if (lease.team && T - 1 <= (int)lease.team->workers.size()) {
            lease.team->run(T, body);
            ran = true;
        }
    }
    if (!ran) {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back([&body, t] { body(t); });
        body(0);
        for (auto& x : th) x.join();
    }
    // the first part that failed for a reason of its own; the "aborted: another part failed" exceptions the surviving parts
    // of a TeamBarrier region throw (TeamAborted) are secondary and only reported when nothing else is
    std::exception_ptr secondary;
    for (auto& e : err) {
        if (!e) continue;
        try { std::rethrow_exception(e); }
        catch (const TeamAborted&) { if (!secondary) secondary = e; }
        catch (...) { std::rethrow_exception(e); }
    }
    if (secondary) std::rethrow_exception(secondary);
}

template <class F>
inline void parallel_ranges(int64_t n, int64_t min_per_thread, F&& fn, int parts = 0) {  // fn(part, begin, end)
    const int T = parts > 0 ? parts : (int)std::min<int64_t>(region_width(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (T <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<int64_t> bound((size_t)T + 1);
    for (int t = 0; t <= T; ++t) bound[(size_t)t] = n * t / T;
    run_parts(bound, fn);
}
// The same with part boundaries chosen so that every part carries about the same total weight
// (weight(i) >= 0: the work of index i).  Parts stay contiguous and ordered; a few very heavy indices
// (a landmark seen by thousands of ranges) no longer land in one part.  fn(part, begin, end).
template <class W, class F>
inline void parallel_ranges_balanced(int64_t n, int64_t min_per_thread, W&& weight, F&& fn, int parts = 0) {
    const int T = parts > 0 ? parts : (int)std::min<int64_t>(region_width(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
    if (T <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<double> pre((size_t)n + 1, 0.0);
    for (int64_t i = 0; i < n; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + 1.0 + (double)weight(i);
    std::vector<int64_t> bound((size_t)T + 1, n);
    bound[0] = 0;
    for (int t = 1; t < T; ++t) {
        const double target = pre[(size_t)n] * t / T;
        bound[(size_t)t] = std::max<int64_t>(bound[(size_t)t - 1],
                                             std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
        bound[(size_t)t] = std::min<int64_t>(bound[(size_t)t], n);
    }
    run_parts(bound, fn);
}
inline int parallel_parts(int64_t n, int64_t min_per_thread) {
    return (int)std::min<int64_t>(region_width(), std::max<int64_t>(1, n / std::max<int64_t>(1, min_per_thread)));
}

// Sense-reversing barrier for a team started by one parallel_ranges call: lets a thread keep
// "its" rows in its own cache across the passes of an iterative setup step.
struct TeamBarrier {
    explicit TeamBarrier(int n) : T(n) {}
    // a member that fails calls abort() instead of arriving: the others leave their wait with an exception
    void abort() { failed.store(true, std::memory_order_release); }
    void wait() {
        if (T <= 1) return;
        if (failed.load(std::memory_order_acquire)) throw TeamAborted();
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == T - 1) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g) {
                if (failed.load(std::memory_order_acquire)) throw TeamAborted();
                if (++spins > 4096) std::this_thread::yield();
            }
        }
    }
    int T;
    std::atomic<int> count{0}, gen{0};
    std::atomic<bool> failed{false};
};

struct Csr {
    int64_t nrows = 0, ncols = 0;
    std::vector<int32_t> ptr, col;
    std::vector<double> val;
    int64_t nnz() const { return (int64_t)col.size(); }
};

struct RowBlocks {
    std::vector<int32_t> first_row;  // nb (+ 1: the end of the last block)
    std::vector<int32_t> end_row;    // nb: blocks of one problem need not be adjacent (replicated problems skip rows)
    std::vector<int32_t> prob;       // nb
    std::vector<int32_t> rs;         // nb: replica stride of a block of replicated rows (see HostSystem::rep), 0 = plain rows
    std::vector<int32_t> part_ptr;   // count + 1 : block range of each problem
    // segments of split long rows (kLongSeg): per block its nonzero range [kbeg, kend) (-1: the block's whole rows) and
    // {first block of the row, segments | segment << 16, first partial-sum slot, long-row ordinal} (segments == 0: none)
    std::vector<int32_t> kbeg, kend, lfirst, lseg, lbase, lid;
    int n_long = 0, n_long_slots = 0;
    int nb() const { return (int)prob.size(); }
};
// A contiguous range of matrix rows to be tiled: plain rows (rs == 0) or rows of replica 0 whose results
// and operands repeat every rs entries for the other replicas.
struct RowSegment { int64_t begin, end; int32_t prob, rs; };

// One level of a chain's nested-dissection factorisation.  The factor blocks are
// stored structure-of-arrays so that the lanes of a wavefront (= consecutive
// runs, separators or nodes) read consecutive doubles:
//   R (run phase)        Lf, Dinv : fac[offR + ((slot*b2 + e)*P + q)*nruns + j]
//                                   slot 0/1, block entry e, position q in run j
//   S (separator phase)  Cl, Cr   : fac[offS + (slot*b2 + e)*nsep + j]
//   B (back-substitution) V, W    : fac[offB + (slot*b2 + e)*N + i]
struct ChainLevelDesc {
    int32_t N;        // nodes on this level
    int32_t p;        // radix (0 = last level: one sequential run)
    int32_t nruns;    // runs on this level (last level: 1)
    int32_t P;        // node positions per run (p - 1; last level: N)
    int32_t nsep;     // separators (N / p; last level: 0)
    int32_t vec_off;  // first node of this level in the chain's scratch (level >= 1)
    // bank-conflict-free vector layout of k_prec_pre: component c of node i of this level lives at
    // lds_off + i * bs + i / p + c (one padding double per run + separator), all levels back to back
    int32_t lds_off;
    uint32_t inv_p;   // 2^20 / p + 1: i / p == (i * inv_p) >> 20 for i < 2^18
    int64_t offR, offS, offB;  // offsets into `fac`, in doubles
    // register-resident coarse levels (k_prec_pre<.., REGDEEP>): the runs / nodes of level >= 1 are served by the staging
    // lanes lane0 .. (disjoint ranges per level, chain_lane_plan); -1: the chain does not fit that kernel
    int32_t lane0;
    int32_t pad_;
};

struct ChainDesc {
    int32_t level_begin, n_levels;
    int32_t prob;
    int32_t node_begin;  // first level-0 node in node_col
    int32_t N;           // level-0 nodes
    int32_t scratch_off; // first scratch node (global scratch, units of nodes)
    int32_t scratch_nodes;
    int32_t col0;        // first column of node 0
    int32_t col_stride;  // node_col[i] == col0 + i * col_stride for every node (0 = irregular)
    // register-resident coarse levels: slot map of this chain's length class in HostSystem::deep_map (-1: not eligible)
    // and the chain's block in the lane-major factor copy (floats; a replica's chain points at its owner's block)
    int32_t deep_map_off;
    int32_t deep_off;
};

// Work item of the preconditioner kernel: a chain, or a block of Jacobi columns.
struct PrecWork {
    int32_t kind;   // 0 = chain, 1 = jacobi block
    int32_t index;  // chain id, or first entry in diag_cols
    int32_t count;  // jacobi: entries in this block
    int32_t prob;
};

inline void mat_mul(const double* A, const double* B, double* C, int bs) {
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
            double s = 0;
            for (int k = 0; k < bs; ++k) s += A[i * bs + k] * B[k * bs + j];
            C[i * bs + j] = s;
        }
}
inline void mat_mul_bt(const double* A, const double* B, double* C, int bs) {  // A * B'
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
            double s = 0;
            for (int k = 0; k < bs; ++k) s += A[i * bs + k] * B[j * bs + k];
            C[i * bs + j] = s;
        }
}
inline void mat_mul_at(const double* A, const double* B, double* C, int bs) {  // A' * B
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
            double s = 0;
            for (int k = 0; k < bs; ++k) s += A[k * bs + i] * B[k * bs + j];
            C[i * bs + j] = s;
        }
}
inline void mat_t(const double* A, double* C, int bs) {
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) C[i * bs + j] = A[j * bs + i];
}
inline bool mat_inv(const double* A, double* Ainv, int bs) {  // Gauss-Jordan, partial pivoting
    double M[kMaxBs][2 * kMaxBs];
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
            M[i][j] = A[i * bs + j];
            M[i][bs + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < bs; ++c) {
        int piv = c;
        for (int r = c + 1; r < bs; ++r)
            if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
        if (!(std::fabs(M[piv][c]) > 0.0)) return false;
        if (piv != c)
            for (int j = 0; j < 2 * bs; ++j) std::swap(M[c][j], M[piv][j]);
        double inv = 1.0 / M[c][c];
        for (int j = 0; j < 2 * bs; ++j) M[c][j] *= inv;
        for (int r = 0; r < bs; ++r) {
            if (r == c) continue;
            double f = M[r][c];
            if (f != 0.0)
                for (int j = 0; j < 2 * bs; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) Ainv[i * bs + j] = M[i][bs + j];
    return true;
}

// ---------------------------------------------------------------------------
// multi-level factorisation of one SPD block-tridiagonal chain
//   diagonal blocks Ad[i], sub-diagonal blocks Bs[i] = T[i, i-1] (Bs[0] unused)
// Node record (4 blocks of bs*bs): interior node  [Lf, Dinv, V, W]
//                                  separator node [Cl, Cr, -, -]
// ---------------------------------------------------------------------------
inline void factor_chain_levels(int bs, int radix, int N0, const std::vector<double>& Ad,
                                const std::vector<double>& Bs, std::vector<ChainLevelDesc>& levels,
                                std::vector<double>& fac, int& scratch_nodes) {
    const int b2 = bs * bs;
    std::vector<double> curA = Ad, curB = Bs;
    int N = N0, vec_off = 0, lds_off = 0;
    double D[16], T1[16], T2[16];
    std::vector<double> aos;  // node records [Lf|Cl, Dinv|Cr, V, W] of the current level
    for (int lvl = 0;; ++lvl) {
        ChainLevelDesc L{};
        L.N = N;
        L.vec_off = (lvl == 0) ? -1 : vec_off;
        if (lvl > 0) vec_off += N;
        aos.assign((size_t)N * 4 * b2, 0.0);
        auto nd = [&](int i, int slot) { return &aos[((size_t)i * 4 + slot) * b2]; };
        const bool last = (N <= radix - 1);  // the closing run must fit a lane's register tile (radix - 1 nodes)
        L.p = last ? 0 : radix;
        L.inv_p = last ? 0u : ((1u << 20) / (uint32_t)radix + 1u);
        L.lds_off = lds_off;
        lds_off += N * bs + (last ? 0 : N / radix) + 1;
        const int nsep = last ? 0 : N / radix;
        L.nsep = nsep;
        L.nruns = nsep + 1;
        L.P = last ? N : radix - 1;
        std::vector<double> nA((size_t)nsep * b2, 0.0), nB((size_t)nsep * b2, 0.0);
        for (int j = 0; j <= nsep; ++j) {
            const int lo = last ? 0 : j * radix;
            const int hi = last ? N : std::min(j * radix + radix - 1, N);
            if (lo >= hi) continue;
            for (int i = lo; i < hi; ++i) {
                if (i == lo) {
                    std::memcpy(D, &curA[(size_t)i * b2], sizeof(double) * b2);
                } else {
                    mat_mul(&curB[(size_t)i * b2], nd(i - 1, 1), nd(i, 0), bs);    // Lf = B Dinv_prev
                    mat_mul_bt(nd(i, 0), &curB[(size_t)i * b2], T1, bs);           // Lf B'
                    for (int k = 0; k < b2; ++k) D[k] = curA[(size_t)i * b2 + k] - T1[k];
                }
                if (!mat_inv(D, nd(i, 1), bs))
                    throw std::runtime_error("chain preconditioner: singular diagonal block");
            }
            if (last) continue;
            const bool hasL = (j >= 1), hasR = (j < nsep);
            for (int side = 0; side < 2; ++side) {
                if ((side == 0 && !hasL) || (side == 1 && !hasR)) continue;
                const int slot = 2 + side;
                // right-hand side: block at lo (V) = B[lo]; block at hi-1 (W) = B[hi]'
                // forward: a_i = rhs_i - Lf_i a_{i-1}
                for (int i = lo; i < hi; ++i) {
                    double* a = nd(i, slot);
                    for (int k = 0; k < b2; ++k) a[k] = 0.0;
                    if (side == 0 && i == lo) std::memcpy(a, &curB[(size_t)lo * b2], sizeof(double) * b2);
                    if (side == 1 && i == hi - 1) mat_t(&curB[(size_t)hi * b2], a, bs);
                    if (i > lo) {
                        mat_mul(nd(i, 0), nd(i - 1, slot), T1, bs);
                        for (int k = 0; k < b2; ++k) a[k] -= T1[k];
                    }
                }
                // diagonal + backward: y_i = Dinv_i a_i - Lf_{i+1}' y_{i+1}
                for (int i = hi - 1; i >= lo; --i) {
                    double* a = nd(i, slot);
                    mat_mul(nd(i, 1), a, T1, bs);
                    if (i + 1 < hi) {
                        mat_mul_at(nd(i + 1, 0), nd(i + 1, slot), T2, bs);
                        for (int k = 0; k < b2; ++k) T1[k] -= T2[k];
                    }
                    std::memcpy(a, T1, sizeof(double) * b2);
                }
            }
        }
        for (int j = 0; j < nsep; ++j) {
            const int s = j * radix + radix - 1;
            std::memcpy(nd(s, 0), &curB[(size_t)s * b2], sizeof(double) * b2);  // Cl = T[s, s-1]
            double* S = &nA[(size_t)j * b2];
            std::memcpy(S, &curA[(size_t)s * b2], sizeof(double) * b2);
            mat_mul(nd(s, 0), nd(s - 1, 3), T1, bs);  // Cl * W_{s-1}
            for (int k = 0; k < b2; ++k) S[k] -= T1[k];
            if (s + 1 < N) {
                mat_t(&curB[(size_t)(s + 1) * b2], nd(s, 1), bs);  // Cr = T[s, s+1] = B[s+1]'
                mat_mul(nd(s, 1), nd(s + 1, 2), T1, bs);           // Cr * V_{s+1}
                for (int k = 0; k < b2; ++k) S[k] -= T1[k];
            }
            if (j >= 1) {
                mat_mul(nd(s, 0), nd(s - 1, 2), T1, bs);  // Cl * V_{s-1}
                for (int k = 0; k < b2; ++k) nB[(size_t)j * b2 + k] = -T1[k];
            }
        }
        // pack the node records into the lane-coalesced layout
        L.offR = (int64_t)fac.size();
        fac.resize(fac.size() + (size_t)2 * b2 * L.P * L.nruns, 0.0);
        L.offS = (int64_t)fac.size();
        fac.resize(fac.size() + (size_t)2 * b2 * nsep, 0.0);
        L.offB = (int64_t)fac.size();
        fac.resize(fac.size() + (size_t)2 * b2 * N, 0.0);
        for (int j = 0; j < L.nruns; ++j) {
            const int lo = last ? 0 : j * radix;
            const int hi = last ? N : std::min(j * radix + radix - 1, N);
            for (int i = lo; i < hi; ++i)
                for (int slot = 0; slot < 2; ++slot)
                    for (int e = 0; e < b2; ++e)
                        fac[L.offR + ((size_t)(slot * b2 + e) * L.P + (i - lo)) * L.nruns + j] = nd(i, slot)[e];
        }
        for (int j = 0; j < nsep; ++j) {
            const int s = j * radix + radix - 1;
            for (int slot = 0; slot < 2; ++slot)
                for (int e = 0; e < b2; ++e) fac[L.offS + (size_t)(slot * b2 + e) * nsep + j] = nd(s, slot)[e];
        }
        if (!last)
            for (int i = 0; i < N; ++i)
                for (int slot = 0; slot < 2; ++slot)
                    for (int e = 0; e < b2; ++e) fac[L.offB + (size_t)(slot * b2 + e) * N + i] = nd(i, 2 + slot)[e];
        levels.push_back(L);
        if (last) break;
        curA.swap(nA);
        curB.swap(nB);
        N = nsep;
    }
    scratch_nodes = vec_off;
}

// The level structure of factor_chain_levels() for a chain of N0 nodes (it depends on nothing else):
// descriptors with offsets relative to the chain's first factor double, total doubles, scratch nodes.
inline void chain_level_layout(int bs, int radix, int N0, std::vector<ChainLevelDesc>& levels, size_t& fac_size,
                               int& scratch_nodes) {
    const int b2 = bs * bs;
    int N = N0, vec_off = 0, lds_off = 0;
    size_t off = 0;
    levels.clear();
    for (int lvl = 0;; ++lvl) {
        ChainLevelDesc L{};
        L.N = N;
        L.vec_off = (lvl == 0) ? -1 : vec_off;
        if (lvl > 0) vec_off += N;
        const bool last = (N <= radix - 1);
        L.p = last ? 0 : radix;
        L.inv_p = last ? 0u : ((1u << 20) / (uint32_t)radix + 1u);
        L.lds_off = lds_off;
        lds_off += N * bs + (last ? 0 : N / radix) + 1;
        const int nsep = last ? 0 : N / radix;
        L.nsep = nsep;
        L.nruns = nsep + 1;
        L.P = last ? N : radix - 1;
        L.offR = (int64_t)off; off += (size_t)2 * b2 * L.P * L.nruns;
        L.offS = (int64_t)off; off += (size_t)2 * b2 * nsep;
        L.offB = (int64_t)off; off += (size_t)2 * b2 * N;
        levels.push_back(L);
        if (last) break;
        N = nsep;
    }
    fac_size = off;
    scratch_nodes = vec_off;
}

// Lane plan of the register-resident coarse levels: level 1 on staging lanes 0 .. N_1 - 1 (its runs on the first
// nruns_1 of them), every further level on a 64-aligned range of its own behind the previous one.  Returns false
// (and leaves lane0 = -1) when the levels do not fit `lanes` staging lanes.
inline bool chain_lane_plan(std::vector<ChainLevelDesc>& lv, int lanes) {
    for (auto& L : lv) { L.lane0 = -1; L.pad_ = 0; }
    if (lv.size() < 2) return true;  // a single level: nothing coarse
    // level 1: node i (spike blocks, slots [0, 2 b2)) on lane i, run j on lane j.  Levels >= 2 keep their run and
    // spike blocks in OTHER slots, so they share lanes with level 1's nodes -- but not with level 1's runs nor with
    // each other: 64-aligned ranges behind the lanes of level 1's runs.
    bool ok = lv[1].N <= lanes && lv[1].nruns <= lanes;
    lv[1].lane0 = 0;
    int next = (lv[1].nruns + 63) / 64 * 64;
    for (size_t l = 2; l < lv.size() && ok; ++l) {
        const int need = std::max(lv[l].N, lv[l].nruns);
        if (next + need > lanes) { ok = false; break; }
        lv[l].lane0 = next;
        next = (next + need + 63) / 64 * 64;
    }
    if (!ok) {
        for (auto& L : lv) L.lane0 = -1;
        return ok;
    }
    // which slot groups each 64-lane wavefront of the staging lanes has to load (k_prec_pre reads lv[0].pad_ instead of
    // walking the level table): bit w = level-1 spike blocks, bit 4 + w = run / separator blocks, bit 8 + w = the spike
    // blocks of a level >= 2, for wavefront w (lanes 64 w .. 64 w + 63)
    int mask = 0;
    for (int w = 0; w * 64 < lanes && w < 4; ++w) {
        const int w0 = 64 * w;
        if (lv[1].p != 0 && w0 < lv[1].N) mask |= 1 << w;
        for (size_t l = 1; l < lv.size(); ++l) {
            if (lv[l].lane0 < w0 + 64 && lv[l].lane0 + lv[l].nruns > w0) mask |= 1 << (4 + w);
            if (l >= 2 && lv[l].p != 0 && lv[l].lane0 < w0 + 64 && lv[l].lane0 + lv[l].N > w0) mask |= 1 << (8 + w);
        }
    }
    lv[0].pad_ = mask;
    return ok;
}
// Layout of the lane-major copy (k_deep_pack -> k_prec_pre<.., REGDEEP>): the three slot groups of a lane -- [0, 2 b2)
// level-1 spikes, [2 b2, 10 b2) run + separator, [10 b2, 12 b2) own-level spikes -- each padded to a multiple of 4 and
// stored as 16-byte packets, packet q of lane dt at float 4 (q * lanes + dt): a lane fetches a group with a few
// 16-byte loads (27 + 1 padding packet per lane at 3 x 3 blocks) instead of one 4-byte load per value.
inline int deep_group_pad(int bs) { return ((2 * bs * bs + 3) & ~3) - 2 * bs * bs; }
inline int deep_padded_slot(int bs, int slot) {
    const int b2 = bs * bs, pad = deep_group_pad(bs);
    return slot < 2 * b2 ? slot : (slot < 10 * b2 ? slot + pad : slot + 2 * pad);
}
inline int deep_padded_slots(int bs) { return 12 * bs * bs + 2 * deep_group_pad(bs); }
// Slot map of the lane-major copy of a chain's coarse-level factors (k_deep_pack -> k_prec_pre<.., REGDEEP>): staging lane
// dt keeps in registers  [0, 2 b2): the spike blocks V, W of level-1 node dt;  [2 b2, 10 b2): the run (6 b2) and
// separator (2 b2) blocks of the run it serves (whatever level);  [10 b2, 12 b2): the spike blocks of its own level's
// node when that level is >= 2.  map[s * lanes + dt] = index into the chain's factor array (relative), -1 = unused.
inline void chain_deep_map(int bs, const std::vector<ChainLevelDesc>& lv, int64_t fac_base, int lanes, std::vector<int32_t>& map) {
    const int b2 = bs * bs, RMAX = 3;
    map.assign((size_t)12 * b2 * lanes, -1);
    auto put = [&](int slot, int dt, int64_t idx) { map[(size_t)slot * lanes + dt] = (int32_t)(idx - fac_base); };
    for (size_t l = 1; l < lv.size(); ++l) {
        const ChainLevelDesc& L = lv[l];
        const bool last = (L.p == 0);
        for (int j = 0; j < L.nruns; ++j) {
            const int lo = last ? 0 : j * L.p;
            const int hi = last ? L.N : std::min(j * L.p + L.p - 1, L.N);
            const int len = std::max(hi - lo, 1);
            for (int q = 0; q < RMAX; ++q)
                for (int slot = 0; slot < 2; ++slot)
                    for (int e = 0; e < b2; ++e)
                        put(2 * b2 + (slot * RMAX + q) * b2 + e, L.lane0 + j,
                            L.offR + ((int64_t)(slot * b2 + e) * L.P + std::min(q, len - 1)) * L.nruns + j);
        }
        for (int j = 0; j < L.nsep; ++j)
            for (int slot = 0; slot < 2; ++slot)
                for (int e = 0; e < b2; ++e) put(2 * b2 + (2 * RMAX + slot) * b2 + e, L.lane0 + j, L.offS + (int64_t)(slot * b2 + e) * L.nsep + j);
        if (!last)
            for (int i = 0; i < L.N; ++i)
                for (int e = 0; e < 2 * b2; ++e) put((l == 1 ? 0 : 10 * b2) + e, L.lane0 + i, L.offB + (int64_t)e * L.N + i);
    }
}

// Reference (host) application of the factorisation: z = M^{-1} r for one chain.
// r/z are indexed by column (level 0 gathers through node_col); `scr` holds the
// level >= 1 vectors.  The HIP kernel performs exactly these operations.
inline void chain_solve_host(const ChainDesc& ch, const ChainLevelDesc* levels, const double* fac,
                             const int32_t* node_col, int bs, const double* r, double* z, double* scr) {
    const int b2 = bs * bs;
    const int32_t* nc = node_col + ch.node_begin;
    for (int lvl = 0; lvl < ch.n_levels; ++lvl) {
        const ChainLevelDesc& L = levels[ch.level_begin + lvl];
        const bool last = (L.p == 0);
        const int nsep = L.nsep;
        auto Rb = [&](int slot, int e, int q, int j) { return fac[L.offR + ((size_t)(slot * b2 + e) * L.P + q) * L.nruns + j]; };
        auto Sb = [&](int slot, int e, int j) { return fac[L.offS + (size_t)(slot * b2 + e) * nsep + j]; };
        auto in = [&](int i, int c) { return lvl == 0 ? r[nc[i] + c] : scr[(size_t)(L.vec_off + i) * bs + c]; };
        auto out = [&](int i, int c) -> double& {
            return lvl == 0 ? z[nc[i] + c] : scr[(size_t)(L.vec_off + i) * bs + c];
        };
        for (int j = 0; j < L.nruns; ++j) {
            const int lo = last ? 0 : j * L.p;
            const int hi = last ? L.N : std::min(j * L.p + L.p - 1, L.N);
            if (lo >= hi) continue;
            double prev[kMaxBs], cur[kMaxBs];
            for (int i = lo; i < hi; ++i) {  // forward
                for (int c = 0; c < bs; ++c) cur[c] = in(i, c);
                if (i > lo)
                    for (int c = 0; c < bs; ++c)
                        for (int k = 0; k < bs; ++k) cur[c] -= Rb(0, c * bs + k, i - lo, j) * prev[k];
                for (int c = 0; c < bs; ++c) { out(i, c) = cur[c]; prev[c] = cur[c]; }
            }
            for (int i = hi - 1; i >= lo; --i) {  // diagonal + backward
                double a[kMaxBs];
                for (int c = 0; c < bs; ++c) a[c] = out(i, c);
                for (int c = 0; c < bs; ++c) {
                    double s = 0;
                    for (int k = 0; k < bs; ++k) s += Rb(1, c * bs + k, i - lo, j) * a[k];
                    cur[c] = s;
                }
                if (i + 1 < hi)
                    for (int c = 0; c < bs; ++c)
                        for (int k = 0; k < bs; ++k) cur[c] -= Rb(0, k * bs + c, i + 1 - lo, j) * prev[k];
                for (int c = 0; c < bs; ++c) { out(i, c) = cur[c]; prev[c] = cur[c]; }
            }
        }
        if (last) break;
        const ChainLevelDesc& Ln = levels[ch.level_begin + lvl + 1];
        for (int j = 0; j < nsep; ++j) {
            const int s = j * L.p + L.p - 1;
            for (int c = 0; c < bs; ++c) {
                double v = in(s, c);
                for (int k = 0; k < bs; ++k) v -= Sb(0, c * bs + k, j) * out(s - 1, k);
                if (s + 1 < L.N)
                    for (int k = 0; k < bs; ++k) v -= Sb(1, c * bs + k, j) * out(s + 1, k);
                scr[(size_t)(Ln.vec_off + j) * bs + c] = v;
            }
        }
    }
    for (int lvl = ch.n_levels - 2; lvl >= 0; --lvl) {
        const ChainLevelDesc& L = levels[ch.level_begin + lvl];
        const ChainLevelDesc& Ln = levels[ch.level_begin + lvl + 1];
        const int nsep = L.nsep;
        auto Bb = [&](int slot, int e, int i) { return fac[L.offB + (size_t)(slot * b2 + e) * L.N + i]; };
        auto out = [&](int i, int c) -> double& {
            return lvl == 0 ? z[nc[i] + c] : scr[(size_t)(L.vec_off + i) * bs + c];
        };
        for (int i = 0; i < L.N; ++i) {
            const int j = i / L.p;
            if (i % L.p == L.p - 1 && j < nsep) {
                for (int c = 0; c < bs; ++c) out(i, c) = scr[(size_t)(Ln.vec_off + j) * bs + c];
                continue;
            }
            for (int c = 0; c < bs; ++c) {
                double v = out(i, c);
                if (j >= 1)
                    for (int k = 0; k < bs; ++k) v -= Bb(0, c * bs + k, i) * scr[(size_t)(Ln.vec_off + j - 1) * bs + k];
                if (j < nsep)
                    for (int k = 0; k < bs; ++k) v -= Bb(1, c * bs + k, i) * scr[(size_t)(Ln.vec_off + j) * bs + k];
                out(i, c) = v;
            }
        }
    }
}

// Chains of more than kSegMaxNodes nodes (device backends): cut into segments of at most that many nodes -- ordinary chains
// for the chain kernel -- with ONE separator node between neighbours.  The separators are Jacobi columns whose inverse
// diagonal is held at zero; after every application of the chain kernel a second kernel (score_join.hpp) solves the
// separators' Schur system (block tridiagonal, n_seg - 1 nodes) and corrects the segments with their spikes: together the
// exact solve with the whole chain's block-tridiagonal matrix, as the streaming kernel computes it.
constexpr int kSegMaxNodes = 1023;
constexpr int kJoinMaxSepsHost = 128;  // == score_join.hpp kJoinMaxSeps (static_assert there): separators of one long chain
constexpr int seg_max_nodes() { return kSegMaxNodes; }  // (shorter segments: slower everywhere, profiles/r05_seg_nodes_ab.txt)
struct JoinChain {
    int32_t prob, n_seg;
    int32_t first_chain;  // H.chains index of segment 0 (the segments follow)
    int32_t sep_begin;    // first separator in join_sep_col (n_seg - 1 of them)
    int32_t owner;        // join chain whose matrix blocks this one's are (replicated problems; itself otherwise)
    int32_t pad_[3];
};
struct JoinItem {
    int32_t chain;  // segment (H.chains index)
    int32_t jc;     // its join chain
    int32_t seg;    // position in the join chain
    int32_t work;   // the segment's work item in prec_work (its slot of the r'z partial sums)
};

// ---------------------------------------------------------------------------
// The assembled batch
// ---------------------------------------------------------------------------
struct HostSystem {
    int count = 0;
    int64_t n_tot = 0, m_tot = 0;
    std::vector<int64_t> xoff, roff;  // count + 1
    std::vector<double> c0;
    double sigma = 1e-6;
    int bs = 0, radix = 4;

    // scaling (x = D xhat, s = shat / E, y = E yhat)
    std::vector<double> D, E, q, b;        // q, b scaled
    std::vector<double> qnorm_u, bnorm_u;  // per problem |q|_inf, |b|_inf (unscaled)
    std::vector<double> qnorm_s, bnorm_s;  // scaled

    // cones ("type 0" = zero-cone row, "type 1" = second-order cone)
    std::vector<int32_t> cone_row, cone_dim, cone_type;
    std::vector<int32_t> cone_block_first, cone_block_prob, cone_part_ptr;

    Csr A;  // scaled, global indices
    Csr K;
    std::vector<double> K0, K1;
    Csr G1;
    Csr G2;
    std::vector<int32_t> g2_split;  // per row: first entry of the A' part
    RowBlocks rbK, rbG1, rbG2;

    // preconditioner
    std::vector<int32_t> node_col;            // level-0 node -> first (global) column
    std::vector<int32_t> pos_diag, pos_sub;   // K.val positions of the block entries (-1 = 0)
    // what a backend that looks the positions up itself needs instead (factor_on_host == false: pos_diag, pos_sub and
    // diag_kpos stay empty): per chain node the column of its chain predecessor (-1: first node, -2: a node whose blocks
    // are its owner's -- replicated chains), per Jacobi column the row of K that holds its diagonal
    std::vector<int32_t> node_prev_owned, diag_row0;
    bool rep_exact = false;  // single replicated problem whose replicas' P values are bit-equal to replica 0's (check_replication)
    std::vector<ChainDesc> chains;
    std::vector<ChainLevelDesc> levels;
    std::vector<double> fac;
    size_t fac_doubles = 0;  // size of the factor storage (== fac.size() when the host holds it)
    int64_t scratch_nodes = 0;
    int max_chain_scratch = 0;
    std::vector<int32_t> diag_cols, diag_kpos;
    std::vector<double> dinv;  // per entry of diag_cols
    std::vector<PrecWork> prec_work;
    std::vector<int32_t> prec_part_ptr;  // count + 1
    // segmented long chains (kSegMaxNodes; empty when there are none)
    std::vector<JoinChain> join_chains;
    std::vector<JoinItem> join_items;
    std::vector<int32_t> join_sep_col, join_sep_diag;  // per separator: its first column, its entry in diag_cols / dinv

    std::vector<double> rho;       // per problem
    mutable std::vector<double> kkt_bytes; // per problem: algorithmic bytes of one K-apply (a backend that streams another layout of K restates it)

    std::vector<int64_t> fac_off;  // per chain: offset of its records in `fac` (doubles)
    std::vector<int64_t> fac_range, fac_range_H;  // per chain: [begin, end) of its factors in `fac` / in the Newton set

    // ---- row replication (score_problem::rep_d / rep_n, checked by check_replication) ----
    // rep > 1: every problem of the batch is  K = I_rep (x) K_row (+ tail).  Then K and G1 = A' hold the rows of
    // replica 0 and of the tail only (a row's index is still its first column's address: the rows of the other
    // replicas are empty), the SpMV kernels apply a replica-0 row to all rep right-hand sides (operands and results
    // repeat every rep_n[p] entries), and a chain of replica k uses the factors of its replica-0 sibling
    // (chain_owner): `levels` / `fac` hold one set per robot.  The Newton matrix of the polish is NOT of that form
    // (active cones couple the rows): chainsH / levelsH / fac_doubles_H describe the same chains with factors of
    // their own.  rep == 1: chainsH == chains, levelsH == levels.
    // lane-major copies of the coarse-level factors (chain_deep_map): slot maps per chain length, sizes of the copies
    // for the factors of K (owner chains) and of the Newton matrix (every chain)
    std::vector<int32_t> deep_map;
    int64_t deep_floats = 0, deep_floats_H = 0;
    bool deep_ok = false;    // every chain has a lane plan (block size <= 3)
    int rep = 1;
    int tile_nnz = kTileNnz;             // nonzeros per SpMV tile of K and G1 (kTileNnz, or half of it: see build_system)
    std::vector<int64_t> rep_n;          // per problem
    std::vector<int32_t> chain_owner;    // per chain
    std::vector<PrecWork> factor_work;   // what a factorisation of K visits: owner chains + the Jacobi blocks
    std::vector<ChainDesc> chainsH;
    std::vector<ChainLevelDesc> levelsH;
    size_t fac_doubles_H = 0;
    bool stored_row(int p, int64_t local) const {  // does K / G1 hold a row for this unknown?
        return rep <= 1 || local < rep_n[(size_t)p] || local >= (int64_t)rep * rep_n[(size_t)p];
    }

    // true: K's values, the chain factorisation and the Jacobi diagonal are computed here, on the
    // host, whenever a penalty changes (the CPU twin).  false: the backend derives them on the
    // device from K0 / K1 / rho (the HIP backend: k_kval + k_factor) -- nothing rho-dependent is
    // computed or uploaded by the host.
    bool factor_on_host = true;
    bool fac_fp32 = true;   // chain factors kept to float precision (score_settings.fac_fp32)
    // true: build_system left the matrices (A, K with K0 / K1, G1, G2), the scales, q, b, the tile tables and kkt_bytes to the
    // backend, which builds them on its device from the raw program (score_setup_device.hpp) and hands back what the host
    // side reads: row pointers, K's columns, the norms of q and b, the tiles.  D, E, q, b then live on the device only.
    bool device_setup = false;
    std::vector<char> rep_exact_all;  // per problem: its replicas' P values are bit-equal to replica 0's
};

inline int find_in_row(const Csr& M, int64_t row, int32_t col) {
    const int32_t* b = M.col.data() + M.ptr[row];
    const int32_t* e = M.col.data() + M.ptr[row + 1];
    const int32_t* it = std::lower_bound(b, e, col);
    if (it != e && *it == col) return (int)(it - M.col.data());
    return -1;
}

inline RowBlocks make_rowblocks(const Csr& M, const std::vector<RowSegment>& segs, int count, int tile_nnz = kTileNnz) {
    RowBlocks rb;
    rb.part_ptr.assign(count + 1, 0);
    for (const RowSegment& sg : segs) {
        int64_t r = sg.begin;
        const int64_t rend = sg.end;
        while (r < rend) {
            rb.first_row.push_back((int32_t)r);
            rb.prob.push_back(sg.prob);
            rb.rs.push_back(sg.rs);
            int64_t len0 = M.ptr[r + 1] - M.ptr[r];
            if (len0 > kLongRow) {  // a long row is a block of its own -- or several (kLongSeg)
                const int nseg = (len0 > kLongSeg) ? (int)std::min<int64_t>(kLongSegMax, (len0 + kLongSeg - 1) / kLongSeg) : 1;
                const int32_t bfirst = (int32_t)rb.prob.size() - 1;
                for (int sgi = 0; sgi < nseg; ++sgi) {
                    if (sgi) { rb.first_row.push_back((int32_t)r); rb.prob.push_back(sg.prob); rb.rs.push_back(sg.rs); }
                    rb.end_row.push_back((int32_t)(r + 1));
                    rb.kbeg.push_back(nseg > 1 ? (int32_t)(M.ptr[r] + len0 * sgi / nseg) : -1);
                    rb.kend.push_back(nseg > 1 ? (int32_t)(M.ptr[r] + len0 * (sgi + 1) / nseg) : -1);
                    rb.lfirst.push_back(bfirst);
                    rb.lseg.push_back(nseg > 1 ? (nseg | (sgi << 16)) : 0);
                    rb.lbase.push_back(rb.n_long_slots);
                    rb.lid.push_back(rb.n_long);
                }
                if (nseg > 1) { rb.n_long_slots += nseg; ++rb.n_long; }
                ++r;
                continue;
            }
            int64_t nn = 0, r1 = r;
            while (r1 < rend && r1 - r < kRowsPerBlock) {
                int64_t len = M.ptr[r1 + 1] - M.ptr[r1];
                if (len > kLongRow || nn + len > tile_nnz) break;
                nn += len;
                ++r1;
            }
            r = r1;
            rb.end_row.push_back((int32_t)r);
            rb.kbeg.push_back(-1); rb.kend.push_back(-1); rb.lfirst.push_back(0); rb.lseg.push_back(0); rb.lbase.push_back(0); rb.lid.push_back(0);
        }
        rb.part_ptr[sg.prob + 1] = (int32_t)rb.prob.size();  // (segments come problem by problem)
    }
    for (int p = 0; p < count; ++p) rb.part_ptr[p + 1] = std::max(rb.part_ptr[p + 1], rb.part_ptr[p]);
    rb.first_row.push_back(segs.empty() ? 0 : (int32_t)segs.back().end);
    return rb;
}
inline std::vector<RowSegment> plain_segments(const std::vector<int64_t>& xoff) {
    std::vector<RowSegment> sg;
    for (size_t p = 0; p + 1 < xoff.size(); ++p) sg.push_back(RowSegment{xoff[p], xoff[p + 1], (int32_t)p, 0});
    return sg;
}

// A CSR matrix whose pattern is borrowed from the caller's score_problem (valid for the duration of score_create)
// and whose values are owned (the equilibrated ones).
struct CsrScaled {
    const int32_t* ptr = nullptr;
    const int32_t* col = nullptr;
    std::vector<double> val;
    int64_t nnz = 0;
};
struct ProblemScaled {
    CsrScaled P, A;
    std::vector<double> q, b, D, E;
    // A' as a position map, built once for the equilibration and reused by append_problem: the entries of column j
    // of A are atpos[atp[j] .. atp[j+1]) (indices into A's arrays, in row order); arow[k] = row of entry k
    std::vector<int32_t> atp, atpos, arow;
};

// Ruiz equilibration of one problem; rows of one cone share a scale.
//   D, E <- 1; per pass: d_j = 1 / sqrt(|column j of [[P, A'], [A, 0]] scaled by the current D, E|_inf),
//   e_g likewise for the rows of cone group g (a zero-cone row is a group of its own); then D *= d, E *= e.
// The matrix itself is NOT rescaled between passes: a pass is one read sweep that forms |v| D_i D_j on the fly
// (P is symmetric -- the ABI takes the full matrix -- so its column norms are its row norms; A's columns go through
// the position map), and the values are scaled once at the end.  One team runs all passes; each member owns a
// range of columns and a range of cone groups, so no member ever updates another's entry.
// rep > 1 (a verified row-replicated problem, HostSystem::rep): the columns of replicas >= 1 and the tail rows
// >= 2 of every cone repeat replica 0's entries -- only replica 0 and the tail are swept, the others copy.
struct PhaseTimer;
inline void phase_mark(PhaseTimer* pt, const char* what);
// The passes of the equilibration on a device (the HIP backend: k_ruiz_*): same formulas, same result as the host
// loop below; returns false when it declines (then the host loop runs).  Everything it needs is in the problem and
// in the position maps ruiz_scale builds first.
struct RuizOffload {
    virtual ~RuizOffload() = default;
    virtual bool passes(const score_problem& p, int iters, int rep, int64_t rep_n, const std::vector<int32_t>& atp,
                        const std::vector<int32_t>& atpos, const std::vector<int32_t>& arow, const std::vector<int32_t>& gstart,
                        double* D, double* E) = 0;
};
inline void ruiz_scale(const score_problem& p, int iters, ProblemScaled& out, int rep = 1, int64_t rep_n = 0, PhaseTimer* pt = nullptr,
                       RuizOffload* offload = nullptr) {
    const int n = p.n, m = p.m;
    const int64_t nnzP = p.P_rowptr[n], nnzA = p.A_rowptr[m];
    out.P.ptr = p.P_rowptr; out.P.col = p.P_col; out.P.nnz = nnzP;
    out.A.ptr = p.A_rowptr; out.A.col = p.A_col; out.A.nnz = nnzA;
    out.D.assign(n, 1.0);
    out.E.assign(m, 1.0);
    out.atp.assign((size_t)n + 1, 0);
    out.atpos.resize((size_t)nnzA);
    out.arow.resize((size_t)nnzA);
    std::vector<int32_t>& atp = out.atp;
    for (int64_t k = 0; k < nnzA; ++k) atp[p.A_col[k] + 1]++;
    for (int j = 0; j < n; ++j) atp[j + 1] += atp[j];
    {
        std::vector<int32_t> fill(atp.begin(), atp.end() - 1);
        for (int r = 0; r < m; ++r)
            for (int k = p.A_rowptr[r]; k < p.A_rowptr[r + 1]; ++k) {
                out.atpos[(size_t)fill[p.A_col[k]]++] = (int32_t)k;
                out.arow[(size_t)k] = r;
            }
    }
    const int64_t ngroups = (int64_t)p.z + p.n_soc;
    std::vector<int32_t> gstart((size_t)ngroups + 1);
    for (int r = 0; r < p.z; ++r) gstart[r] = r;
    {
        int row = p.z;
        for (int c = 0; c < p.n_soc; ++c) { gstart[(size_t)p.z + c] = row; row += p.soc_dims[c]; }
        gstart[(size_t)ngroups] = row;
    }
    phase_mark(pt, "  ruiz: A' map, groups");
    const bool repl = rep > 1;
    const int64_t nr = repl ? rep_n : 0;
    const int64_t n_act = repl ? n - (int64_t)(rep - 1) * nr : n;   // columns swept: replica 0, then the tail
    auto act_col = [&](int64_t a) -> int64_t { return (!repl || a < nr) ? a : a + (int64_t)(rep - 1) * nr; };
    std::vector<double> d((size_t)n_act), e((size_t)ngroups);
    const int T = parallel_parts(n_act, 4096);
    TeamBarrier bar(T);
    double* D = out.D.data();
    double* E = out.E.data();
    // (parts of equal WORK: a pose column carries a dozen entries, a range variable's two)
    auto col_weight = [&](int64_t a) {
        const int64_t j = act_col(a);
        return (double)(p.P_rowptr[j + 1] - p.P_rowptr[j]) + 2.0 * (double)(atp[j + 1] - atp[j]);
    };
    const bool offloaded = iters > 0 && offload && offload->passes(p, iters, rep, nr, atp, out.atpos, out.arow, gstart, D, E);
    if (iters > 0 && !offloaded)
        parallel_ranges_balanced(n_act, 4096, col_weight, [&, T](int t, int64_t a0, int64_t a1) {
            const int64_t g0 = ngroups * t / T, g1 = ngroups * (t + 1) / T;
            struct Guard {  // (an exception in this part must not leave the others waiting at the barrier)
                TeamBarrier& b; bool ok = false;
                ~Guard() { if (!ok) b.abort(); }
            } guard{bar};
            for (int it = 0; it < iters; ++it) {
                for (int64_t a = a0; a < a1; ++a) {
                    const int64_t j = act_col(a);
                    double mx = 0.0;
                    for (int k = p.P_rowptr[j]; k < p.P_rowptr[j + 1]; ++k) mx = std::max(mx, std::fabs(p.P_val[k]) * D[p.P_col[k]]);
                    for (int k = atp[j]; k < atp[j + 1]; ++k) {
                        const int32_t q = out.atpos[(size_t)k];
                        mx = std::max(mx, std::fabs(p.A_val[q]) * E[out.arow[(size_t)q]]);
                    }
                    mx *= D[j];
                    d[(size_t)a] = mx > 1e-12 ? 1.0 / std::sqrt(mx) : 1.0;
                }
                for (int64_t g = g0; g < g1; ++g) {
                    const int r0 = gstart[(size_t)g];
                    // (replicated: the head row and the first tail row stand for the whole cone)
                    const int r1 = repl ? std::min(r0 + 2, gstart[(size_t)g + 1]) : gstart[(size_t)g + 1];
                    double mx = 0.0;
                    for (int k = p.A_rowptr[r0]; k < p.A_rowptr[r1]; ++k) mx = std::max(mx, std::fabs(p.A_val[k]) * D[p.A_col[k]]);
                    mx *= E[r0];
                    e[(size_t)g] = mx > 1e-12 ? 1.0 / std::sqrt(mx) : 1.0;
                }
                bar.wait();  // d, e complete (everybody has read the old D, E)
                for (int64_t a = a0; a < a1; ++a) {
                    const int64_t j = act_col(a);
                    const double v = D[j] * d[(size_t)a];
                    D[j] = v;
                    if (repl && a < nr)
                        for (int q = 1; q < rep; ++q) D[j + q * nr] = v;
                }
                for (int64_t g = g0; g < g1; ++g)
                    for (int r = gstart[(size_t)g]; r < gstart[(size_t)g + 1]; ++r) E[r] *= e[(size_t)g];
                bar.wait();  // D, E updated: the next pass reads them
            }
            guard.ok = true;
        }, T);
    phase_mark(pt, "  ruiz: passes");
    // the equilibrated values, once
    out.P.val.resize((size_t)nnzP);
    out.A.val.resize((size_t)nnzA);
    out.q.resize(n);
    out.b.resize(m);
    parallel_ranges(n, 16384, [&](int, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            const double di = D[i];
            for (int k = p.P_rowptr[i]; k < p.P_rowptr[i + 1]; ++k) out.P.val[(size_t)k] = p.P_val[k] * di * D[p.P_col[k]];
            out.q[(size_t)i] = p.q[i] * di;
        }
    });
    parallel_ranges(m, 16384, [&](int, int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const double er = E[r];
            for (int k = p.A_rowptr[r]; k < p.A_rowptr[r + 1]; ++k) out.A.val[(size_t)k] = p.A_val[k] * er * D[p.A_col[k]];
            out.b[(size_t)r] = p.b[r] * er;
        }
    });
    phase_mark(pt, "  ruiz: scaled values");
}

// Does the problem have the row-replicated structure its hint claims (include/score_hip.h, score_problem::rep_d)?
// Pattern and values, to a relative 1e-12 (assemblers that sum a row's terms in a different order per replica
// differ in the last bits; the solver then uses replica 0's values for all of them -- a perturbation far below
// the inexactness of its PCG solves, and the residual tests keep using the problem as given).
// exact (optional): the replicas' values of P are BIT-equal to replica 0's (a backend may then derive the replicas' copies of
// the scaled matrices from replica 0's and get what the host computes from the rows as given).
inline bool check_replication(const score_problem& p, bool* exact = nullptr) {
    const int d = p.rep_d;
    const int64_t nr = p.rep_n, n = p.n;
    if (d < 2 || d > 3 || nr < 1 || (int64_t)d * nr > n) return false;
    if (p.z != 0) return false;
    const int64_t t0 = (int64_t)d * nr;
    auto close = [](double a, double b) { return std::fabs(a - b) <= 1e-12 * std::max(std::fabs(a), std::fabs(b)); };
    std::atomic<bool> ok{true}, same{true};
    parallel_ranges(nr, 8192, [&](int, int64_t i0, int64_t i1) {
        bool all_same = true;
        for (int64_t i = i0; i < i1 && ok.load(std::memory_order_relaxed); ++i) {
            const int a0 = p.P_rowptr[i], a1 = p.P_rowptr[i + 1];
            for (int k = a0; k < a1; ++k)
                if (p.P_col[k] >= nr) { ok = false; return; }
            for (int rpl = 1; rpl < d; ++rpl) {
                const int64_t ir = i + rpl * nr;
                const int b0 = p.P_rowptr[ir];
                if (p.P_rowptr[ir + 1] - b0 != a1 - a0) { ok = false; return; }
                for (int k = 0; k < a1 - a0; ++k) {
                    if (p.P_col[b0 + k] != p.P_col[a0 + k] + rpl * nr || !close(p.P_val[b0 + k], p.P_val[a0 + k])) { ok = false; return; }
                    all_same = all_same && p.P_val[b0 + k] == p.P_val[a0 + k];
                }
            }
        }
        if (!all_same) same = false;
    });
    if (!ok) return false;
    if (exact) *exact = same.load();
    for (int64_t i = t0; i < n; ++i)
        for (int k = p.P_rowptr[i]; k < p.P_rowptr[i + 1]; ++k)
            if (p.P_col[k] < t0) return false;
    // cones: head row on tail columns, then d rows that repeat replica by replica (all cones have d + 1 rows: cone c starts at row c (d + 1))
    for (int c = 0; c < p.n_soc; ++c)
        if (p.soc_dims[c] != d + 1) return false;
    parallel_ranges(p.n_soc, 8192, [&](int, int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1 && ok.load(std::memory_order_relaxed); ++c) {
            const int64_t row = c * (d + 1);
            for (int k = p.A_rowptr[row]; k < p.A_rowptr[row + 1]; ++k)
                if (p.A_col[k] < t0) { ok = false; return; }
            const int a0 = p.A_rowptr[row + 1], a1 = p.A_rowptr[row + 2];
            for (int k = a0; k < a1; ++k)
                if (p.A_col[k] >= nr) { ok = false; return; }
            for (int rpl = 1; rpl < d; ++rpl) {
                const int b0 = p.A_rowptr[row + 1 + rpl];
                if (p.A_rowptr[row + 2 + rpl] - b0 != a1 - a0) { ok = false; return; }
                for (int k = 0; k < a1 - a0; ++k)
                    if (p.A_col[b0 + k] != p.A_col[a0 + k] + rpl * nr || !close(p.A_val[b0 + k], p.A_val[a0 + k])) { ok = false; return; }
            }
        }
    });
    if (!ok) return false;
    // chains: replica by replica, each a shifted copy of replica 0's
    if (p.n_chains > 0) {
        if (p.n_chains % d != 0) return false;
        const int nc0 = p.n_chains / d;
        const int nodes0 = p.chain_ptr[nc0] - p.chain_ptr[0];
        for (int rpl = 0; rpl < d; ++rpl)
            for (int c = 0; c < nc0; ++c) {
                const int cc = rpl * nc0 + c;
                if (p.chain_ptr[cc] != p.chain_ptr[c] + rpl * nodes0 || p.chain_ptr[cc + 1] != p.chain_ptr[c + 1] + rpl * nodes0) return false;
                for (int j = p.chain_ptr[c]; j < p.chain_ptr[c + 1]; ++j) {
                    const int64_t col = p.node_first_col[j];
                    if (rpl == 0 && (col < 0 || col + p.block_size > nr)) return false;
                    if (p.node_first_col[j + rpl * nodes0] != col + rpl * nr) return false;
                }
            }
    }
    return true;
}

inline void validate_problem(const score_problem& p) {
    if (p.n <= 0) throw std::runtime_error("score_problem: n must be positive");
    if (p.m < 0 || p.z < 0 || p.z > p.m) throw std::runtime_error("score_problem: bad m / z");
    if (!p.P_rowptr || !p.q || !p.A_rowptr) throw std::runtime_error("score_problem: null array");
    int64_t tot = p.z;
    for (int c = 0; c < p.n_soc; ++c) {
        if (p.soc_dims[c] < 1) throw std::runtime_error("score_problem: cone dimension < 1");
        tot += p.soc_dims[c];
    }
    if (tot != p.m) throw std::runtime_error("score_problem: z + sum(soc_dims) != m");
    if (p.P_rowptr[0] != 0 || p.A_rowptr[0] != 0) throw std::runtime_error("score_problem: rowptr[0] != 0");
    // (row pointers first, serially: the column checks below index with them.  Then the rows in parallel parts; a part that
    //  finds a fault throws, run_parts hands the first part's exception on)
    for (int i = 0; i < p.n; ++i)
        if (p.P_rowptr[i + 1] < p.P_rowptr[i]) throw std::runtime_error("score_problem: P_rowptr not monotone");
    for (int r = 0; r < p.m; ++r)
        if (p.A_rowptr[r + 1] < p.A_rowptr[r]) throw std::runtime_error("score_problem: A_rowptr not monotone");
    parallel_ranges(p.n, 16384, [&](int, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i)
            for (int k = p.P_rowptr[i]; k < p.P_rowptr[i + 1]; ++k) {
                if (p.P_col[k] < 0 || p.P_col[k] >= p.n) throw std::runtime_error("score_problem: P column out of range");
                if (k > p.P_rowptr[i] && p.P_col[k] <= p.P_col[k - 1])
                    throw std::runtime_error("score_problem: P columns must be sorted and unique per row");
            }
    });
    parallel_ranges(p.m, 16384, [&](int, int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r)
            for (int k = p.A_rowptr[r]; k < p.A_rowptr[r + 1]; ++k) {
                if (p.A_col[k] < 0 || p.A_col[k] >= p.n) throw std::runtime_error("score_problem: A column out of range");
                if (k > p.A_rowptr[r] && p.A_col[k] <= p.A_col[k - 1])
                    throw std::runtime_error("score_problem: A columns must be sorted and unique per row");
            }
    });
    if (p.n_chains > 0) {
        if (p.block_size < 1 || p.block_size > kMaxBs) throw std::runtime_error("score_problem: block_size must be 1..4");
        if (!p.chain_ptr || !p.node_first_col) throw std::runtime_error("score_problem: null chain hint");
        std::vector<char> used(p.n, 0);
        for (int j = p.chain_ptr[0]; j < p.chain_ptr[p.n_chains]; ++j)
            for (int c = 0; c < p.block_size; ++c) {
                int col = p.node_first_col[j] + c;
                if (col < 0 || col >= p.n) throw std::runtime_error("score_problem: chain node column out of range");
                if (used[col]) throw std::runtime_error("score_problem: chain nodes overlap");
                used[col] = 1;
            }
    }
}

// wall-clock marks of the setup phases, printed when settings.verbose is set
struct PhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    long flt = 0, csw = 0;
    static void usage(long& f, long& c) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        f = ru.ru_minflt;
        c = ru.ru_nvcsw + ru.ru_nivcsw;
    }
    explicit PhaseTimer(bool enabled) : on(enabled), t(std::chrono::steady_clock::now()) {
        if (on) usage(flt, csw);
    }
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        long f, c;
        usage(f, c);
        std::fprintf(stderr, "[score setup] %-28s %8.2f ms  (%ld page faults, %ld context switches)\n", what,
                     std::chrono::duration<double, std::milli>(now - t).count(), f - flt, c - csw);
        t = std::chrono::steady_clock::now();
        flt = f;
        csw = c;
    }
};

inline void phase_mark(PhaseTimer* pt, const char* what) { if (pt) pt->mark(what); }

// Append problem `b` (already scaled) to the batch: A, K (pattern + K0/K1), G1, G2.
inline void append_problem(HostSystem& H, int pi, const score_problem& p, const ProblemScaled& S, PhaseTimer& pt) {
    const int n = p.n, m = p.m;
    const int64_t xo = H.xoff[pi], ro = H.roff[pi];
    // ---- A (global indices) ----
    const size_t a_base = H.A.col.size(), a_row0 = H.A.ptr.size();
    H.A.col.resize(a_base + (size_t)S.A.nnz); H.A.val.resize(a_base + (size_t)S.A.nnz);
    H.A.ptr.resize(a_row0 + (size_t)m);
    parallel_ranges(m, 16384, [&](int, int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            for (int k = S.A.ptr[r]; k < S.A.ptr[r + 1]; ++k) {
                H.A.col[a_base + (size_t)k] = (int32_t)(xo + S.A.col[k]);
                H.A.val[a_base + (size_t)k] = S.A.val[(size_t)k];
            }
            H.A.ptr[a_row0 + (size_t)r] = (int32_t)(a_base + (size_t)S.A.ptr[r + 1]);
        }
    });
    // ---- A' (local): the position map of the equilibration, with the scaled values ----
    const std::vector<int32_t>& atp = S.atp;
    std::vector<int32_t> atr((size_t)S.A.nnz);
    std::vector<double> atv((size_t)S.A.nnz);
    parallel_ranges(S.A.nnz, 65536, [&](int, int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const int32_t q = S.atpos[(size_t)k];
            atr[(size_t)k] = S.arow[(size_t)q];
            atv[(size_t)k] = S.A.val[(size_t)q];
        }
    });
    pt.mark("  append: A, A'");
    // ---- K = P + sigma I + rho A'A, kept as K0 + rho K1 on the union pattern ----
    // A row of K has a few dozen entries: it is gathered into a small buffer, ordered by column
    // (stable, so equal columns are summed in the order they were met) and merged; row ranges run
    // in parallel and touch no O(n) scratch.
    struct Ent { int32_t j; double v0, v1; };
    auto gather_row = [&](int64_t i, std::vector<Ent>& buf) {
        buf.clear();
        buf.push_back(Ent{(int32_t)i, H.sigma, 0.0});
        for (int k = S.P.ptr[i]; k < S.P.ptr[i + 1]; ++k) buf.push_back(Ent{S.P.col[k], S.P.val[k], 0.0});
        for (int t2 = atp[i]; t2 < atp[i + 1]; ++t2) {
            const int r = atr[t2];
            const double a = atv[t2];
            for (int k = S.A.ptr[r]; k < S.A.ptr[r + 1]; ++k) buf.push_back(Ent{S.A.col[k], 0.0, a * S.A.val[k]});
        }
        // stable order by column.  Typical rows hold a few dozen mostly ordered entries (insertion sort);
        // a landmark seen by thousands of ranges gathers thousands, alternating between its own column and
        // the other endpoint's -- quadratic for an insertion sort (8 such rows were 7 of the 8 ms of this phase)
        if (buf.size() > 64) {
            std::stable_sort(buf.begin(), buf.end(), [](const Ent& x, const Ent& y) { return x.j < y.j; });
        } else {
            for (size_t x = 1; x < buf.size(); ++x) {
                const Ent e = buf[x];
                size_t y = x;
                while (y > 0 && buf[y - 1].j > e.j) { buf[y] = buf[y - 1]; --y; }
                buf[y] = e;
            }
        }
    };
    const size_t k_row0 = H.K.ptr.size() - 1;  // == xo
    H.K.ptr.resize(k_row0 + 1 + n);
    // one sweep: every thread merges its rows into its own buffers (reserved for the worst case, so
    // they never re-allocate), the row lengths are prefix-summed, the buffers copied into place
    struct KPart { std::vector<int32_t> col; std::vector<double> k0, k1; int64_t i0 = 0, i1 = 0; };
    const int k_parts = parallel_parts(n, 8192);
    std::vector<KPart> kparts((size_t)k_parts);
    // (replicated problems: K holds the rows of replica 0 and of the tail only, see HostSystem::rep)
    auto stored = [&](int64_t i) { return H.stored_row(pi, i); };
    auto k_row_weight = [&](int64_t i) {  // entries gathered for row i; long rows cost n log n in the sort
        if (!stored(i)) return 0.0;
        const double g = (double)(S.P.ptr[i + 1] - S.P.ptr[i]) + 2.0 * (double)(atp[i + 1] - atp[i]);
        return g > 64.0 ? 4.0 * g : g;
    };
    parallel_ranges_balanced(n, 8192, k_row_weight, [&](int t, int64_t i0, int64_t i1) {
        KPart Q;  // thread-local (no cache line shared with a neighbour's vector headers), handed over at the end
        Q.i0 = i0; Q.i1 = i1;
        size_t ub = 0;
        for (int64_t i = i0; i < i1; ++i) {
            if (!stored(i)) continue;
            ub += 1 + (size_t)(S.P.ptr[i + 1] - S.P.ptr[i]);
            for (int t2 = atp[i]; t2 < atp[i + 1]; ++t2) ub += (size_t)(S.A.ptr[atr[t2] + 1] - S.A.ptr[atr[t2]]);
        }
        Q.col.reserve(ub); Q.k0.reserve(ub); Q.k1.reserve(ub);
        std::vector<Ent> buf;
        for (int64_t i = i0; i < i1; ++i) {
            if (!stored(i)) { H.K.ptr[k_row0 + 1 + i] = 0; continue; }
            gather_row(i, buf);
            int32_t cnt = 0;
            size_t x = 0;
            while (x < buf.size()) {
                const int32_t j = buf[x].j;
                double a0 = 0.0, a1 = 0.0;
                for (; x < buf.size() && buf[x].j == j; ++x) { a0 += buf[x].v0; a1 += buf[x].v1; }
                Q.col.push_back((int32_t)(xo + j));
                Q.k0.push_back(a0);
                Q.k1.push_back(a1);
                ++cnt;
            }
            H.K.ptr[k_row0 + 1 + i] = cnt;
        }
        kparts[t] = std::move(Q);
    }, k_parts);
    pt.mark("  append: K rows");
    for (int i = 0; i < n; ++i) H.K.ptr[k_row0 + 1 + i] += H.K.ptr[k_row0 + i];
    const size_t k_end = (size_t)H.K.ptr[k_row0 + n];
    H.K.col.resize(k_end); H.K0.resize(k_end); H.K1.resize(k_end);
    parallel_ranges((int64_t)kparts.size(), 1, [&](int, int64_t p0, int64_t p1) {
        for (int64_t pi2 = p0; pi2 < p1; ++pi2) {
            const KPart& Q = kparts[pi2];
            if (Q.col.empty()) continue;
            const size_t o = (size_t)H.K.ptr[k_row0 + Q.i0];
            std::memcpy(&H.K.col[o], Q.col.data(), Q.col.size() * sizeof(int32_t));
            std::memcpy(&H.K0[o], Q.k0.data(), Q.k0.size() * sizeof(double));
            std::memcpy(&H.K1[o], Q.k1.data(), Q.k1.size() * sizeof(double));
        }
    });
    pt.mark("  append: K fill");
    // ---- G1 = [0 | A'] and G2 = [P | A'] (columns address the contiguous buffer [xt ; u]) ----
    const size_t g1_base = H.G1.col.size(), g2_base = H.G2.col.size();
    const size_t g1_row0 = H.G1.ptr.size(), g2_row0 = H.G2.ptr.size(), sp_row0 = H.g2_split.size();
    H.G1.ptr.resize(g1_row0 + n); H.G2.ptr.resize(g2_row0 + n); H.g2_split.resize(sp_row0 + n);
    std::vector<int32_t> g1p((size_t)n + 1, 0);  // G1 row starts (local): stored rows only
    for (int i = 0; i < n; ++i) {
        g1p[(size_t)i + 1] = g1p[(size_t)i] + (stored(i) ? atp[i + 1] - atp[i] : 0);
        H.G1.ptr[g1_row0 + i] = (int32_t)(g1_base + g1p[(size_t)i + 1]);
        H.g2_split[sp_row0 + i] = (int32_t)(g2_base + S.P.ptr[i + 1] + atp[i]);
        H.G2.ptr[g2_row0 + i] = (int32_t)(g2_base + S.P.ptr[i + 1] + atp[i + 1]);
    }
    // (four value-initialising resizes of 2-9 MB each: one thread per array)
    parallel_ranges(4, 1, [&](int, int64_t v0, int64_t v1) {
        for (int64_t v = v0; v < v1; ++v) {
            if (v == 0) H.G1.col.resize(g1_base + (size_t)g1p[(size_t)n]);
            else if (v == 1) H.G1.val.resize(g1_base + (size_t)g1p[(size_t)n]);
            else if (v == 2) H.G2.col.resize(g2_base + (size_t)S.P.nnz + atr.size());
            else H.G2.val.resize(g2_base + (size_t)S.P.nnz + atr.size());
        }
    });
    pt.mark("  append: G resize");
    const int32_t ucol0 = (int32_t)(H.n_tot + ro);
    parallel_ranges(n, 16384, [&](int, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            size_t o1 = g1_base + (size_t)g1p[(size_t)i], o2 = g2_base + S.P.ptr[i] + atp[i];
            const bool st_ = stored(i);
            for (int k = S.P.ptr[i]; k < S.P.ptr[i + 1]; ++k, ++o2) {
                H.G2.col[o2] = (int32_t)(xo + S.P.col[k]);
                H.G2.val[o2] = S.P.val[k];
            }
            for (int t2 = atp[i]; t2 < atp[i + 1]; ++t2, ++o2) {
                if (st_) {
                    H.G1.col[o1] = ucol0 + atr[t2];
                    H.G1.val[o1] = atv[t2];
                    ++o1;
                }
                H.G2.col[o2] = ucol0 + atr[t2];
                H.G2.val[o2] = atv[t2];
            }
        }
    });
}

// (Re)compute everything that depends on rho for problem `pi`: K values, G1
// values, the chain factorisation and the Jacobi diagonal.
inline void refresh_rho(HostSystem& H, int pi) {
    if (!H.factor_on_host) return;
    const double rho = H.rho[pi];
    const int64_t r0 = H.xoff[pi], r1 = H.xoff[pi + 1];
    const int64_t k0 = H.K.ptr[r0], k1 = H.K.ptr[r1];
    parallel_ranges(k1 - k0, 1 << 17, [&](int, int64_t a, int64_t b) {
        for (int64_t k = k0 + a; k < k0 + b; ++k) H.K.val[k] = H.K0[k] + rho * H.K1[k];
    });
    const int bs = H.bs, b2 = bs * bs;
    std::vector<int32_t> mine;  // (a replica's chain uses its owner's factors)
    for (size_t ci = 0; ci < H.chains.size(); ++ci)
        if (H.chains[ci].prob == pi && H.chain_owner[ci] == (int32_t)ci) mine.push_back((int32_t)ci);
    parallel_ranges((int64_t)mine.size(), 1, [&](int, int64_t c0, int64_t c1) {
        std::vector<double> Ad, Bs, fac;
        std::vector<ChainLevelDesc> lv;
        for (int64_t c = c0; c < c1; ++c) {
            const size_t ci = mine[c];
            const ChainDesc& ch = H.chains[ci];
            Ad.assign((size_t)ch.N * b2, 0.0);
            Bs.assign((size_t)ch.N * b2, 0.0);
            for (int i = 0; i < ch.N; ++i) {
                const size_t g = (size_t)(ch.node_begin + i) * b2;
                for (int k = 0; k < b2; ++k) {
                    int pd = H.pos_diag[g + k];
                    Ad[(size_t)i * b2 + k] = pd >= 0 ? H.K.val[pd] : 0.0;
                    int ps = H.pos_sub[g + k];
                    Bs[(size_t)i * b2 + k] = (i > 0 && ps >= 0) ? H.K.val[ps] : 0.0;
                }
            }
            lv.clear(); fac.clear();
            int scr = 0;
            factor_chain_levels(bs, H.radix, ch.N, Ad, Bs, lv, fac, scr);
            // level structure depends only on (N, radix): it was laid out at setup
            if (H.fac_fp32)  // what the device keeps: the factors to float precision (k_fac_round)
                for (double& v : fac) v = (double)(float)v;
            std::memcpy(&H.fac[H.fac_off[ci]], fac.data(), sizeof(double) * fac.size());
        }
    });
    for (size_t e = 0; e < H.diag_cols.size(); ++e) {
        int32_t c = H.diag_cols[e];
        if (c < r0 || c >= r1) continue;
        H.dinv[e] = 1.0 / H.K.val[H.diag_kpos[e]];
    }
}

// The tile tables of K, G1 and G2 from their row pointers (three independent serial scans).
inline void make_system_rowblocks(HostSystem& H) {
    const int count = H.count;
    parallel_ranges(3, 1, [&](int, int64_t k0, int64_t k1) {  // three independent serial scans
        for (int64_t k = k0; k < k1; ++k) {
            // K and G1 of a replicated problem: the rows of replica 0 (applied to all replicas), then the tail
            std::vector<RowSegment> sg;
            if (k < 2 && H.rep > 1) {
                for (int p = 0; p < count; ++p) {
                    const int64_t nr = H.rep_n[(size_t)p];
                    sg.push_back(RowSegment{H.xoff[p], H.xoff[p] + nr, p, (int32_t)nr});
                    sg.push_back(RowSegment{H.xoff[p] + (int64_t)H.rep * nr, H.xoff[p + 1], p, 0});
                }
            } else {
                sg = plain_segments(H.xoff);
            }
            // (G1 = A' has one or two entries per row: its tiles are bounded by their 256 rows, far below 1024 nonzeros --
            //  a replicated problem runs it with 4 instead of 8 nonzero slots per lane: half the issued loads and LDS
            //  planes; measured rhs 6.7 -> 6.2 us on the headline problem, 55.9 -> 47.6 us in the batch of 16)
            if (k == 0) H.rbK = make_rowblocks(H.K, sg, count, H.tile_nnz);
            else if (k == 1) H.rbG1 = make_rowblocks(H.G1, sg, count, H.rep > 1 ? kTileNnz / 2 : H.tile_nnz);
            else H.rbG2 = make_rowblocks(H.G2, sg, count);
        }
    });
}

// algorithmic bytes of one K-apply per problem: 12 B per nonzero (fp64 value
// + int32 column), 4 B per row pointer, vector p read once, w written once
inline void fill_kkt_bytes(HostSystem& H) {
    H.kkt_bytes.assign(H.count, 0.0);
    for (int p = 0; p < H.count; ++p) {
        // (a replicated problem streams K_row and the tail once: the rows K holds; p and w are whole vectors)
        const double nnz = (double)(H.K.ptr[H.xoff[p + 1]] - H.K.ptr[H.xoff[p]]);
        const double n = (double)(H.xoff[p + 1] - H.xoff[p]);
        const double rows = H.rep > 1 ? n - (double)(H.rep - 1) * (double)H.rep_n[(size_t)p] : n;
        H.kkt_bytes[p] = 12.0 * nnz + 4.0 * (rows + 1) + 16.0 * n;
    }
}

// allow_rep: the backend can run replicated problems (HostSystem::rep); the CPU twin cannot (and is the better
// check for not doing so: it applies the full K the problem defines).
struct DeviceSetupDeclined : std::runtime_error {
    DeviceSetupDeclined() : std::runtime_error("device setup declined") {}
};
// device_setup_ok (optional): called once sizes and the replication structure are known; true = the backend builds the
// matrices on its device (HostSystem::device_setup) and the host sweeps over them are skipped.
inline void build_system(const score_problem* probs, int count, const score_settings& st, HostSystem& H,
                         bool factor_on_host = true, bool allow_rep = false, RuizOffload* ruiz_offload = nullptr,
                         const std::function<bool(const HostSystem&)>& device_setup_ok = nullptr, bool trusted = false) {
    if (count <= 0) throw std::runtime_error("score_create: count must be positive");
    BuildScope scope;
    PhaseTimer pt(st.verbose != 0);
    H = HostSystem();
    H.factor_on_host = factor_on_host;
    H.fac_fp32 = st.fac_fp32 != 0;
    H.count = count;
    H.sigma = st.sigma;
    H.radix = std::min(4, std::max(2, st.chain_radix));
    H.xoff.assign(count + 1, 0);
    H.roff.assign(count + 1, 0);
    int bs = 0;
    // trusted: skeleton problems of the library's own device assembler (sizes, cones, chain and replication hints; no
    // matrices) -- nothing to validate, the replicated structure holds by construction
    for (int p = 0; p < count; ++p) {
        if (!trusted) validate_problem(probs[p]);
        pt.mark("validate");
        H.xoff[p + 1] = H.xoff[p] + probs[p].n;
        H.roff[p + 1] = H.roff[p] + probs[p].m;
        if (probs[p].n_chains > 0) {
            if (bs && bs != probs[p].block_size)
                throw std::runtime_error("score_create_batch: all problems must share block_size");
            bs = probs[p].block_size;
        }
    }
    H.bs = bs;
    H.n_tot = H.xoff[count];
    H.m_tot = H.roff[count];
    {   // row replication: every problem of the batch must carry the same, verified, hint
        int rep = (allow_rep && std::getenv("SCORE_NO_REPLICATION") == nullptr) ? probs[0].rep_d : 0;
        H.rep_exact = false;
        H.rep_exact_all.assign((size_t)count, 0);
        if (trusted) {
            for (int p = 0; p < count; ++p)
                if (probs[p].rep_d != rep) rep = 0;
            H.rep_exact = rep > 1 && count == 1;
            H.rep_exact_all.assign((size_t)count, rep > 1 ? 1 : 0);
        } else
        if (rep > 1 && count == 1) {
            bool ex = false;
            if (probs[0].rep_d != rep || !check_replication(probs[0], &ex)) rep = 0;
            H.rep_exact = rep > 1 && ex;
            H.rep_exact_all[0] = H.rep_exact;
        } else if (rep > 1 && count > 1) {  // (a batch: one problem per part, the sweeps inside run serially)
            std::atomic<bool> all{true};
            parallel_ranges(count, 1, [&](int, int64_t p0, int64_t p1) {
                for (int64_t p = p0; p < p1 && all.load(std::memory_order_relaxed); ++p) {
                    bool ex = false;
                    if (probs[p].rep_d != rep || !check_replication(probs[p], &ex)) all = false;
                    H.rep_exact_all[(size_t)p] = ex;
                }
            });
            if (!all) rep = 0;
        }
        H.rep = rep > 1 ? rep : 1;
        H.rep_n.assign((size_t)count, 0);
        if (H.rep > 1)
            for (int p = 0; p < count; ++p) H.rep_n[(size_t)p] = probs[p].rep_n;
        pt.mark("replication check");
    }
    if (H.n_tot + H.m_tot >= (int64_t)1 << 31) throw std::runtime_error("batch too large for 32-bit indices");
    const bool lite = device_setup_ok && device_setup_ok(H);
    H.device_setup = lite;
    if (trusted && !lite) throw DeviceSetupDeclined();  // (skeleton problems hold no matrices: the caller assembles on the host)
    H.A.nrows = H.m_tot; H.A.ncols = H.n_tot; H.A.ptr.assign(1, 0);
    H.K.nrows = H.K.ncols = H.n_tot; H.K.ptr.assign(1, 0);
    H.G1.nrows = H.n_tot; H.G1.ncols = H.n_tot + H.m_tot; H.G1.ptr.assign(1, 0);
    H.G2.nrows = H.n_tot; H.G2.ncols = H.n_tot + H.m_tot; H.G2.ptr.assign(1, 0);
    H.cone_part_ptr.assign(count + 1, 0);
    H.prec_part_ptr.assign(count + 1, 0);
    H.rho.assign(count, st.rho);
    // Equilibration is independent per problem: a batch of small problems (each below the row count
    // at which ruiz_scale splits its own passes over threads) is scaled one problem per thread.
    std::vector<ProblemScaled> scaled((size_t)count);
    // (host setup of a BATCH: the passes stay on the host -- offloading each problem's passes from its part's thread measured
    //  slower, 850-1010 graphs/s against 980-1090: profiles/TRIED.md; the device setup path does not come here at all)
    // SCORE_HOST_SETUP=1 (the host setup asked for explicitly -- tests compare it with the device setup bit for bit): the passes
    // run where the device setup's do, so that both sides start from the same scales.
    RuizOffload* batch_offload = std::getenv("SCORE_HOST_SETUP") ? ruiz_offload : nullptr;
    if (count > 1 && !lite) {
        parallel_ranges(count, 1, [&](int, int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; ++p) ruiz_scale(probs[p], std::max(0, st.scale_iters), scaled[(size_t)p], H.rep, H.rep_n[(size_t)p], nullptr, batch_offload);
        });
        pt.mark("ruiz (all problems)");
    }
    // The per-problem pieces of A, K (pattern + K0 / K1), G1 and G2 do not depend on each other: in a
    // batch they are built one problem per thread into private mini-systems -- global column indices
    // (the offsets are known up front), local row pointers -- and stitched together below, each piece
    // copied into its own range.  A single problem is appended in place (its row loops are split over
    // threads inside append_problem).  Either way the arrays come out identical.
    std::vector<HostSystem> pieces;
    if (count > 1 && !lite) {
        pieces.resize((size_t)count);
        parallel_ranges(count, 1, [&](int, int64_t p0, int64_t p1) {
            PhaseTimer quiet(false);
            for (int64_t p = p0; p < p1; ++p) {
                HostSystem& Q = pieces[(size_t)p];
                Q.count = 1;
                Q.rep = H.rep;
                Q.rep_n.assign(1, H.rep_n[(size_t)p]);
                Q.sigma = H.sigma;
                Q.n_tot = H.n_tot; Q.m_tot = H.m_tot;
                Q.xoff.assign(1, H.xoff[p]); Q.roff.assign(1, H.roff[p]);
                Q.A.ptr.assign(1, 0); Q.K.ptr.assign(1, 0); Q.G1.ptr.assign(1, 0); Q.G2.ptr.assign(1, 0);
                append_problem(Q, 0, probs[p], scaled[(size_t)p], quiet);
            }
        });
        pt.mark("append (K, G1, G2), all problems");
        // sizes, then parallel copies into place
        std::vector<size_t> oA(count + 1, 0), oK(count + 1, 0), oG1(count + 1, 0), oG2(count + 1, 0);
        for (int p = 0; p < count; ++p) {
            oA[p + 1] = oA[p] + pieces[p].A.col.size();
            oK[p + 1] = oK[p] + pieces[p].K.col.size();
            oG1[p + 1] = oG1[p] + pieces[p].G1.col.size();
            oG2[p + 1] = oG2[p] + pieces[p].G2.col.size();
        }
        H.A.col.resize(oA[count]); H.A.val.resize(oA[count]); H.A.ptr.resize((size_t)H.m_tot + 1);
        H.K.col.resize(oK[count]); H.K0.resize(oK[count]); H.K1.resize(oK[count]); H.K.ptr.resize((size_t)H.n_tot + 1);
        H.G1.col.resize(oG1[count]); H.G1.val.resize(oG1[count]); H.G1.ptr.resize((size_t)H.n_tot + 1);
        H.G2.col.resize(oG2[count]); H.G2.val.resize(oG2[count]); H.G2.ptr.resize((size_t)H.n_tot + 1);
        H.g2_split.resize((size_t)H.n_tot);
        parallel_ranges(count, 1, [&](int, int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; ++p) {
                HostSystem& Q = pieces[(size_t)p];
                const size_t n = (size_t)probs[p].n, m = (size_t)probs[p].m;
                const size_t xo = (size_t)H.xoff[p], ro = (size_t)H.roff[p];
                auto copy = [](auto& dst, size_t off, const auto& src) { if (!src.empty()) std::memcpy(&dst[off], src.data(), src.size() * sizeof(src[0])); };
                copy(H.A.col, oA[p], Q.A.col); copy(H.A.val, oA[p], Q.A.val);
                for (size_t r = 0; r < m; ++r) H.A.ptr[ro + r + 1] = (int32_t)(oA[p] + (size_t)Q.A.ptr[r + 1]);
                copy(H.K.col, oK[p], Q.K.col); copy(H.K0, oK[p], Q.K0); copy(H.K1, oK[p], Q.K1);
                copy(H.G1.col, oG1[p], Q.G1.col); copy(H.G1.val, oG1[p], Q.G1.val);
                copy(H.G2.col, oG2[p], Q.G2.col); copy(H.G2.val, oG2[p], Q.G2.val);
                for (size_t i = 0; i < n; ++i) {
                    H.K.ptr[xo + i + 1] = (int32_t)(oK[p] + (size_t)Q.K.ptr[i + 1]);
                    H.G1.ptr[xo + i + 1] = (int32_t)(oG1[p] + (size_t)Q.G1.ptr[i + 1]);
                    H.G2.ptr[xo + i + 1] = (int32_t)(oG2[p] + (size_t)Q.G2.ptr[i + 1]);
                    H.g2_split[xo + i] = (int32_t)(oG2[p] + (size_t)Q.g2_split[i]);
                }
                Q = HostSystem();  // release the piece
            }
        });
        H.A.ptr[0] = 0; H.K.ptr[0] = 0; H.G1.ptr[0] = 0; H.G2.ptr[0] = 0;
        pt.mark("stitch");
    }
    for (int p = 0; p < count; ++p) {
        const score_problem& pr = probs[p];
        if (count == 1 && !lite) {
            ruiz_scale(pr, std::max(0, st.scale_iters), scaled[0], H.rep, H.rep_n[0], &pt, ruiz_offload);  // (one problem: the passes may run on the device)
            pt.mark("ruiz");
        }
        H.c0.push_back(pr.c0);
        if (!lite) {
        ProblemScaled S = std::move(scaled[(size_t)p]);
        H.D.insert(H.D.end(), S.D.begin(), S.D.end());
        H.E.insert(H.E.end(), S.E.begin(), S.E.end());
        H.q.insert(H.q.end(), S.q.begin(), S.q.end());
        H.b.insert(H.b.end(), S.b.begin(), S.b.end());
        double qu = 0, bu = 0, qs = 0, bsn = 0;
        for (int j = 0; j < pr.n; ++j) { qu = std::max(qu, std::fabs(pr.q[j])); qs = std::max(qs, std::fabs(S.q[j])); }
        for (int r = 0; r < pr.m; ++r) { bu = std::max(bu, std::fabs(pr.b[r])); bsn = std::max(bsn, std::fabs(S.b[r])); }
        H.qnorm_u.push_back(qu); H.bnorm_u.push_back(bu); H.qnorm_s.push_back(qs); H.bnorm_s.push_back(bsn);
        if (count == 1) {
            append_problem(H, p, pr, S, pt);
            pt.mark("append (K, G1, G2)");
        }
        }
        // cones, in blocks of kConesPerBlock that never straddle problems
        const size_t c_first = H.cone_row.size();
        for (int r = 0; r < pr.z; ++r) {
            H.cone_row.push_back((int32_t)(H.roff[p] + r)); H.cone_dim.push_back(1); H.cone_type.push_back(0);
        }
        int row = pr.z;
        for (int c = 0; c < pr.n_soc; ++c) {
            H.cone_row.push_back((int32_t)(H.roff[p] + row)); H.cone_dim.push_back(pr.soc_dims[c]); H.cone_type.push_back(1);
            row += pr.soc_dims[c];
        }
        for (size_t c = c_first; c < H.cone_row.size(); c += kConesPerBlock) {
            H.cone_block_first.push_back((int32_t)c);
            H.cone_block_prob.push_back(p);
        }
        H.cone_part_ptr[p + 1] = (int32_t)H.cone_block_prob.size();
    }
    H.cone_block_first.push_back((int32_t)H.cone_row.size());
    if (factor_on_host) H.K.val.assign(H.K.col.size(), 0.0);  // (a backend that derives K = K0 + rho K1 itself never reads it)
    // Tile size of K and G1.  Measured on the headline problem,
    // whose replicated K is 250 tiles of 2048 nonzeros -- one per CU: half-size tiles are SLOWER (kp 7.3 -> 7.8 us,
    // kpb 7.8 -> 8.6 us): the SpMV of a single problem is a chain of dependent trips to memory, not a throughput loop.
    H.tile_nnz = kTileNnz;
    if (!lite) make_system_rowblocks(H);
    pt.mark("row blocks");

    // ---- preconditioner layout ----
    const int b2 = bs * bs;
    std::vector<char> in_chain(H.n_tot, 0);
    std::vector<int32_t> node_prev;  // column of each chain node's predecessor (-1: first of its chain)
    const int max_nodes = 1 << 20;
    // (the twin factors on the host and keeps whole chains: its streaming solve is the specification)
    bool segments_ok = !factor_on_host && H.radix == 4 && bs >= 1 && bs <= 4 && std::getenv("SCORE_NO_SEGMENTS") == nullptr;
    // The second level keeps one long chain's separators in LDS (score_join.hpp: kJoinMaxSeps = 128: chains of up to 129
    // segments, 132 k nodes; round 5 stopped at 65 segments).  A handle with a longer chain keeps ALL its chains whole for the
    // streaming kernel k_prec -- which holds the coarse levels' vectors in LDS and therefore ends at ~18 k nodes of 3 x 3 blocks:
    // score_create then fails with "chain too long for the LDS-resident chain solver", as it did for such chains in every round.
    if (segments_ok) {
        const int seg_max = seg_max_nodes();
        for (int p = 0; p < count && segments_ok; ++p)
            for (int c = 0; c < probs[p].n_chains; ++c) {
                const int Nall = probs[p].chain_ptr[c + 1] - probs[p].chain_ptr[c];
                if (Nall > seg_max && (Nall + 1 + seg_max) / (seg_max + 1) - 1 > kJoinMaxSepsHost) { segments_ok = false; break; }
            }
    }
    for (int p = 0; p < count; ++p) {
        const score_problem& pr = probs[p];
        const size_t jc_first = H.join_chains.size();
        for (int c = 0; c < pr.n_chains; ++c) {
            const int nb = pr.chain_ptr[c], ne = pr.chain_ptr[c + 1];
            if (ne <= nb) continue;
            if (ne - nb > max_nodes) throw std::runtime_error("chain too long");
            // a chain of more than kSegMaxNodes nodes: segments with one separator node between neighbours (JoinChain)
            const int Nall = ne - nb;
            const int seg_max = seg_max_nodes();
            const int n_seg = (segments_ok && Nall > seg_max) ? (Nall + 1 + seg_max) / (seg_max + 1) : 1;
            if (n_seg > 1) {
                JoinChain jc{};
                jc.prob = p; jc.n_seg = n_seg; jc.first_chain = (int32_t)H.chains.size();
                jc.sep_begin = (int32_t)H.join_sep_col.size(); jc.owner = (int32_t)H.join_chains.size();
                H.join_chains.push_back(jc);
            }
            const int seg_nodes = Nall - (n_seg - 1);
            int at = nb;
            for (int sg = 0; sg < n_seg; ++sg) {
                const int len = seg_nodes / n_seg + (sg < seg_nodes % n_seg ? 1 : 0);
                ChainDesc ch{};
                ch.prob = p;
                ch.node_begin = (int32_t)H.node_col.size();
                ch.N = len;
                for (int j = at; j < at + len; ++j) {
                    const int32_t col = (int32_t)(H.xoff[p] + pr.node_first_col[j]);
                    H.node_col.push_back(col);
                    node_prev.push_back(j > at ? (int32_t)(H.xoff[p] + pr.node_first_col[j - 1]) : -1);
                    for (int a = 0; a < bs; ++a) in_chain[col + a] = 1;
                }
                ch.col0 = H.node_col[ch.node_begin];
                ch.col_stride = 0;
                if (ch.N >= 2) {
                    const int32_t st0 = H.node_col[ch.node_begin + 1] - H.node_col[ch.node_begin];
                    bool even = st0 > 0;
                    for (int i = 2; i < ch.N && even; ++i)
                        even = (H.node_col[ch.node_begin + i] - H.node_col[ch.node_begin + i - 1]) == st0;
                    if (even) ch.col_stride = st0;
                }
                if (n_seg > 1) H.join_items.push_back(JoinItem{(int32_t)H.chains.size(), (int32_t)H.join_chains.size() - 1, sg, -1});
                H.chains.push_back(ch);
                at += len;
                if (sg + 1 < n_seg) {  // the separator after this segment: a Jacobi column (below)
                    H.join_sep_col.push_back((int32_t)(H.xoff[p] + pr.node_first_col[at]));
                    ++at;
                }
            }
        }
        // replicated problem: the chains come replica by replica (check_replication); replica k's chain c is
        // replica 0's chain c shifted by k * rep_n and uses its factors
        const size_t first = H.chain_owner.size(), mine = H.chains.size() - first;
        const size_t per_rep = H.rep > 1 ? mine / (size_t)H.rep : mine;
        for (size_t c = 0; c < mine; ++c) H.chain_owner.push_back((int32_t)(first + (per_rep ? c % per_rep : c)));
        const size_t jmine = H.join_chains.size() - jc_first, jper = H.rep > 1 ? jmine / (size_t)H.rep : jmine;
        for (size_t c = 0; c < jmine; ++c) H.join_chains[jc_first + c].owner = (int32_t)(jc_first + (jper ? c % jper : c));
    }
    for (size_t ci = 0; ci < H.chains.size(); ++ci) {
        const ChainDesc& a = H.chains[ci];
        const ChainDesc& o = H.chains[(size_t)H.chain_owner[ci]];
        if (a.N != o.N || a.prob != o.prob) throw std::runtime_error("replicated chains do not line up");
    }
    std::vector<char> node_owned(H.node_col.size(), 1);  // nodes whose blocks are looked up in K
    for (size_t ci = 0; ci < H.chains.size(); ++ci)
        if (H.chain_owner[ci] != (int32_t)ci)
            for (int i = 0; i < H.chains[ci].N; ++i) node_owned[(size_t)H.chains[ci].node_begin + i] = 0;
    H.node_prev_owned.resize(H.node_col.size());
    for (size_t g = 0; g < H.node_col.size(); ++g) H.node_prev_owned[g] = node_owned[g] ? node_prev[g] : -2;
    // K.val positions of the diagonal / sub-diagonal block entries of every chain node (host factorisation only: a device
    // backend finds them with a kernel once K's pattern is up)
    if (factor_on_host) {
    H.pos_diag.assign(H.node_col.size() * b2, -1);
    H.pos_sub.assign(H.node_col.size() * b2, -1);
    parallel_ranges((int64_t)H.node_col.size(), 2048, [&](int, int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g) {
            if (!node_owned[(size_t)g]) continue;
            const int32_t col = H.node_col[g], pc = node_prev[g];
            size_t o = (size_t)g * b2;
            for (int a = 0; a < bs; ++a)
                for (int bcol = 0; bcol < bs; ++bcol, ++o) {
                    H.pos_diag[o] = find_in_row(H.K, col + a, col + bcol);
                    if (pc >= 0) H.pos_sub[o] = find_in_row(H.K, col + a, pc + bcol);
                }
        }
    });
    }
    pt.mark("chain positions");
    // level layout (structure only) + storage
    H.fac_off.clear();
    H.fac_range.clear(); H.fac_range_H.clear();
    struct Layout { int N; std::vector<ChainLevelDesc> lv; size_t fac_size; int scr; int32_t map_off; };
    constexpr int kDeepLanes = 256;
    H.deep_ok = bs >= 1 && bs <= 3 && !H.chains.empty();
    size_t fac_total = H.fac.size();
    std::vector<Layout> layouts;  // the level structure depends only on (N, radix): one dry run per length
    size_t fac_total_H = 0;
    int64_t scratch_H = 0;
    H.chainsH = H.chains;
    for (size_t ci = 0; ci < H.chains.size(); ++ci) {
        ChainDesc& ch = H.chains[ci];
        const Layout* lay = nullptr;
        for (const auto& L : layouts)
            if (L.N == ch.N) lay = &L;
        if (!lay) {
            Layout L;
            L.N = ch.N;
            L.scr = 0;
            chain_level_layout(bs, H.radix, ch.N, L.lv, L.fac_size, L.scr);
            L.map_off = -1;
            if (bs >= 1 && bs <= 3 && chain_lane_plan(L.lv, kDeepLanes)) {  // (kPrecThreads - kPreRunLanes staging lanes)
                std::vector<int32_t> mp;
                chain_deep_map(bs, L.lv, 0, kDeepLanes, mp);
                L.map_off = (int32_t)H.deep_map.size();
                H.deep_map.insert(H.deep_map.end(), mp.begin(), mp.end());
            }
            layouts.push_back(std::move(L));
            lay = &layouts.back();
        }
        std::vector<ChainLevelDesc> lv = lay->lv;
        const size_t fac_size = lay->fac_size;
        const int scr = lay->scr;
        ch.n_levels = (int32_t)lv.size();
        if (ch.n_levels > 20) throw std::runtime_error("chain too long: more than 20 partition levels");
        const int64_t deep_sz = (int64_t)deep_padded_slots(bs) * kDeepLanes;
        ch.deep_map_off = lay->map_off;
        if (lay->map_off < 0) H.deep_ok = false;
        {   // the Newton matrix: every chain has factors of its own
            ChainDesc& cH = H.chainsH[ci];
            cH.deep_map_off = lay->map_off;
            cH.deep_off = (int32_t)H.deep_floats_H;
            H.deep_floats_H += deep_sz;
            cH.level_begin = (int32_t)H.levelsH.size();
            cH.n_levels = ch.n_levels;
            for (auto L : lv) {
                L.offR += (int64_t)fac_total_H; L.offS += (int64_t)fac_total_H; L.offB += (int64_t)fac_total_H;
                H.levelsH.push_back(L);
            }
            H.fac_range_H.push_back((int64_t)fac_total_H);
            H.fac_range_H.push_back((int64_t)(fac_total_H + fac_size));
            fac_total_H += fac_size;
            cH.scratch_off = (int32_t)scratch_H;
            cH.scratch_nodes = scr;
            scratch_H += scr;
        }
        if (H.chain_owner[ci] == (int32_t)ci) {
            ch.level_begin = (int32_t)H.levels.size();
            const int64_t dbl_base = (int64_t)fac_total;
            for (auto& L : lv) {
                L.offR += dbl_base;
                L.offS += dbl_base;
                L.offB += dbl_base;
                H.levels.push_back(L);
            }
            H.fac_off.push_back(dbl_base);
            H.fac_range.push_back(dbl_base);
            H.fac_range.push_back(dbl_base + (int64_t)fac_size);
            fac_total += fac_size;
            ch.deep_off = (int32_t)H.deep_floats;
            H.deep_floats += deep_sz;
        } else {  // (owners precede their replicas)
            ch.level_begin = H.chains[(size_t)H.chain_owner[ci]].level_begin;
            H.fac_off.push_back(H.fac_off[(size_t)H.chain_owner[ci]]);
            H.fac_range.push_back(H.fac_range[2 * (size_t)H.chain_owner[ci]]);
            H.fac_range.push_back(H.fac_range[2 * (size_t)H.chain_owner[ci] + 1]);
            ch.deep_off = H.chains[(size_t)H.chain_owner[ci]].deep_off;
        }
        ch.scratch_off = (int32_t)H.scratch_nodes;
        ch.scratch_nodes = scr;
        H.scratch_nodes += scr;
        H.max_chain_scratch = std::max(H.max_chain_scratch, scr);
    }
    H.fac_doubles_H = fac_total_H;
    H.scratch_nodes = std::max(H.scratch_nodes, scratch_H);
    // (the factors themselves live on the device when the backend derives them there)
    H.fac.assign(H.factor_on_host ? fac_total : 0, 0.0);
    H.fac_doubles = fac_total;
    pt.mark("level layout");
    // Jacobi columns + work list (problem-major: chains, then Jacobi blocks)
    std::vector<int32_t> chain_item, sep_of_col;  // (segmented chains: join item of a chain, separator of a column)
    std::vector<char> sep_comp;
    if (!H.join_chains.empty()) {
        chain_item.assign(H.chains.size(), -1);
        for (size_t i = 0; i < H.join_items.size(); ++i) chain_item[(size_t)H.join_items[i].chain] = (int32_t)i;
        sep_of_col.assign((size_t)H.n_tot, -1);
        sep_comp.assign((size_t)H.n_tot, 0);
        for (size_t sp = 0; sp < H.join_sep_col.size(); ++sp)
            for (int a = 0; a < bs; ++a) { sep_of_col[(size_t)H.join_sep_col[sp] + a] = (int32_t)sp; sep_comp[(size_t)H.join_sep_col[sp] + a] = (char)a; }
        H.join_sep_diag.assign(H.join_sep_col.size() * (size_t)bs, -1);
    }
    size_t ci = 0;
    for (int p = 0; p < count; ++p) {
        const size_t c_first = ci;
        while (ci < H.chains.size() && H.chains[ci].prob == p) {
            if (H.chain_owner[ci] == (int32_t)ci) H.factor_work.push_back(PrecWork{0, (int32_t)ci, 0, p});
            ++ci;
        }
        // (Measured and not kept: dealing a replicated problem's work items in groups of 8 robots, replica after replica,
        //  so that the chains sharing one factor set sit 8 items apart -- on one XCD's L2 under round-robin placement:
        //  no change, 61.6 vs 62.5 us at 16 problems.  The chain kernel is bound by its dependent phases at one
        //  workgroup per CU, not by where its factors come from.)
        for (size_t c = c_first; c < ci; ++c) {
            if (!chain_item.empty() && chain_item[c] >= 0) H.join_items[(size_t)chain_item[c]].work = (int32_t)H.prec_work.size();
            H.prec_work.push_back(PrecWork{0, (int32_t)c, 0, p});
        }
        const size_t d_first = H.diag_cols.size();
        for (int64_t c = H.xoff[p]; c < H.xoff[p + 1]; ++c)
            if (!in_chain[c]) {
                if (!sep_of_col.empty() && sep_of_col[(size_t)c] >= 0) H.join_sep_diag[(size_t)sep_of_col[(size_t)c] * bs + (size_t)sep_comp[(size_t)c]] = (int32_t)H.diag_cols.size();
                H.diag_cols.push_back((int32_t)c);
                // (a column of replica k finds its diagonal in replica 0's row)
                int64_t c0_ = c;
                if (H.rep > 1) {
                    const int64_t local = c - H.xoff[p], nr = H.rep_n[(size_t)p];
                    if (local < (int64_t)H.rep * nr) c0_ = H.xoff[p] + local % nr;
                }
                H.diag_row0.push_back((int32_t)c0_);
                if (factor_on_host) H.diag_kpos.push_back(find_in_row(H.K, c0_, (int32_t)c0_));
            }
        // (one Jacobi item = what a 512-thread workgroup requests in one batch of loads: 6 entries per lane)
        constexpr size_t kJacobiItem = 3072;
        for (size_t e = d_first; e < H.diag_cols.size(); e += kJacobiItem) {
            const PrecWork jw{1, (int32_t)e, (int32_t)std::min<size_t>(kJacobiItem, H.diag_cols.size() - e), p};
            H.prec_work.push_back(jw);
            H.factor_work.push_back(jw);
        }
        H.prec_part_ptr[p + 1] = (int32_t)H.prec_work.size();
    }
    H.dinv.assign(H.diag_cols.size(), 1.0);
    // algorithmic bytes of one K-apply per problem: 12 B per nonzero (fp64 value
    // + int32 column), 4 B per row pointer, vector p read once, w written once
    H.kkt_bytes.assign(count, 0.0);
    if (!lite) fill_kkt_bytes(H);
    pt.mark("jacobi + work list");
    for (int p = 0; p < count; ++p) refresh_rho(H, p);
    pt.mark("refresh_rho (factor)");
}

}  // namespace score
