// score_hip.hip -- MI355X (gfx950) backend of the SCORE conic solver + C ABI.
//
// One ADMM ("SOCP") iteration on the device, all problems of the batch in
// lock-step, per-problem scalars kept in device memory (no host round trip):
//
//   k_spmv<RHS>    r  = sigma x - q + [-K | A'] [xt ; u]       CSR-stream SpMV
//   k_prec<INIT>   z  = M^-1 r ; p = z ; partial r'z           chain + Jacobi
//   repeat cg_iters times:
//     k_spmv<KP>   w  = K p ; partial p'w                      CSR-stream SpMV
//     k_prec<STEP> a = r'z / p'w ; xt += a p ; r -= a w ; z = M^-1 r ; partial r'z
//     k_spmv<KPB>  p  = z + (r'z_new / r'z_old) p fused into the next w = K p
//   (the last CG iteration replaces STEP/pupdate by k_xupdate:
//                  xt += a p ; x = alpha xt + (1 - alpha) x)
//   k_cone         v = alpha (b - A xt) + (1 - alpha) s ; s = Proj_K(v - y/rho) ;
//                  y += rho (s - v) ; u = rho (b - s) - y      one cone per lane
//
// Kernels and their design notes: score_kernels.hpp.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstring>
#include <malloc.h>
#include <mutex>
#include <string>
#include <memory>
#include <optional>
#include <vector>

#include "score_driver.hpp"
#include <future>
#include "score_assemble.hpp"
#include "score_gn_kernels.hpp"
#include "score_round.hpp"
#include "score_kernels.hpp"
#include "score_polish.hpp"
#include "score_polish_device.hpp"
#include "score_setup_device.hpp"
#include "score_prec_wave.hpp"
#include "score_join.hpp"
#include "score_link.hpp"
#include "score_generate.hpp"

namespace {

using namespace score;

thread_local std::string g_err;

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));  \
    } while (0)

// ---- how a host thread waits for the device (round 6) ----
// A driver thread alternates queueing launches and waiting for small results in host-mapped memory.  Alone in the process it
// spins (a result is picked up within a fraction of a microsecond: the latency of ONE default solve).  As soon as several
// threads are inside solves at once (the lock-step groups of a Monte-Carlo sweep: 4 per rank) or several ranks share the node,
// the waits go ECONOMY: a short spin, then sleeps of kEconomySleepNs with the thread's timer slack lowered to 1 us -- a waiting
// thread costs no CPU, the Newton PCG is queued deeper ahead of the device (kEconomyDepth instead of 3) so that the device does
// not run dry while its driver sleeps.  Rounds 4-5 spun, then yielded: with idle CPUs around a yield returns at once, and every
// driver thread burnt a full CPU for the length of its solves -- 2.6-2.8 ms of host CPU per problem (BENCH_r05), which caps eight
// ranks on the 16 CPUs the boxes grant at a third of the GPUs' capacity.
// (HostWaitStats, wait_policy(), economy_waits(): score_host.hpp -- the host thread teams follow the same policy)
inline void economy_sleep() {
    static thread_local bool slack_set = false;
    if (!slack_set) { prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); slack_set = true; }  // (default 50 us: a 25 us sleep would take 75)
    struct timespec ts = {0, kEconomySleepNs};
    const auto t0 = std::chrono::steady_clock::now();
    nanosleep(&ts, nullptr);
    wait_stats().sleep_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    wait_stats().sleeps.fetch_add(1, std::memory_order_relaxed);
}
struct ActiveSolve {  // (every entry point that creates, solves, reads or destroys a handle: two of them at once = economy waits)
    ActiveSolve() { wait_stats().active_solves.fetch_add(1, std::memory_order_relaxed); }
    ~ActiveSolve() { wait_stats().active_solves.fetch_sub(1, std::memory_order_relaxed); }
};
// hipStreamSynchronize spins inside the runtime (the default scheduling policy): with economy waits the stream is polled with
// sleeps in between instead -- setup, read-back and teardown wait for the device as cheaply as the solves do
inline hipError_t sync_stream(hipStream_t s) {
    if (!economy_waits()) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 10.0) economy_sleep();
    }
}

// Every entry point that takes a handle runs with the handle's device current and restores the
// caller's device on exit: a host thread that never called hipSetDevice (a pool worker on a rank
// with LOCAL_RANK > 0) would otherwise capture / launch / allocate on device 0 while the handle's
// stream lives on device N.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) {
            HIP_CHECK(hipSetDevice(dev));
            switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---------------------------------------------------------------------------
// backend
// ---------------------------------------------------------------------------
// Uploads go through the calling handle's stream (never the legacy stream): several host
// threads may drive different handles while one of them is capturing a graph.
thread_local hipStream_t tl_copy_stream = nullptr;

// Handles come and go at a high rate in Monte-Carlo use (one per lock-step group of trials), and
// hipMalloc / hipFree / hipHostMalloc / hipHostFree cost milliseconds each (hipFree synchronises the
// device): freed blocks are parked in a small process-wide cache, per device and size class (powers
// of two), and handed to the next handle that asks.  Bounded: at most kMaxCachedBytes stay parked.
struct BlockCache {
    struct Block { void* p; size_t bytes; int dev; bool host; unsigned long long stamp; };
    unsigned long long clock = 0;
    std::mutex mu;
    std::vector<Block> free_blocks;
    size_t cached[2] = {0, 0};  // parked bytes: device blocks, pinned host blocks
    // Caps of the parked bytes, per kind.  Device blocks: an eighth of the device's memory, at most 32 GiB (round 5: the 4 GiB
    // both kinds shared until then were filled by the handles of one Monte-Carlo sweep -- 8 live handles of 0.3-0.5 GB each in
    // a dozen size classes -- and every other sweep paid for evictions: hipFree, a device synchronisation, 5-15 ms inside
    // score_destroy; sweeps of 30 ms became 40-47).  Pinned host blocks: 4 GiB.  SCORE_CACHE_MB: both caps (0 = park nothing).
    static size_t max_cached_bytes(bool host) {
        static const long env_mb = [] { const char* e = std::getenv("SCORE_CACHE_MB"); return e ? std::max(0L, std::atol(e)) : -1L; }();
        if (env_mb >= 0) return (size_t)env_mb << 20;
        if (host) return (size_t)4 << 30;
        static const size_t dev_cap = [] {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return (size_t)4 << 30; }
            return std::min<size_t>(tot / 8, (size_t)32 << 30);
        }();
        return dev_cap;
    }
    // size classes: powers of two up to 1 MiB, multiples of 2 MiB above (a 65 MiB request parks 66 MiB, not 128)
    static size_t round_up(size_t b) {
        if (b <= ((size_t)1 << 20)) {
            size_t r = 4096;
            while (r < b) r <<= 1;
            return r;
        }
        const size_t g = (size_t)2 << 20;
        return (b + g - 1) / g * g;
    }
    static hipError_t raw_alloc(void** p, size_t bytes, bool host) {
        return host ? hipHostMalloc(p, bytes, hipHostMallocMapped) : hipMalloc(p, bytes);
    }
    void* take(size_t& bytes, int dev, bool host) {
        bytes = round_up(bytes);
        {
            std::lock_guard<std::mutex> lk(mu);
            // same class first; failing that, a parked block of the same kind at most half as large again
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); ++i) {
                const Block& b = free_blocks[i];
                if (b.dev != dev || b.host != host || b.bytes < bytes || b.bytes > bytes + bytes / 2) continue;
                if (best == free_blocks.size() || b.bytes < free_blocks[best].bytes) best = i;
            }
            if (best != free_blocks.size()) {
                void* p = free_blocks[best].p;
                bytes = free_blocks[best].bytes;
                cached[host ? 1 : 0] -= bytes;
                free_blocks[best] = free_blocks.back();
                free_blocks.pop_back();
                return p;
            }
        }
        void* p = nullptr;
        const auto t_raw = std::chrono::steady_clock::now();
        hipError_t e = raw_alloc(&p, bytes, host);
        if (trace_on("cache"))
            std::fprintf(stderr, "[score cache] %s of %zu bytes: %.2f ms\n", host ? "hipHostMalloc" : "hipMalloc", bytes,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_raw).count());
        if (e != hipSuccess) {
            // out of device or locked memory: what this cache holds back (other size classes, devices, kinds) is the
            // first thing to give up -- release every parked block and try once more
            (void)hipGetLastError();
            trim();
            e = raw_alloc(&p, bytes, host);
        }
        if (e != hipSuccess)
            throw std::runtime_error(std::string(host ? "hipHostMalloc" : "hipMalloc") + " of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return p;
    }
    // A block that does not fit under the cap makes room for itself: the blocks parked LONGEST ago go (round 5: a cache filled
    // to its cap by one large handle -- 16 headline problems in one -- used to refuse every later block, and every later create
    // and destroy paid for raw hipMalloc / hipFree, the latter a device synchronisation: 64 fresh graphs 30 -> 38 ms per sweep,
    // solve_score() on the headline graph 15 -> 24 ms in a process that had held such a handle).
    void give(void* p, size_t bytes, int dev, bool host) {
        if (!p) return;
        std::vector<Block> evict;
        bool parked = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            const size_t cap = max_cached_bytes(host);
            size_t& held = cached[host ? 1 : 0];
            if (bytes <= cap) {
                while (held + bytes > cap) {  // the blocks of this kind parked longest ago make room
                    size_t oldest = free_blocks.size();
                    for (size_t i = 0; i < free_blocks.size(); ++i)
                        if (free_blocks[i].host == host && (oldest == free_blocks.size() || free_blocks[i].stamp < free_blocks[oldest].stamp)) oldest = i;
                    if (oldest == free_blocks.size()) break;
                    evict.push_back(free_blocks[oldest]);
                    held -= free_blocks[oldest].bytes;
                    free_blocks[oldest] = free_blocks.back();
                    free_blocks.pop_back();
                }
                free_blocks.push_back(Block{p, bytes, dev, host, ++clock});
                held += bytes;
                parked = true;
            }
        }
        for (const Block& b : evict) { if (b.host) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
        if (!parked) { if (host) (void)hipHostFree(p); else (void)hipFree(p); }
    }
    // release every parked block (score_trim_caches; also the retry path of take()); returns the bytes freed
    size_t trim() {
        std::vector<Block> all;
        size_t freed = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(free_blocks);
            freed = cached[0] + cached[1];
            cached[0] = cached[1] = 0;
        }
        for (const Block& b : all) {
            if (b.host) (void)hipHostFree(b.p); else (void)hipFree(b.p);
        }
        return freed;
    }
    ~BlockCache() {  // process exit: the runtime may already be gone, leave the blocks to it
    }
};
// idle HIP streams, per device: hipStreamCreate / hipStreamDestroy cost 3-4 ms each
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
    hipStream_t take(int dev) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].first == dev) {
                    hipStream_t s = idle[i].second;
                    idle[i] = idle.back();
                    idle.pop_back();
                    return s;
                }
        }
        hipStream_t s = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return s;
    }
    void give(int dev, hipStream_t s) {
        if (!s) return;
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() < 64) idle.emplace_back(dev, s);
        else (void)hipStreamDestroy(s);
    }
    void trim() {
        std::vector<std::pair<int, hipStream_t>> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(idle);
        }
        for (auto& e : all) (void)hipStreamDestroy(e.second);
    }
};
inline StreamPool& stream_pool() {
    static StreamPool* p = new StreamPool();  // leaked on purpose, like the block cache
    return *p;
}

inline BlockCache& block_cache() {
    static BlockCache* c = new BlockCache();  // intentionally leaked (see ~BlockCache)
    return *c;
}

// Device memory of one handle comes from a few large allocations: ~70 hipMalloc / hipFree pairs per
// handle cost ~20 ms (hipFree synchronises the device), a handful cost ~1 ms.
struct DevArena {
    int dev = 0;
    std::vector<size_t> chunk_bytes;
    std::vector<void*> chunks;
    char* cur = nullptr;
    size_t left = 0, next_chunk = (size_t)8 << 20;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) {
            const size_t sz = std::max(bytes, next_chunk);
            next_chunk = std::min<size_t>(next_chunk * 2, (size_t)64 << 20);
            size_t got = sz;
            void* p = block_cache().take(got, dev, false);
            chunks.push_back(p);
            chunk_bytes.push_back(got);
            cur = (char*)p;
            left = got;
        }
        void* r = cur;
        cur += bytes;
        left -= bytes;
        return r;
    }
    // (the owner has synchronised its stream: nothing in flight touches these blocks)
    void release_all() {
        for (size_t i = 0; i < chunks.size(); ++i) block_cache().give(chunks[i], chunk_bytes[i], dev, false);
        chunks.clear(); chunk_bytes.clear();
        cur = nullptr; left = 0; next_chunk = (size_t)8 << 20;
    }
    ~DevArena() { release_all(); }
};
thread_local DevArena* tl_arena = nullptr;  // set while a handle is being initialised on this thread

// Uploads of a handle under construction go through pinned staging when they are SMALL.  hipMemcpyAsync from pageable
// memory blocks the caller for the whole transfer and every array is followed by a stream synchronisation: a create issues
// ~80 uploads of which ~65 are small tables (tile records, cone tables, per-problem ranges), 15 us each in calls and waits
// whatever their size -- 1 ms of a 2 ms create of a 100-pose graph, 0.9 ms of a headline create.  A small source is copied
// into a pinned chunk, its transfer queued on the handle's stream, and the caller moves on (the source may die at once); the
// chunks go back to the block cache after ONE synchronisation at the end of the setup.  Large arrays keep the pageable
// path: their cost is bandwidth, and staging them (copy teams of host threads into 32 MB pinned chunks) measured no gain on
// the headline create.  Measured (round 4): create of 1 x 100 poses 2.11 -> 1.17 ms, 1 x 500: 2.40 -> 1.48, 4 x 1000: 4.06 ->
// 2.82, 20 x 1000: 9.2 -> 8.6-9.0 ms.  Round 5: EVERYTHING goes through pinned memory of the library's own -- see staged_d2h
// below for why no pageable pointer is ever handed to the runtime (the size-limit switches of rounds 4-5 are gone: TRIED.md).
inline size_t stage_limit_bytes() { return ~(size_t)0; }
struct StageArena {
    int dev = 0;
    std::vector<size_t> chunk_bytes;
    std::vector<void*> chunks;
    size_t at = 0;        // chunk in use
    char* cur = nullptr;
    size_t left = 0;
    size_t inflight = 0;  // bytes staged since the last synchronisation
    static constexpr size_t kMaxInflight = (size_t)192 << 20;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) {
            // the next parked chunk if it is large enough, a new one otherwise
            size_t k = chunks.empty() ? 0 : at + 1;
            while (k < chunks.size() && chunk_bytes[k] < bytes) ++k;
            if (k >= chunks.size()) {
                size_t got = std::max(bytes, (size_t)8 << 20);  // (a pinned allocation costs ~0.2 ms per MB when the cache misses)
                void* p = block_cache().take(got, dev, true);
                chunks.push_back(p);
                chunk_bytes.push_back(got);
                k = chunks.size() - 1;
            }
            at = k;
            cur = (char*)chunks[k];
            left = chunk_bytes[k];
        }
        void* r = cur;
        cur += bytes;
        left -= bytes;
        return r;
    }
    // copy `bytes` from src into a pinned slot and queue its transfer to `dst` on `st`; false: too large, the caller
    // takes its own path
    bool upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
        if (!bytes) return true;
        if (bytes > stage_limit_bytes()) return false;
        if (inflight + bytes > kMaxInflight && inflight > 0) {  // everything queued so far has to leave its slots: start over
            flush();
            HIP_CHECK(sync_stream(st));
            inflight = 0; at = 0;
            cur = chunks.empty() ? nullptr : (char*)chunks[0];
            left = chunks.empty() ? 0 : chunk_bytes[0];
        }
        char* pin = (char*)take(bytes);
        if (bytes >= ((size_t)1 << 20))
            score::parallel_ranges((int64_t)bytes, (int64_t)1 << 19, [&](int, int64_t b0, int64_t b1) { std::memcpy(pin + b0, (const char*)src + b0, (size_t)(b1 - b0)); });
        else
            std::memcpy(pin, src, bytes);
        inflight += bytes;
        if (batch_depth > 0) {
            // Inside an UploadBatch: tables that follow each other go up in ONE transfer when both their pinned slots and their
            // device buffers are neighbours (both arenas hand out 256-byte-aligned pieces in order, so a run of `x.upload(v)` calls
            // is exactly that).  The alignment gap between two of them is sent as zeros: it belongs to no buffer (or is the zeroed
            // padding of upload_padded).  Nothing that reads the tables may be launched before the batch closes.
            const size_t span = (pend_bytes + 255) & ~(size_t)255;
            if (pend_bytes && pend_st == st && pend_dst + span == (char*)dst && pend_pin + span == pin) {
                if (span > pend_bytes) std::memset(pend_pin + pend_bytes, 0, span - pend_bytes);
                pend_bytes = span + bytes;
            } else {
                flush();
                pend_dst = (char*)dst; pend_pin = pin; pend_bytes = bytes; pend_st = st;
            }
            return true;
        }
        HIP_CHECK(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, st));
        return true;
    }
    int batch_depth = 0;
    char* pend_dst = nullptr; char* pend_pin = nullptr; size_t pend_bytes = 0; hipStream_t pend_st = nullptr;
    void flush() {
        if (!pend_bytes) return;
        const size_t nb = pend_bytes;
        pend_bytes = 0;
        HIP_CHECK(hipMemcpyAsync(pend_dst, pend_pin, nb, hipMemcpyHostToDevice, pend_st));
    }
    // (the caller has synchronised the stream the transfers were queued on)
    void release() {
        for (size_t i = 0; i < chunks.size(); ++i) block_cache().give(chunks[i], chunk_bytes[i], dev, true);
        chunks.clear(); chunk_bytes.clear();
        cur = nullptr; left = 0; at = 0; inflight = 0;
    }
    ~StageArena() { release(); }
};
thread_local StageArena* tl_stage = nullptr;  // set while a handle's setup uploads on this thread
// Small fills likewise (a create issued ~40 runtime fills of ~5 us each: padding entries, counters, sentinels): inside a batch they
// are listed and written by ONE launch of k_fill_list when the batch closes (before the merged transfers: "zero, then upload into
// it" keeps its order).  Fills of a megabyte or more are issued at once (their buffers are nobody's upload target).
constexpr int kFillListMax = 24;
struct FillSeg { uint32_t* p; unsigned long long words; uint32_t pattern; uint32_t blk0; };
struct FillList { FillSeg s[kFillListMax]; int count; };
constexpr unsigned kFillWordsPerBlock = 256 * 16;
__global__ __launch_bounds__(256) void k_fill_list(FillList L) {
    int q = 0;
    while (q + 1 < L.count && L.s[q + 1].blk0 <= blockIdx.x) ++q;
    const FillSeg sg = L.s[q];
    const unsigned long long w0 = (unsigned long long)(blockIdx.x - sg.blk0) * kFillWordsPerBlock;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const unsigned long long w = w0 + (unsigned long long)u * 256 + threadIdx.x;
        if (w < sg.words) sg.p[w] = sg.pattern;
    }
}
struct FillBatchState {
    int depth = 0;
    hipStream_t st = nullptr;
    FillList L{};
    unsigned blocks = 0;
    void flush() {
        if (!L.count) return;
        const unsigned g = blocks;
        const FillList out = L;
        L.count = 0; blocks = 0;
        hipLaunchKernelGGL(k_fill_list, dim3(g), dim3(256), 0, st, out);
        HIP_CHECK(hipGetLastError());
    }
    void add(void* p, size_t bytes, uint32_t pattern, hipStream_t s) {
        if (L.count && (s != st || L.count == kFillListMax)) flush();
        st = s;
        FillSeg& sg = L.s[L.count++];
        sg.p = (uint32_t*)p; sg.words = bytes / 4; sg.pattern = pattern; sg.blk0 = blocks;
        blocks += (unsigned)((sg.words + kFillWordsPerBlock - 1) / kFillWordsPerBlock);
    }
};
thread_local FillBatchState tl_fills;
// every fill of the library goes through these two (pattern: a 32-bit word)
inline void fill_words_async(void* p, uint32_t pattern, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    if (tl_fills.depth > 0 && bytes < ((size_t)1 << 20) && bytes % 4 == 0 && ((uintptr_t)p % 4) == 0) { tl_fills.add(p, bytes, pattern, st); return; }
    if (pattern == 0) HIP_CHECK(hipMemsetAsync(p, 0, bytes, st));
    else if (bytes % 4 == 0) HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)p, (int)pattern, bytes / 4, st));
    else throw std::runtime_error("fill_words_async: a patterned fill of a size that is no multiple of 4");
}
inline void fill_zero_async(void* p, size_t bytes, hipStream_t st) { fill_words_async(p, 0u, bytes, st); }
// A run of small uploads and fills with no launch in between (round 6: a create issued ~60 transfers of 2-5 us each, most of them
// tables that follow each other): opened around such a run, closed (= the listed fills written, the merged transfers queued)
// before anything reads the buffers.
struct UploadBatch {
    StageArena* a;
    UploadBatch() : a(tl_stage) { if (a) ++a->batch_depth; ++tl_fills.depth; }
    ~UploadBatch() {
        if (--tl_fills.depth == 0) {
            try { tl_fills.flush(); } catch (...) { tl_fills.L.count = 0; tl_fills.blocks = 0; }
        }
        if (!a) return;
        if (--a->batch_depth == 0) {
            try { a->flush(); } catch (...) { a->pend_bytes = 0; }  // (a failed transfer surfaces at the next checked call on the stream)
        }
    }
    UploadBatch(const UploadBatch&) = delete;
    UploadBatch& operator=(const UploadBatch&) = delete;
};

// Host <-> device copies never hand PAGEABLE memory to the runtime.  For a transfer of a megabyte or more the runtime registers
// the caller's pages with the driver; when the caller later frees that memory (NumPy arrays of the previous solve, the vectors
// of a model construction) the change of the address space evicts EVERY queue of the process until the driver has restored
// them -- measured (round 5, 20 x 5000 poses: 12 MB of x, y, s per solve): the first kernel after such a free started 14-37 ms
// late, whichever handle or thread it belonged to; solves 47 / 46 / 21 / 21 ms became 32 / 21 / 21 / 21 with the download
// staged, 21 / 21 / 21 / 21 with the uploads of score_create staged as well.  Both directions therefore go through pinned
// blocks of the library's own (block cache), in pieces of at most 16 MB; synchronous.
inline void staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    size_t got = std::min<size_t>(bytes, (size_t)16 << 20);
    char* pin = (char*)block_cache().take(got, dev, true);
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += got) {
        const size_t nb = std::min(got, bytes - off);
        if (nb >= ((size_t)1 << 20))
            score::parallel_ranges((int64_t)nb, (int64_t)1 << 19, [&](int, int64_t b0, int64_t b1) { std::memcpy(pin + b0, (const char*)src + off + b0, (size_t)(b1 - b0)); });
        else
            std::memcpy(pin, (const char*)src + off, nb);
        e = hipMemcpyAsync((char*)dst + off, pin, nb, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = sync_stream(st);
    }
    block_cache().give(pin, got, dev, true);
    HIP_CHECK(e);
}
inline void staged_d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    size_t got = std::min<size_t>(bytes, (size_t)16 << 20);
    char* pin = (char*)block_cache().take(got, dev, true);
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += got) {
        const size_t nb = std::min(got, bytes - off);
        e = hipMemcpyAsync(pin, (const char*)src + off, nb, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = sync_stream(st);
        if (e != hipSuccess) break;
        if (nb >= ((size_t)1 << 20))
            score::parallel_ranges((int64_t)nb, (int64_t)1 << 19, [&](int, int64_t b0, int64_t b1) { std::memcpy((char*)dst + off + b0, pin + b0, (size_t)(b1 - b0)); });
        else
            std::memcpy((char*)dst + off, pin, nb);
    }
    block_cache().give(pin, got, dev, true);
    HIP_CHECK(e);
}

__global__ __launch_bounds__(256) void k_zero16(uint4* __restrict__ p, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}

template <class T>
struct DevBuf {
    T* d = nullptr;
    size_t n = 0;
    bool owned = false;  // allocated with hipMalloc (not from the handle's arena)
    void alloc(size_t count) {
        release();
        n = count;
        const size_t bytes = std::max<size_t>(1, count) * sizeof(T);
        if (tl_arena) {
            d = (T*)tl_arena->take(bytes);
            owned = false;
        } else {
            HIP_CHECK(hipMalloc((void**)&d, bytes));
            owned = true;
        }
    }
    void upload(const std::vector<T>& h) {
        if (h.size() != n || !d) alloc(h.size());
        if (!h.empty()) {
            if (tl_stage && tl_stage->upload(d, h.data(), h.size() * sizeof(T), tl_copy_stream)) return;
            staged_h2d(d, h.data(), h.size() * sizeof(T), tl_copy_stream);
        }
    }
    // (staged: the source may go away when the call returns)
    void upload_from(const T* src, size_t count) {
        if (count != n || !d) alloc(count);
        if (count && tl_stage && tl_stage->upload(d, src, count * sizeof(T), tl_copy_stream)) return;
        if (count) staged_h2d(d, src, count * sizeof(T), tl_copy_stream);
    }
    // upload into an allocation with `pad` extra zeroed elements at the end
    void upload_padded(const std::vector<T>& h, size_t pad) {
        if (h.size() + pad != n || !d) alloc(h.size() + pad);
        fill_zero_async(d + h.size(), pad * sizeof(T), tl_copy_stream);
        if (tl_stage && tl_stage->upload(d, h.data(), h.size() * sizeof(T), tl_copy_stream)) return;
        if (!h.empty()) staged_h2d(d, h.data(), h.size() * sizeof(T), tl_copy_stream);
        HIP_CHECK(sync_stream(tl_copy_stream));
    }
    // copy into the existing allocation (which may be larger: padded), never re-allocating
    void upload_into(const std::vector<T>& h) {
        if (!d || h.size() > n) throw std::runtime_error("upload_into: buffer too small");
        if (!h.empty()) staged_h2d(d, h.data(), h.size() * sizeof(T), tl_copy_stream);
    }
    // (a kernel of the library's own for large blocks: hipMemsetAsync took 22-25 ms for the 57 MB iterate block of a
    //  20 x 5000-pose problem in the first two solves of a handle and 0.03 ms afterwards -- measured, round 5)
    void zero(hipStream_t st) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= ((size_t)1 << 20) && bytes % 16 == 0 && ((uintptr_t)d % 16) == 0) {
            const size_t n16 = bytes / 16;
            const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 4096);
            hipLaunchKernelGGL(k_zero16, dim3(grid), dim3(256), 0, st, (uint4*)d, n16);
        } else if (n) fill_zero_async(d, bytes, st);
    }
    // a window into somebody else's allocation
    void view(T* ptr, size_t count) {
        release();
        d = ptr;
        n = count;
        owned = false;
    }
    void release() {
        if (d && owned) (void)hipFree(d);
        d = nullptr;
        n = 0;
        owned = false;
    }
    ~DevBuf() { release(); }
};

// Several zero-initialised buffers as ONE allocation and ONE fill (a fill is a 4-5 us dispatch -- two or three when its range
// is not aligned --, and a create had a hundred of them): add() the buffers, commit() allocates from the current arena
// (tl_arena) and queues the fill on the stream.
struct ZeroGroup {
    struct Item { void** d; size_t* n; size_t count, bytes, off; };
    std::vector<Item> items;
    size_t total = 0;
    template <class T>
    void add(DevBuf<T>& b, size_t count) {
        b.release();
        const size_t bytes = (std::max<size_t>(1, count) * sizeof(T) + 255) & ~(size_t)255;
        items.push_back(Item{(void**)&b.d, &b.n, count, bytes, total});
        total += bytes;
    }
    void commit(hipStream_t st) {
        if (!total) return;
        if (!tl_arena) throw std::runtime_error("ZeroGroup: no arena");
        char* base = (char*)tl_arena->take(total);
        for (const Item& it : items) { *it.d = base + it.off; *it.n = it.count; }
        fill_zero_async(base, total, st);
        items.clear();
        total = 0;
    }
};

// SCORE_NO_LONG_SPIN=1: the split long rows of every matrix go through the ticket path (CsrDev::long_spin)
inline bool long_spin_enabled() {
    static const bool on = std::getenv("SCORE_NO_LONG_SPIN") == nullptr;
    return on;
}

struct CsrBufs {
    DevBuf<int32_t> ptr, col, first_row, blk_prob, blk_rs, split;
    DevBuf<int4> blk_meta, blk_long;
    DevBuf<double> long_part;               // split long rows: kLongVals sums per segment
    DevBuf<unsigned long long> long_cnt;    // ... and one arrival counter per row (never reset: a launch adds `segments`)
    void alloc_long(int n_long, int n_slots) {
        long_part.alloc((size_t)std::max(1, n_slots) * kLongVals);
        long_cnt.alloc((size_t)std::max(1, n_long));
        fill_zero_async(long_cnt.d, long_cnt.n * sizeof(unsigned long long), tl_copy_stream);
        // (every slot starts as "not published": the polling mode of the split rows -- CsrDev::long_spin -- reads it that way)
        fill_words_async(long_part.d, (uint32_t)kLongSentinel32, long_part.n * sizeof(double), tl_copy_stream);
    }
    int spin_max_tiles = 0;  // long_spin when the matrix has at most this many tiles (0: never); set where the matrix is made
    DevBuf<double> val;
    int nblocks = 0;
    int rep = 1;     // right-hand sides per row of the replicated blocks (HostSystem::rep), 1 = plain rows only
    int unroll = kUnroll;  // nonzeros per lane of a tile (HostSystem::tile_nnz / 256)
    int rs_in = 0;   // replicated blocks: operand stride between replicas (0 = the block's own replica stride)
    // the tile tables of a matrix whose pattern is already on the device (ptr, col filled by kernels: the Newton matrix built
    // by score_polish_device.hpp); M.ptr is its host copy
    void adopt_tiles(const Csr& M, const RowBlocks& rb) {
        UploadBatch ub;
        first_row.upload(rb.first_row);
        blk_prob.upload(rb.prob);
        blk_rs.upload(rb.rs);
        std::vector<int4> meta(rb.nb()), lg(rb.nb());
        for (int b = 0; b < rb.nb(); ++b) {
            const bool seg = rb.kbeg[b] >= 0;
            meta[b] = make_int4(rb.first_row[b], rb.end_row[b], seg ? rb.kbeg[b] : M.ptr[rb.first_row[b]], seg ? rb.kend[b] : M.ptr[rb.end_row[b]]);
            lg[b] = make_int4(rb.lfirst[b], rb.lseg[b], rb.lbase[b], rb.lid[b]);
        }
        blk_meta.upload(meta);
        blk_long.upload(lg);
        alloc_long(rb.n_long, rb.n_long_slots);
        nblocks = rb.nb();
    }
    // values = false: the value array is only allocated (zeroed); a kernel fills it
    // columns = false: the column array is only allocated as well (pad zeroed); a kernel fills it
    void upload(const Csr& M, const RowBlocks& rb, const std::vector<int32_t>* sp = nullptr, bool values = true, bool columns = true) {
        UploadBatch ub;
        ptr.upload(M.ptr);
        // The SpMV issues its loads unconditionally on clamped indices; for an empty tile at the
        // very end of the matrix the clamp lands one past the last nonzero.  Pad with harmless
        // entries (column 0, value 0) so that such a read stays in bounds and gathers x[0].
        if (columns) {
            col.upload_padded(M.col, 64);
        } else {
            col.alloc(M.col.size() + 64);
            fill_zero_async(col.d + M.col.size(), 64 * sizeof(int32_t), tl_copy_stream);
        }
        if (values) {
            val.upload_padded(M.val, 64);
        } else {
            val.alloc(M.col.size() + 64);
            fill_zero_async(val.d, val.n * sizeof(double), tl_copy_stream);
        }
        first_row.upload(rb.first_row);
        blk_prob.upload(rb.prob);
        blk_rs.upload(rb.rs);
        std::vector<int4> meta(rb.nb()), lg(rb.nb());
        for (int b = 0; b < rb.nb(); ++b) {
            const bool seg = rb.kbeg[b] >= 0;  // a segment of a split long row (kLongSeg)
            meta[b] = make_int4(rb.first_row[b], rb.end_row[b], seg ? rb.kbeg[b] : M.ptr[rb.first_row[b]], seg ? rb.kend[b] : M.ptr[rb.end_row[b]]);
            lg[b] = make_int4(rb.lfirst[b], rb.lseg[b], rb.lbase[b], rb.lid[b]);
        }
        blk_meta.upload(meta);
        blk_long.upload(lg);
        alloc_long(rb.n_long, rb.n_long_slots);
        if (sp) split.upload(*sp);
        nblocks = rb.nb();
    }
    CsrDev dev() const {
        return CsrDev{ptr.d, col.d, val.d, first_row.d, blk_prob.d, blk_meta.d, blk_rs.d, split.d, nblocks, blk_long.d, long_part.d, long_cnt.d,
                      (nblocks <= spin_max_tiles && long_spin_enabled()) ? 1 : 0};
    }
};

// Device side of a band view (score_band.hpp): the unified tile tables, V and the remainder arrays.  The source CsrBufs
// keeps its pattern and values (the factor kernels, k_kval and the CSR tiles of the view read them).
struct BandBufs {
    bool on = false;
    BandLayout L;  // host copy (part_ptr, dst, bytes)
    DevBuf<double> V;
    DevBuf<int32_t> rem_col, rowseg, dst, prob, rs, part_ptr;
    DevBuf<int4> meta, meta2, lng;
    DevBuf<double> long_part;
    DevBuf<unsigned long long> long_cnt;
    int nblocks = 0;
    void upload(BandLayout&& l) {
        L = std::move(l);
        on = L.on;
        if (!on) return;
        nblocks = L.nb();
        UploadBatch ub;
        V.alloc((size_t)L.v_size + 64);
        fill_zero_async(V.d, V.n * sizeof(double), tl_copy_stream);  // padding slots stay zero for ever
        rem_col.upload_padded(L.rem_col, 64);
        rowseg.upload(L.rowseg); dst.upload(L.dst); prob.upload(L.prob); rs.upload(L.rs); part_ptr.upload(L.part_ptr);
        std::vector<int4> m((size_t)nblocks), m2((size_t)nblocks);
        for (int b = 0; b < nblocks; ++b) {
            m[(size_t)b] = make_int4(L.meta[4 * (size_t)b], L.meta[4 * (size_t)b + 1], L.meta[4 * (size_t)b + 2], L.meta[4 * (size_t)b + 3]);
            m2[(size_t)b] = make_int4(L.meta2[4 * (size_t)b], L.meta2[4 * (size_t)b + 1], L.meta2[4 * (size_t)b + 2], L.meta2[4 * (size_t)b + 3]);
        }
        meta.upload(m); meta2.upload(m2);
        std::vector<int4> lg((size_t)nblocks);
        for (int b = 0; b < nblocks; ++b) lg[(size_t)b] = make_int4(L.lng[4 * (size_t)b], L.lng[4 * (size_t)b + 1], L.lng[4 * (size_t)b + 2], L.lng[4 * (size_t)b + 3]);
        lng.upload(lg);
        long_part.alloc((size_t)std::max(1, L.n_long_slots) * kLongVals);
        long_cnt.alloc((size_t)std::max(1, L.n_long));
        fill_zero_async(long_cnt.d, long_cnt.n * sizeof(unsigned long long), tl_copy_stream);
        fill_words_async(long_part.d, (uint32_t)kLongSentinel32, long_part.n * sizeof(double), tl_copy_stream);
    }
    BandDev dev() const {
        BandDev d{};
        d.val = V.d; d.rem_col = rem_col.d; d.rowseg = rowseg.d; d.meta2 = meta2.d; d.rem0 = (int32_t)L.rem0; d.bs = L.bs;
        for (int j = 0; j < kBandMaxS / 2; ++j) d.offw[j] = L.offw[j];
        return d;
    }
};

// The passes of the Ruiz equilibration on the device (RuizOffload, score_host.hpp): the raw P and A of ONE problem go
// up once, 3 small launches per pass run back to back, D and E come back.  The host loop it replaces is ten
// barrier-separated sweeps of a 16-thread team -- 3 ms on a quiet host, several times that on a busy one; this is
// ~1 ms either way.  Buffers and stream come from the process-wide caches and go back before it returns.
struct RuizDevice : RuizOffload {
    int device = 0;
    // What the passes held on the device stays there for the handle's setup (HipBackend::init derives the equilibrated A,
    // G1 and G2 from it -- k_derive_a / k_derive_g -- instead of uploading them) and goes back to the block cache when the
    // setup is over (drop()).
    DevArena keep;
    bool kept = false;
    DevBuf<int32_t> Pp, Pc, Ap, Ac, dat, dpos, drow, dg;
    DevBuf<double> Pv, Av, dD, dE, dd, de;
    int64_t k_n = 0, k_m = 0, k_nnzA = 0, k_nr = 0;
    int k_rep = 1;
    std::mutex mu;
    void drop() {
        kept = false;
        for (DevBuf<int32_t>* b : {&Pp, &Pc, &Ap, &Ac, &dat, &dpos, &drow, &dg}) b->release();
        for (DevBuf<double>* b : {&Pv, &Av, &dD, &dE, &dd, &de}) b->release();
        keep.release_all();
    }
    bool passes(const score_problem& p, int iters, int rep, int64_t rep_n, const std::vector<int32_t>& atp,
                const std::vector<int32_t>& atpos, const std::vector<int32_t>& arow, const std::vector<int32_t>& gstart,
                double* D, double* E) override {
        if (std::getenv("SCORE_NO_DEVICE_RUIZ")) return false;
        std::lock_guard<std::mutex> one_at_a_time(mu);  // (the kept buffers are members)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return false;  // (score_create reports it)
        DeviceGuard guard(device);
        const int n = p.n, m = p.m;
        const int64_t nnzP = p.P_rowptr[n], nnzA = p.A_rowptr[m];
        const int64_t ngroups = (int64_t)gstart.size() - 1;
        const int64_t n_act = rep > 1 ? n - (int64_t)(rep - 1) * rep_n : n;
        hipStream_t st = stream_pool().take(device);
        bool ok = true;
        drop();
        {
            DevArena& arena = keep;
            arena.dev = device;
            StageArena stage;  // (the raw matrices go up through pinned staging; released after the synchronisation below)
            stage.dev = device;
            struct Scope {
                DevArena* a_; hipStream_t s_; StageArena* g_;
                Scope(DevArena* a, hipStream_t s, StageArena* g) : a_(tl_arena), s_(tl_copy_stream), g_(tl_stage) {
                    tl_arena = a; tl_copy_stream = s;
                    if (stage_limit_bytes() > 0) tl_stage = g;
                }
                ~Scope() { tl_arena = a_; tl_copy_stream = s_; tl_stage = g_; }
            } scope(&arena, st, &stage);
            try {
                Pp.upload_from(p.P_rowptr, (size_t)n + 1);
                if (rep > 1) {
                    // only the columns of replica 0 and of the tail are swept (P is symmetric: their rows): two contiguous
                    // ranges of P's entries go up, the rest of the buffers is never read
                    Pc.alloc((size_t)nnzP); Pv.alloc((size_t)nnzP);
                    const int64_t e0 = p.P_rowptr[rep_n], t0 = p.P_rowptr[(int64_t)rep * rep_n];
                    auto up = [&](void* dst, const void* src, size_t bytes) {
                        if (!(tl_stage && tl_stage->upload(dst, src, bytes, st))) staged_h2d(dst, src, bytes, st);
                    };
                    if (e0 > 0) {
                        up(Pc.d, p.P_col, (size_t)e0 * sizeof(int32_t));
                        up(Pv.d, p.P_val, (size_t)e0 * sizeof(double));
                    }
                    if (nnzP > t0) {
                        up(Pc.d + t0, p.P_col + t0, (size_t)(nnzP - t0) * sizeof(int32_t));
                        up(Pv.d + t0, p.P_val + t0, (size_t)(nnzP - t0) * sizeof(double));
                    }
                } else {
                    Pc.upload_from(p.P_col, (size_t)nnzP); Pv.upload_from(p.P_val, (size_t)nnzP);
                }
                Ap.upload_from(p.A_rowptr, (size_t)m + 1); Ac.upload_from(p.A_col, (size_t)nnzA); Av.upload_from(p.A_val, (size_t)nnzA);
                dat.upload_from(atp.data(), atp.size()); dpos.upload_from(atpos.data(), atpos.size());
                dg.upload_from(gstart.data(), gstart.size());
                // (D = E = 1 and the row of every entry of A are made on the device: 4 MB less to send)
                drow.alloc(std::max<size_t>(1, arow.size())); dD.alloc((size_t)std::max(1, n)); dE.alloc((size_t)std::max(1, m));
                {
                    RuizInitArgs ia{};
                    ia.D = dD.d; ia.E = dE.d; ia.n = n; ia.m = m; ia.A_ptr = Ap.d; ia.arow = drow.d;
                    const unsigned gi = (unsigned)((std::max(n, m) + kThreads - 1) / kThreads);
                    hipLaunchKernelGGL(k_ruiz_init, dim3(std::max(1u, gi)), dim3(kThreads), 0, st, ia);
                }
                dd.alloc((size_t)std::max<int64_t>(1, n_act)); de.alloc((size_t)std::max<int64_t>(1, ngroups));
                RuizArgs a{};
                a.P_ptr = Pp.d; a.P_col = Pc.d; a.P_val = Pv.d; a.A_ptr = Ap.d; a.A_col = Ac.d; a.A_val = Av.d;
                a.atp = dat.d; a.atpos = dpos.d; a.arow = drow.d; a.gstart = dg.d;
                a.D = dD.d; a.E = dE.d; a.d = dd.d; a.e = de.d;
                a.n_act = n_act; a.nr = rep > 1 ? rep_n : 0; a.ngroups = ngroups; a.rep = rep;
                const unsigned gc = (unsigned)((n_act + (kThreads / 64) - 1) / (kThreads / 64));
                const unsigned gg = (unsigned)std::max<int64_t>(1, (ngroups + kThreads - 1) / kThreads);
                const unsigned ga = (unsigned)std::max<int64_t>(1, (std::max(n_act, ngroups) + kThreads - 1) / kThreads);
                for (int it = 0; it < iters; ++it) {
                    if (n_act) hipLaunchKernelGGL(k_ruiz_cols, dim3(gc), dim3(kThreads), 0, st, a);
                    if (ngroups) hipLaunchKernelGGL(k_ruiz_groups, dim3(gg), dim3(kThreads), 0, st, a);
                    hipLaunchKernelGGL(k_ruiz_apply, dim3(ga), dim3(kThreads), 0, st, a);
                }
                HIP_CHECK(hipGetLastError());
                staged_d2h(D, dD.d, (size_t)n * sizeof(double), st);
                if (m) staged_d2h(E, dE.d, (size_t)m * sizeof(double), st);
                HIP_CHECK(sync_stream(st));
            } catch (const std::exception&) {
                (void)sync_stream(st);
                (void)hipGetLastError();
                ok = false;  // the host loop takes over (D, E may be partly written: reset them)
                std::fill(D, D + n, 1.0);
                std::fill(E, E + m, 1.0);
            }
        }
        stream_pool().give(device, st);
        if (ok) { kept = true; k_n = n; k_m = m; k_nnzA = nnzA; k_rep = rep; k_nr = rep > 1 ? rep_n : 0; }
        else drop();
        return ok;
    }
};

struct HipBackend {
    // K's values, the chain factors and the Jacobi diagonal are derived on the device from K0, K1 and
    // rho (derive_rho_data): the host neither factors nor uploads anything when a penalty changes
    static constexpr bool kFactorOnHost = false;
    static bool allow_rep() { return true; }  // replicated problems (HostSystem::rep): K_row streamed once for all replicas
    RuizDevice ruiz_dev;
    RuizOffload* ruiz_offload(const score_settings& s_) { ruiz_dev.device = s_.device; return &ruiz_dev; }
    const HostSystem* H = nullptr;
    score_settings st{};
    hipStream_t stream = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_iters = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevArena arena;  // declared before every buffer: destroyed after them

    CsrBufs K, G1, G2;
    BandBufs Kb, Hb;  // band views of K and of the Newton matrix (score_band.hpp); off: the CSR-stream kernels serve them
    // SCORE_NO_BAND=1: the CSR-stream kernels serve K (tests: both layouts against the twin)
    bool band_k(const HostSystem&) const {
        // K through its band view whatever the batch size (what a problem computes must not depend on its batch mates: the two
        // layouts add a row's terms in different orders).  A lock-step batch streams K from HBM and the view moves 19 % fewer
        // bytes: batch of 16 headline problems kp 46 -> 34 us, kpb 50 -> 40 us; a single problem's product is a chain of
        // dependent trips bound by its slowest workgroup (the landmark rows' segments): 6.4 / 7.1 us either way.
        return std::getenv("SCORE_NO_BAND") == nullptr;
    }
    // (a band view of the Newton matrix was measured without effect -- 64 config-5 trials 7.0 ms per solve with and without,
    //  headline default solve 4.5 ms either way -- and cost setup time: removed in round 6, profiles/TRIED.md)
    static constexpr bool band_h(const HostSystem&) { return false; }
    int kblocks() const { return Kb.on ? Kb.nblocks : K.nblocks; }   // tiles of the K product (= p'w partials per launch)
    int hblocks() const { return Hb.on ? Hb.nblocks : Hm.nblocks; }
    DevBuf<int32_t> A_ptr, A_col;
    DevBuf<double> A_val;
    DevBuf<double> q, b, invD, invE, rho, fac, dinv, K0d, K1d;
    DevBuf<int32_t> kposd, kposs, kdiagpos;  // K.val positions of the chain blocks / Jacobi diagonals
    DevBuf<int32_t> done, cone_row, cone_dim, cone_type, cone_block_first, cone_block_prob;
    DevBuf<int4> cone_meta;
    DevBuf<int2> cone_large;    // {cone, problem} of the cones with more than kWaveCone rows (k_cone_wave)
    int n_large_cones = 0;
    DevBuf<int32_t> cone_cols;  // 8 per cone (two int4)
    DevBuf<double> cone_vals;   // 8 per cone (four double2)
    DevBuf<int32_t> node_col, diag_cols, prec_part_ptr, kblk_part_ptr;
    DevBuf<PrecWork> prec_work, factor_work;   // factor_work: what a factorisation of K visits (HostSystem::factor_work)
    DevBuf<ChainDesc> chains, chainsH;         // chainsH / levelsH: the same chains with factors of their own (Newton matrix)
    DevBuf<ChainLevelDesc> levels, levelsH;
    DevBuf<PrecRecord> prec_rec, prec_recH;    // one record per work item: work + chain + level table (k_prec_pre)
    DevBuf<int64_t> fac_rangeK, fac_rangeH;    // per chain: its factor range (k_fac_round_items)
    DevBuf<int64_t> q_entpart;                 // per problem: its entry range in the Newton matrix
    int64_t q_ent_max = 0;
    static constexpr int64_t kHelpEntries = (int64_t)kPrecThreads * kPrecChunk;  // vector entries per update helper
    int n_help = 0;                            // update-helper records appended to prec_rec / prec_recH
    DevBuf<int32_t> vb_first, vb_end, vb_prob; // blocks of <= 256 vector entries per problem (k_xupdate)
    int n_vblocks = 0;
    DevBuf<double> xtu, xy, s, r, z, p, p2, w, kx, step;
    DevBuf<double> pw_part, rz_part0, rz_part1, rz_meas0, rz_meas1, pres_part, dres_part;
    // graphs drawn by the library's generator on this device (score_generate_manhattan): the measurement arrays of the handle's
    // graphs as they lie in the generator's device memory -- the assembler kernels read them there, nothing is uploaded again
    struct GenSource {
        const int32_t* rel_base; const int32_t* rel_to; const double* rel_t; const double* rel_R; const double* rel_kappa; const double* rel_tau;
        const int32_t* rng_a; const int32_t* rng_b; const double* rng_dist; const double* rng_prec;
    };
    const GenSource* gen_src = nullptr;  // (set by score_create_from_generated for the duration of the create)
    // ---- segmented long chains (score_join.hpp): K's set, the Newton matrix's set ----
    int n_join_items = 0, n_join_chains = 0, n_join_seps = 0;
    bool join_suspend = false;  // the spike solves of a refresh: the chain kernel alone
    DevBuf<JoinChain> join_jc;
    DevBuf<JoinItem> join_items;
    DevBuf<int32_t> join_sep_col, join_sep_diag, join_pcol, join_pprev, join_posd_K, join_poss_K, join_posd_H, join_poss_H, join_zero;
    DevBuf<double> join_W_K, join_W_H, join_data_K, join_data_H, join_rhs, join_tmp_p, join_tmp_rz, join_zb;
    // ---- semismooth-Newton polish (score_polish*.hpp) ----
    DevBuf<float> fac32, q_fac32;  // float copies of the chain factors (ADMM / Newton), see k_fac_round
    DevBuf<float> deepK, deepH;    // lane-major copies of their coarse levels (k_deep_pack -> k_prec_pre<.., float, true>)
    DevBuf<int32_t> deep_map;
    bool prec_reg = false;         // every chain has a lane plan: the register-resident variant serves the 4-byte streams
    size_t prec_reg_lds = 0;
    bool use_fac32 = false;     // ADMM-loop factors (K)
    bool newton_fac32 = false;  // Newton-polish factors (H): fac_fp32 = 2 only, see DESIGN.md section 4
    PolishData Q;
    std::future<void> polish_build;  // build_polish (or only its structure check) runs beside the uploads of init()
    bool polish_on_device = false;   // the Newton matrix's pattern and lists come from score_polish_device.hpp
    bool derive_ag = false;          // the equilibrated A, G1, G2 were derived on the device (k_derive_a / k_derive_g)
    int64_t hm_nnz = 0;              // entries of the Newton matrix
    CsrBufs Hm;
    DevBuf<double> q_Pon, q_ccoef, q_Bbuf, q_fpart, q_X0, q_X1, q_g, q_delta, q_fac, q_dinv, q_work, q_dummy, q_gd, q_pw;
    DevBuf<double> q_aabs, q_ck, q_theta, q_xstar;
    DevBuf<int32_t> q_cptr, q_ccone, q_cab, q_head, q_ishead, q_posd, q_poss, q_diagpos, q_hblk_part, q_long, q_long_prob;
    int n_long = 0;
    // lock-step polish of a batch (count > 1)
    DevBuf<int32_t> q_skip, q_reref, q_fskip, q_act;  // (q_skip, q_reref, q_fskip: views into ctl)
    DevBuf<double> q_step;
    DevBuf<int64_t> q_seg_begin, q_seg_end;
    static constexpr int kFlagSlots = 32;
    char* h_ring = nullptr;      // pinned (host-mapped) ring of upload slots
    size_t ring_slot_bytes = 0;
    int flag_slot = 0, ring_used = 0;
    DevBuf<double> q_negg;                   // -gradient of the last evaluation (right-hand side of the next PCG)
    DevBuf<double> q_gate_tol2, q_gate_ref;  // device-side PCG termination (pcg_gate)
    DevBuf<int32_t> q_gate_used;
    int32_t* h_gate = nullptr;   // [gate flags | iterations used]          } windows into h_rep
    double* h_gd = nullptr;      // partials of g'delta                      }
    // Everything the host reads back between launches lives in ONE device allocation, mirrored by
    // one pinned host buffer, so that a convergence check (ADMM) or a Newton iteration costs a
    // single device-to-host copy:  [pres | dres | fpart | gd | gate flags, gate counts]
    DevBuf<double> rep;          // window: the device address of h_rep (host-mapped pinned memory)
    double* h_rep = nullptr;
    size_t h_rep_bytes = 0, h_ring_bytes = 0;
    size_t rep_dres_off = 0;     // doubles
    // Kernels write their per-workgroup partials straight into that host-mapped memory; small device
    // arrays the device itself reads (r'z measurements, PCG gate words) are pushed there by k_push,
    // which then publishes a sequence number the host spins on: no copy command, no stream
    // synchronisation on the path of a convergence check or a Newton iteration.
    unsigned long long* h_seq = nullptr;   // [0]: last published sequence number (host-mapped)
    unsigned long long* d_seq = nullptr;
    unsigned long long seq_next = 0;
    double* h_meas = nullptr;    // [rz_meas0 | rz_meas1] as pushed
    double* d_meas = nullptr;
    int32_t* d_gate_host = nullptr;  // device address of h_gate
    int32_t* h_gate_live = nullptr;  // [fired: epoch | used: epoch << 12 | STEPs] per problem, written by pcg_gate itself
    int32_t* d_gate_live = nullptr;
    int32_t gate_epoch = 0;
    char* d_ring = nullptr;      // device address of h_ring
    size_t n_fpart = 0, n_gd = 0;
    // ... and everything the host tells the kernels per problem in one upload: [step | tol2 | skip]
    DevBuf<double> ctl;
    DevBuf<int32_t> q_pcgdone;   // window into rep: raised by pcg_gate
    int pcg_used_total = 0;
    double newton_eta_max = 1e-1;  // inexact Newton: linear residual <= min(eta_max, coef * |g|^pow)
    double newton_eta_coef = 1.0, newton_eta_pow = 0.5;
    double* h_newton = nullptr;  // window into h_rep: partials of the cone part of F

    // (Rounds 4-5 could evaluate the previous iteration's cones inside the right-hand-side kernel -- five launches per ADMM
    //  iteration instead of six, opt-in, measured no faster: 13.9 us against 6.6 + 7.0 + 1.4 -- removed in round 6, TRIED.md.)
    void begin_sequence() {}
    int cg_iters = 2;
    const double* last_rz = nullptr;  // r'z partials / direction of the pending end-of-PCG update
    const double* last_p = nullptr;
    const double* dbg_p = nullptr;   // the buffer the last queued iteration left its last direction in (score_debug_get "p")
    double* h_pres = nullptr;  // pinned
    double* h_dres = nullptr;
    int n_cone_blocks = 0, n_prec = 0;  // n_prec: work items of the ACTIVE preconditioner launch (split or not)
    // split chain kernel (score_split.hpp / score_prec_wave.hpp)
    SplitSystem split;
    std::vector<int32_t> active_part_ptr;   // host copy of the active per-problem work ranges
    DevBuf<PrecWork> split_work;
    DevBuf<SplitItem> split_items;
    DevBuf<SplitPlan> split_plans;
    DevBuf<int32_t> split_stage;
    DevBuf<double> split_xbuf;
    DevBuf<unsigned int> split_xflag, split_epoch;
    size_t split_lds = 0;
    unsigned long long split_poll_limit = 0;
    size_t prec_lds = 0;
    bool prec_lds0 = true;
    bool prec_pre = false;  // every chain fits the lane budget of k_prec_pre (4 x 4 blocks: with the 4-byte factor stream only)
    size_t prec_pre_lds = 0;
    static bool n_prec_chains(const HostSystem& h) { return !h.chains.empty(); }

    ~HipBackend() {
        if (trace_on("host"))
            std::fprintf(stderr, "[score host] Newton PCG queueing: %.2f ms for %ld launches (%.2f us each); waits: %ld, %.2f ms\n", enq_ms, enq_launches,
                         enq_launches ? 1e3 * enq_ms / (double)enq_launches : 0.0, waits, wait_ms);
        PhaseTimer pt(st.verbose != 0);
        if (pre_slot && h_ring) release_prequeued();  // (never pending here; a kernel waiting for the host must not outlive its ring)
        if (stream) (void)sync_stream(stream);
        pt.mark("destroy: sync");
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        pt.mark("destroy: graph");
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        block_cache().give(h_rep, h_rep_bytes, st.device, true);
        block_cache().give(h_ring, h_ring_bytes, st.device, true);
        for (auto& pb : setup_pinned) block_cache().give(pb.first, pb.second, st.device, true);  // (an init that threw: drained above)
        pt.mark("destroy: events, pinned blocks");
        stream_pool().give(st.device, stream);  // (drained above)
        pt.mark("destroy: stream");
    }

    // ---- the handle's matrices built on the device (score_setup_device.hpp; HostSystem::device_setup) ----
    DevBuf<double> Dd, Ed;                      // the equilibration's scales (the host keeps none)
    DevBuf<int32_t> tab_xoff, tab_roff, tab_nr;  // ProbTab
    DevArena setup_tmp;                          // scratch of the setup: back to the block cache when init() is over
    std::vector<std::pair<void*, size_t>> setup_pinned;  // ... and its pinned staging blocks
    int64_t g1_nnz = 0, nnzP_full = 0;
    ProbTab prob_tab() const {
        ProbTab t{};
        t.xoff = tab_xoff.d; t.roff = tab_roff.d; t.nr = tab_nr.d; t.count = H->count; t.rep = H->rep;
        return t;
    }
    // asked by build_system once sizes and the replication structure are known
    bool device_setup_allowed(const HostSystem& h, const score_settings& s_) const {
        if (std::getenv("SCORE_HOST_SETUP") || std::getenv("SCORE_HOST_POLISH_BUILD") || std::getenv("SCORE_NO_DEVICE_RUIZ")) return false;  // (switches that ask for a host-side piece)
        if (h.m_tot <= 0 || s_.chain_split > 0 || band_h(h)) return false;   // (linear mode keeps K0 on the host)
        if (h.rep > 1)
            for (char ex : h.rep_exact_all)
                if (!ex) return false;  // (replicas that differ in their last bits: the host uses every replica's own values)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || s_.device < 0 || s_.device >= ndev) return false;  // (init reports it)
        return true;
    }
    // the same question for handles made from factor graphs (score_create_from_graphs): record bounds by formula
    bool device_setup_ok_graphs(const HostSystem& h, const score_graph* graphs, const score_settings& s_) const {
        if (std::getenv("SCORE_HOST_ASSEMBLE") || !device_setup_allowed(h, s_)) return false;
        const int d = graphs[0].dim, D1 = d + 1;
        if (h.rep != d) return false;  // (the assembler kernels write one replica's rows: SCORE_NO_REPLICATION takes the host assembler)
        int64_t rec = h.n_tot, con = 0;
        for (int p = 0; p < h.count; ++p) {
            const score_graph& g = graphs[p];
            if (g.dim != d || g.relaxation != graphs[0].relaxation) return false;
            rec += g.n_rel * (D1 + 3 * D1 * D1) * d + g.n_rng * 9 * d + g.n_lprior * d;
            con += g.n_rng * (int64_t)(4 * d * d + 1);
        }
        return rec + con + 64 < ((int64_t)1 << 31);
    }
    bool device_setup_ok(const HostSystem& h, const score_problem* probs, const score_settings& s_) const {
        if (!device_setup_allowed(h, s_)) return false;
        // record bounds of the K and Newton-matrix builds: 32-bit positions
        int64_t rec = h.n_tot, sq = 0;
        for (int p = 0; p < h.count; ++p) {
            const score_problem& pr = probs[p];
            rec += pr.P_rowptr[pr.n];
            for (int r = 0; r < pr.m; ++r) { const int64_t L = pr.A_rowptr[r + 1] - pr.A_rowptr[r]; sq += L * L; }
        }
        // (the Newton matrix's contributions: (entries of a cone's tail rows)^2 per cone, at most (sum of the rows' lengths)^2)
        int64_t con = 0;
        for (int p = 0; p < h.count; ++p) {
            const score_problem& pr = probs[p];
            int row = pr.z;
            for (int c = 0; c < pr.n_soc; ++c) {
                const int64_t L = pr.A_rowptr[row + pr.soc_dims[c]] - pr.A_rowptr[row];
                con += L * L;
                row += pr.soc_dims[c];
            }
        }
        return rec + sq + 64 < ((int64_t)1 << 31) && rec + con + 64 < ((int64_t)1 << 31);
    }
    template <class T>
    void copy_up(T* dst, const T* src, size_t count) {  // staged when small, pageable otherwise (returns once the source is consumed)
        if (!count) return;
        if (tl_stage && tl_stage->upload(dst, src, count * sizeof(T), stream)) return;
        staged_h2d(dst, src, count * sizeof(T), stream);
    }
    // records (key, idx, v0, v1; rec_max of them, the unused tail padded with row n_rows) -> CSR pattern + summed values:
    // stable sort, flags, scan, merge (an entry adds its records in order), row counts -> row pointers.  Everything lands in
    // arrays of the CURRENT arena (the caller's scratch); result[0] = entries.
    struct MergeOut {
        DevBuf<int32_t> ptr, col;
        DevBuf<double> o0, o1;
        DevBuf<long long> result;
    };
    // rows' records contiguous and in row order (offsets row_off[0 .. n_rows]): every row sorted by (column, position) into
    // key_out / idx_out (score_setup_device.hpp, k_row_rank_sort); sk0 / sk1: two scratch arrays of rec_max keys for the long rows
    static int pos_bits(int64_t rec_max) { int b = 1; while (((int64_t)1 << b) < rec_max) ++b; return b; }
    struct RowSort {
        DevBuf<int32_t> seg;   // begin[cap], end[cap], count
        int32_t cap = 0;
        size_t bytes = 0;
    };
    void sort_rows_plan(RowSort& rs, int64_t n_rows, int64_t rec_max, int bits, unsigned long long* sk0, unsigned long long* sk1) {
        rs.cap = (int32_t)(rec_max / kShortRow + 1);
        rs.seg.alloc((size_t)2 * rs.cap + 1);
        fill_zero_async(rs.seg.d, ((size_t)2 * rs.cap + 1) * sizeof(int32_t), stream);  // (unused segments stay [0, 0))
        HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, rs.bytes, sk0, sk1, (unsigned)rec_max, (unsigned)rs.cap, rs.seg.d, rs.seg.d + rs.cap, 0,
                                                     (unsigned)(bits + pos_bits(rec_max)), stream));
    }
    void sort_rows(RowSort& rs, int64_t n_rows, int64_t rec_max, int bits, const unsigned long long* key_in, unsigned long long* key_out, uint32_t* idx_out,
                   unsigned long long* sk0, unsigned long long* sk1, const long long* row_off, void* scratch, const uint32_t* idx_in = nullptr) {
        RowSortArgs a{};
        a.key = key_in; a.idx = idx_in; a.off = row_off; a.key_out = key_out; a.idx_out = idx_out; a.sk = sk0;
        a.seg_b = rs.seg.d; a.seg_e = rs.seg.d + rs.cap; a.seg_n = rs.seg.d + 2 * rs.cap; a.seg_cap = rs.cap;
        a.rec_max = rec_max; a.n_rows = n_rows; a.pbits = pos_bits(rec_max);
        const unsigned grec = (unsigned)((rec_max + 255) / 256);
        hipLaunchKernelGGL(k_row_rank_sort, dim3(grec), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(k_rows_long_list, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, a);
        HIP_CHECK(rocprim::segmented_radix_sort_keys(scratch, rs.bytes, sk0, sk1, (unsigned)rec_max, (unsigned)rs.cap, a.seg_b, a.seg_e, 0,
                                                     (unsigned)(bits + a.pbits), stream));
        a.sk = sk1;
        hipLaunchKernelGGL(k_row_keys_back, dim3(grec), dim3(256), 0, stream, a);
    }
    void merge_records(int64_t n_rows, int64_t rec_max, DevBuf<unsigned long long>& key0, DevBuf<uint32_t>& idx0, const double* v0,
                       const double* v1, int32_t qcol, double* qout, MergeOut& out, const long long* row_off = nullptr) {
        DevBuf<unsigned long long> key1, flag, flag_s;
        DevBuf<uint32_t> idx1;
        DevBuf<int32_t> row_cnt;
        DevBuf<int4> long_run;
        key1.alloc((size_t)rec_max); idx1.alloc((size_t)rec_max); flag.alloc((size_t)rec_max); flag_s.alloc((size_t)rec_max);
        const int64_t long_max = rec_max / kLongRun + 1;
        long_run.alloc((size_t)long_max);
        out.ptr.alloc((size_t)n_rows + 1); out.col.alloc((size_t)rec_max + 64); out.o0.alloc((size_t)rec_max + 64);
        if (v1) out.o1.alloc((size_t)rec_max + 64);
        // records in no particular order (the assembler's): into their rows' slots first -- count, scan, scatter
        DevBuf<unsigned long long> key_b;
        DevBuf<uint32_t> idx_b;
        DevBuf<long long> cnt_b;
        DevBuf<unsigned int> cur_b;
        const bool bucket = !row_off;
        std::optional<UploadBatch> fills;  // (the fills below in one launch; closed before the first kernel)
        fills.emplace();
        {
            ZeroGroup zg;
            zg.add(row_cnt, (size_t)n_rows + 1); zg.add(out.result, 2);
            if (bucket) { zg.add(cnt_b, (size_t)n_rows + 2); zg.add(cur_b, (size_t)n_rows + 1); }
            zg.commit(stream);
        }
        int bits = 1;
        while (((int64_t)1 << bits) <= n_rows) ++bits;  // (the padding row n_rows sorts last)
        if (qcol >= 0) bits = 31;                         // (the q records' column, kQCol, is the largest there is)
        size_t tb = 0, tb2 = 0, tb3 = 0, tb4 = 0;
        if (bucket) {
            key_b.alloc((size_t)rec_max); idx_b.alloc((size_t)rec_max);
            HIP_CHECK(rocprim::exclusive_scan(nullptr, tb4, cnt_b.d, cnt_b.d, (long long)0, (size_t)n_rows + 2, rocprim::plus<long long>(), stream));
        }
        RowSort rs;
        sort_rows_plan(rs, n_rows, rec_max, bits, flag.d, flag_s.d);  // (the flag arrays are free until k_rec_flags)
        tb = rs.bytes;
        fills.reset();
        HIP_CHECK(rocprim::inclusive_scan(nullptr, tb2, flag.d, flag_s.d, (size_t)rec_max, rocprim::plus<unsigned long long>(), stream));
        HIP_CHECK(rocprim::exclusive_scan(nullptr, tb3, row_cnt.d, out.ptr.d, (int32_t)0, (size_t)n_rows + 1, rocprim::plus<int32_t>(), stream));
        DevBuf<unsigned char> scratch;
        scratch.alloc(std::max(std::max(tb, tb4), std::max(tb2, tb3)) + 256);
        if (bucket) {
            const unsigned grec0 = (unsigned)((rec_max + 255) / 256);
            hipLaunchKernelGGL(k_rec_count, dim3(grec0), dim3(256), 0, stream, (const unsigned long long*)key0.d, rec_max, n_rows, (unsigned long long*)cnt_b.d);
            HIP_CHECK(rocprim::exclusive_scan((void*)scratch.d, tb4, cnt_b.d, cnt_b.d, (long long)0, (size_t)n_rows + 2, rocprim::plus<long long>(), stream));
            hipLaunchKernelGGL(k_rec_scatter, dim3(grec0), dim3(256), 0, stream, (const unsigned long long*)key0.d, (const uint32_t*)idx0.d, rec_max, n_rows,
                               (const long long*)cnt_b.d, cur_b.d, key_b.d, idx_b.d);
            hipLaunchKernelGGL(k_rec_pad, dim3(grec0), dim3(256), 0, stream, key_b.d, idx_b.d, (const long long*)(cnt_b.d + n_rows), rec_max, n_rows);
            sort_rows(rs, n_rows, rec_max, bits, key_b.d, key1.d, idx1.d, flag.d, flag_s.d, cnt_b.d, (void*)scratch.d, idx_b.d);
        } else sort_rows(rs, n_rows, rec_max, bits, key0.d, key1.d, idx1.d, flag.d, flag_s.d, row_off, (void*)scratch.d);
        RecArgs ra{};
        ra.n_rows = n_rows; ra.rec_max = rec_max; ra.key = key1.d; ra.idx = idx1.d; ra.v0 = v0; ra.v1 = v1;
        ra.flag = flag.d; ra.row_cnt = row_cnt.d; ra.col = out.col.d; ra.o0 = out.o0.d; ra.o1 = v1 ? out.o1.d : nullptr;
        ra.result = out.result.d; ra.long_run = long_run.d; ra.long_max = (int32_t)long_max; ra.qcol = qcol; ra.qout = qout;
        const unsigned grec = (unsigned)((rec_max + 255) / 256);
        hipLaunchKernelGGL(k_rec_flags, dim3(grec), dim3(256), 0, stream, ra);
        HIP_CHECK(rocprim::inclusive_scan((void*)scratch.d, tb2, flag.d, flag_s.d, (size_t)rec_max, rocprim::plus<unsigned long long>(), stream));
        ra.flag = flag_s.d;
        hipLaunchKernelGGL(k_rec_merge, dim3(grec), dim3(256), 0, stream, ra);
        hipLaunchKernelGGL(k_rec_total, dim3(1), dim3(64), 0, stream, ra);
        hipLaunchKernelGGL(k_rec_long, dim3((unsigned)((long_max + 3) / 4)), dim3(256), 0, stream, ra);
        HIP_CHECK(rocprim::exclusive_scan((void*)scratch.d, tb3, row_cnt.d, out.ptr.d, (int32_t)0, (size_t)n_rows + 1, rocprim::plus<int32_t>(), stream));
        HIP_CHECK(hipGetLastError());
    }
    struct ArenaSwap {  // allocations of a scope go to another arena
        DevArena* keep;
        explicit ArenaSwap(DevArena* a) : keep(tl_arena) { tl_arena = a; }
        ~ArenaSwap() { tl_arena = keep; }
    };

    // The raw program of the handle on the device, in global numbering: P (the rows of replica 0 and of the tail), q, A, b.
    // From the caller's problems (uploads) or from the factor graphs themselves (the device assembler, k_ga_*).
    struct RawDev {
        DevBuf<int32_t> Pp, Pc, Ac;
        DevBuf<double> Pv, Av, qraw, braw;
        MergeOut pm;                 // (graph path: P as the merge left it)
        const int32_t* P_ptr = nullptr; const int32_t* P_col = nullptr; const double* P_val = nullptr;
        std::vector<int32_t> Aptr;   // global row pointers of A (host copy)
        int64_t pe = 0;              // stored entries of P (graph path: an upper bound)
        int64_t ae = 0, sq = 0, n_stored = 0, nnzP_full = 0;
        bool exact = true;           // pe / nnzP_full are exact counts
    };
    void fill_tab(const HostSystem& h) {
        const int count = h.count;
        std::vector<int32_t> xo32((size_t)count + 1), ro32((size_t)count + 1), nr32((size_t)count, 0);
        for (int p = 0; p <= count; ++p) { xo32[(size_t)p] = (int32_t)h.xoff[p]; ro32[(size_t)p] = (int32_t)h.roff[p]; }
        for (int p = 0; p < count; ++p) nr32[(size_t)p] = h.rep > 1 ? (int32_t)h.rep_n[(size_t)p] : 0;
        ArenaSwap persist(&arena);
        UploadBatch ub;
        tab_xoff.upload(xo32); tab_roff.upload(ro32); tab_nr.upload(nr32);
    }
    void raw_from_problems(const HostSystem& h, const score_problem* probs, RawDev& R) {
        const int count = h.count;
        const int64_t n = h.n_tot, m = h.m_tot;
        const int rep = h.rep;
        // ---- host: global row pointers of the raw matrices (P: the rows of replica 0 and of the tail), entry offsets ----
        std::vector<int32_t> Pptr((size_t)n + 1), pent((size_t)count + 1), aent((size_t)count + 1);
        R.Aptr.assign((size_t)m + 1, 0);
        int64_t pe = 0, ae = 0, sq = 0;
        R.nnzP_full = 0;
        for (int p = 0; p < count; ++p) {
            const score_problem& pr = probs[p];
            const int64_t nr = rep > 1 ? h.rep_n[(size_t)p] : 0, t0 = (int64_t)rep * nr;
            pent[(size_t)p] = (int32_t)pe; aent[(size_t)p] = (int32_t)ae;
            int32_t* pp = &Pptr[(size_t)h.xoff[p]];
            if (rep > 1) {
                const int64_t e0 = pr.P_rowptr[nr], et = pr.P_rowptr[pr.n] - pr.P_rowptr[t0];
                for (int64_t i = 0; i < nr; ++i) pp[i] = (int32_t)(pe + pr.P_rowptr[i]);
                for (int64_t i = nr; i < t0; ++i) pp[i] = (int32_t)(pe + e0);
                for (int64_t i = t0; i < pr.n; ++i) pp[i] = (int32_t)(pe + e0 + (pr.P_rowptr[i] - pr.P_rowptr[t0]));
                pe += e0 + et;
                R.nnzP_full += (int64_t)rep * e0 + et;
            } else {
                for (int64_t i = 0; i < pr.n; ++i) pp[i] = (int32_t)(pe + pr.P_rowptr[i]);
                pe += pr.P_rowptr[pr.n];
                R.nnzP_full += pr.P_rowptr[pr.n];
            }
            int32_t* ap = &R.Aptr[(size_t)h.roff[p]];
            for (int64_t r = 0; r < pr.m; ++r) {
                ap[r] = (int32_t)(ae + pr.A_rowptr[r]);
                const int64_t L = pr.A_rowptr[r + 1] - pr.A_rowptr[r];
                sq += L * L;
            }
            ae += pr.A_rowptr[pr.m];
        }
        Pptr[(size_t)n] = (int32_t)pe; R.Aptr[(size_t)m] = (int32_t)ae;
        pent[(size_t)count] = (int32_t)pe; aent[(size_t)count] = (int32_t)ae;
        R.pe = pe; R.ae = ae; R.sq = sq; R.exact = true;
        R.n_stored = 0;
        for (int p = 0; p < count; ++p) R.n_stored += (h.xoff[p + 1] - h.xoff[p]) - (rep > 1 ? (int64_t)(rep - 1) * h.rep_n[(size_t)p] : 0);
        DevBuf<int32_t> d_pent, d_aent;
        R.Pp.upload(Pptr);
        R.Pc.alloc((size_t)pe + 1); R.Pv.alloc((size_t)pe + 1); R.Ac.alloc((size_t)ae + 1); R.Av.alloc((size_t)ae + 1);
        R.qraw.alloc((size_t)n); R.braw.alloc((size_t)std::max<int64_t>(1, m));
        for (int p = 0; p < count; ++p) {
            const score_problem& pr = probs[p];
            const int64_t nr = rep > 1 ? h.rep_n[(size_t)p] : 0, t0 = (int64_t)rep * nr;
            const int64_t o = pent[(size_t)p];
            if (rep > 1) {
                const int64_t e0 = pr.P_rowptr[nr], b0 = pr.P_rowptr[t0], et = pr.P_rowptr[pr.n] - b0;
                copy_up(R.Pc.d + o, pr.P_col, (size_t)e0); copy_up(R.Pv.d + o, pr.P_val, (size_t)e0);
                copy_up(R.Pc.d + o + e0, pr.P_col + b0, (size_t)et); copy_up(R.Pv.d + o + e0, pr.P_val + b0, (size_t)et);
            } else {
                copy_up(R.Pc.d + o, pr.P_col, (size_t)pr.P_rowptr[pr.n]); copy_up(R.Pv.d + o, pr.P_val, (size_t)pr.P_rowptr[pr.n]);
            }
            copy_up(R.Ac.d + aent[(size_t)p], pr.A_col, (size_t)pr.A_rowptr[pr.m]); copy_up(R.Av.d + aent[(size_t)p], pr.A_val, (size_t)pr.A_rowptr[pr.m]);
            copy_up(R.qraw.d + h.xoff[p], pr.q, (size_t)pr.n); copy_up(R.braw.d + h.roff[p], pr.b, (size_t)pr.m);
        }
        if (count > 1) {
            d_pent.upload(pent); d_aent.upload(aent);
            if (pe) hipLaunchKernelGGL(k_globalise, dim3((unsigned)((pe + 255) / 256)), dim3(256), 0, stream, R.Pc.d, (const int32_t*)d_pent.d, (const int32_t*)tab_xoff.d, count, pe);
            if (ae) hipLaunchKernelGGL(k_globalise, dim3((unsigned)((ae + 255) / 256)), dim3(256), 0, stream, R.Ac.d, (const int32_t*)d_aent.d, (const int32_t*)tab_xoff.d, count, ae);
        }
        HIP_CHECK(hipGetLastError());
        R.P_ptr = R.Pp.d; R.P_col = R.Pc.d; R.P_val = R.Pv.d;
    }

    // The device assembler: the factor graphs' flat arrays go up as they are, every measurement writes its records (k_ga_*),
    // the merge turns them into P (replica 0 + tail rows) and q; A and b are written in place.  c0 comes from the host (a sum
    // over the ranges, the priors and the few measurements at the pinned pose, in assemble_graph's order).
    void raw_from_graphs(HostSystem& h, const score_graph* graphs, RawDev& R) {
        const int count = h.count;
        const int64_t n = h.n_tot, m = h.m_tot;
        const int d = graphs[0].dim, D1 = d + 1, relax = graphs[0].relaxation;
        const int per_rel = D1 + 3 * D1 * D1, per_rng = relax == 0 ? 1 : 9;
        std::vector<GaProb> gp((size_t)count);
        std::vector<int32_t> rel_off((size_t)count + 1, 0), rng_off((size_t)count + 1, 0), pri_off((size_t)count + 1, 0), pin_off((size_t)count + 1, 0);
        std::vector<int32_t> pin_edge;
        R.Aptr.assign((size_t)m + 1, 0);
        int64_t ae = 0, sq = 0;
        for (int p = 0; p < count; ++p) {
            const score_graph& g = graphs[p];
            int64_t Np = 0;
            for (int c = 0; c < g.n_chains; ++c) Np += g.chain_len[c];
            GaProb& P = gp[(size_t)p];
            P.xoff = (int32_t)h.xoff[p]; P.roff = (int32_t)h.roff[p];
            P.Np = (int32_t)Np; P.Nl = g.n_landmarks; P.Nr = (int32_t)g.n_rng;
            P.n_rep = (int32_t)h.rep_n[(size_t)p];
            P.rel_off = rel_off[(size_t)p]; P.rng_off = rng_off[(size_t)p]; P.pri_off = pri_off[(size_t)p]; P.pin_off = pin_off[(size_t)p];
            rel_off[(size_t)p + 1] = rel_off[(size_t)p] + (int32_t)g.n_rel;
            rng_off[(size_t)p + 1] = rng_off[(size_t)p] + (int32_t)g.n_rng;
            pri_off[(size_t)p + 1] = pri_off[(size_t)p] + (int32_t)g.n_lprior;
            for (int64_t e = 0; e < g.n_rel; ++e)
                if (g.rel_base[e] == 0 || g.rel_to[e] == 0) pin_edge.push_back((int32_t)(rel_off[(size_t)p] + e));
            pin_off[(size_t)p + 1] = (int32_t)pin_edge.size();
            // rows of A: a head row (SOCP: the distance variable; QCQP: none) and d rows per range
            P.a_off = (int32_t)ae;
            int32_t* ap = &R.Aptr[(size_t)h.roff[p]];
            for (int64_t r = 0; r < g.n_rng; ++r) {
                const int cnt = relax == 0 ? (g.rng_a[r] != 0) + (g.rng_b[r] != 0) : 1;
                const int head = relax == 0 ? 1 : 0;
                ap[r * D1] = (int32_t)ae;
                ae += head;
                for (int k = 0; k < d; ++k) { ap[r * D1 + 1 + k] = (int32_t)ae; ae += cnt; }
                sq += head + (int64_t)d * cnt * cnt;
            }
        }
        R.Aptr[(size_t)m] = (int32_t)ae;
        {   // (the assembler writes A through its row pointers: they go up first, into the handle's own arena)
            ArenaSwap persist(&arena);
            A_ptr.upload(R.Aptr);
        }
        // record slots: P records [relative poses | ranges | priors] problem by problem, then the q records likewise
        int64_t o = 0;
        for (int p = 0; p < count; ++p) {
            GaProb& P = gp[(size_t)p];
            const score_graph& g = graphs[p];
            P.rec_rel = (int32_t)o; o += g.n_rel * per_rel;
            P.rec_rng = (int32_t)o; o += g.n_rng * per_rng;
            P.rec_pri = (int32_t)o; o += g.n_lprior;
        }
        const int64_t p_slots = o;
        for (int p = 0; p < count; ++p) {
            GaProb& P = gp[(size_t)p];
            const score_graph& g = graphs[p];
            P.recq_pin = (int32_t)o; o += (int64_t)(pin_off[(size_t)p + 1] - pin_off[(size_t)p]) * d * D1;
            P.recq_rng = (int32_t)o; o += relax == 0 ? g.n_rng : 0;
            P.recq_pri = (int32_t)o; o += g.n_lprior * d;
        }
        const int64_t rec_max = o + 64;
        if (rec_max >= ((int64_t)1 << 31)) throw std::runtime_error("score_graph: too many measurement records for one handle");
        R.pe = p_slots; R.ae = ae; R.sq = sq; R.exact = false;
        R.nnzP_full = (int64_t)std::max(1, h.rep) * p_slots;
        R.n_stored = 0;
        for (int p = 0; p < count; ++p) R.n_stored += (h.xoff[p + 1] - h.xoff[p]) - (h.rep > 1 ? (int64_t)(h.rep - 1) * h.rep_n[(size_t)p] : 0);
        // ---- the graphs' arrays, concatenated: everything into ONE pinned block and up in ONE transfer (a transfer per array
        //      and graph was ~200 copy dispatches for a 16-trial handle) ----
        const int64_t n_rel = rel_off[(size_t)count], n_rng = rng_off[(size_t)count], n_pri = pri_off[(size_t)count], n_pin = (int64_t)pin_edge.size();
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t i4 = sizeof(int32_t), f8 = sizeof(double);
        size_t off = 0;
        auto region = [&](size_t bytes) { const size_t o_ = off; off += al(std::max<size_t>(bytes, 8)); return o_; };
        // (graphs from the library's generator: those ten arrays are read where the generator left them -- no room for them here)
        const size_t gs = gen_src ? 0 : 1;
        const size_t o_rel_base = region(gs * n_rel * i4), o_rel_to = region(gs * n_rel * i4), o_rel_t = region(gs * n_rel * d * f8), o_rel_R = region(gs * n_rel * d * d * f8);
        const size_t o_rel_kappa = region(gs * n_rel * f8), o_rel_tau = region(gs * n_rel * f8);
        const size_t o_rng_a = region(gs * n_rng * i4), o_rng_b = region(gs * n_rng * i4), o_rng_dist = region(gs * n_rng * f8), o_rng_prec = region(gs * n_rng * f8);
        const size_t o_pri_lm = region(n_pri * i4), o_pri_t = region(n_pri * d * f8), o_pri_prec = region(n_pri * f8);
        const size_t o_relp = region(n_rel * i4), o_rngp = region(n_rng * i4), o_prip = region(n_pri * i4), o_pinp = region(n_pin * i4), o_pin = region(n_pin * i4);
        const size_t o_gp = region((size_t)count * sizeof(GaProb));
        const size_t pack_bytes = off;
        size_t got_h = pack_bytes;
        char* hb = (char*)block_cache().take(got_h, st.device, true);
        setup_pinned.push_back({hb, got_h});
        DevBuf<unsigned char> pack;
        pack.alloc(pack_bytes);
        auto cp = [&](size_t o_, const void* src, size_t bytes) { if (bytes) std::memcpy(hb + o_, src, bytes); };
        // (graph by graph on the host team: 5 MB of copies for a 16-trial handle)
        score::parallel_ranges((int64_t)count, 1, [&](int, int64_t p_lo, int64_t p_hi) {
        for (int p = (int)p_lo; p < (int)p_hi; ++p) {
            const score_graph& g = graphs[p];
            const size_t eo = (size_t)rel_off[(size_t)p], ro = (size_t)rng_off[(size_t)p], po = (size_t)pri_off[(size_t)p];
            if (!gen_src) {
            cp(o_rel_base + eo * i4, g.rel_base, (size_t)g.n_rel * i4); cp(o_rel_to + eo * i4, g.rel_to, (size_t)g.n_rel * i4);
            cp(o_rel_t + eo * d * f8, g.rel_t, (size_t)g.n_rel * d * f8); cp(o_rel_R + eo * d * d * f8, g.rel_R, (size_t)g.n_rel * d * d * f8);
            cp(o_rel_kappa + eo * f8, g.rel_kappa, (size_t)g.n_rel * f8); cp(o_rel_tau + eo * f8, g.rel_tau, (size_t)g.n_rel * f8);
            cp(o_rng_a + ro * i4, g.rng_a, (size_t)g.n_rng * i4); cp(o_rng_b + ro * i4, g.rng_b, (size_t)g.n_rng * i4);
            cp(o_rng_dist + ro * f8, g.rng_dist, (size_t)g.n_rng * f8); cp(o_rng_prec + ro * f8, g.rng_prec, (size_t)g.n_rng * f8);
            }
            cp(o_pri_lm + po * i4, g.lprior_lm, (size_t)g.n_lprior * i4); cp(o_pri_t + po * d * f8, g.lprior_t, (size_t)g.n_lprior * d * f8);
            cp(o_pri_prec + po * f8, g.lprior_prec, (size_t)g.n_lprior * f8);
            std::fill((int32_t*)(hb + o_relp) + eo, (int32_t*)(hb + o_relp) + eo + g.n_rel, p);
            std::fill((int32_t*)(hb + o_rngp) + ro, (int32_t*)(hb + o_rngp) + ro + g.n_rng, p);
            std::fill((int32_t*)(hb + o_prip) + po, (int32_t*)(hb + o_prip) + po + g.n_lprior, p);
            std::fill((int32_t*)(hb + o_pinp) + pin_off[(size_t)p], (int32_t*)(hb + o_pinp) + pin_off[(size_t)p + 1], p);
        }
        });
        cp(o_pin, pin_edge.data(), (size_t)n_pin * i4);
        cp(o_gp, gp.data(), (size_t)count * sizeof(GaProb));
        HIP_CHECK(hipMemcpyAsync(pack.d, hb, pack_bytes, hipMemcpyHostToDevice, stream));
        struct Slice { const void* d; };
        auto dI = [&](size_t o_) { return (const int32_t*)(pack.d + o_); };
        auto dF = [&](size_t o_) { return (const double*)(pack.d + o_); };
        struct { const int32_t* d; } rel_base{dI(o_rel_base)}, rel_to{dI(o_rel_to)}, rng_a{dI(o_rng_a)}, rng_b{dI(o_rng_b)}, pri_lm{dI(o_pri_lm)}, d_pin{dI(o_pin)},
            rel_prob{dI(o_relp)}, rng_prob{dI(o_rngp)}, pri_prob{dI(o_prip)}, pin_prob{dI(o_pinp)};
        struct { const double* d; } rel_t{dF(o_rel_t)}, rel_R{dF(o_rel_R)}, rel_kappa{dF(o_rel_kappa)}, rel_tau{dF(o_rel_tau)}, rng_dist{dF(o_rng_dist)},
            rng_prec{dF(o_rng_prec)}, pri_t{dF(o_pri_t)}, pri_prec{dF(o_pri_prec)};
        struct { const GaProb* d; } d_gp{(const GaProb*)(pack.d + o_gp)};
        // ---- records ----
        DevBuf<unsigned long long> key0;
        DevBuf<uint32_t> idx0;
        DevBuf<double> val;
        key0.alloc((size_t)rec_max); idx0.alloc((size_t)rec_max); val.alloc((size_t)rec_max);
        R.Ac.alloc((size_t)ae + 1); R.Av.alloc((size_t)ae + 1);
        R.qraw.alloc((size_t)n); R.braw.alloc((size_t)std::max<int64_t>(1, m));
        fill_zero_async(R.qraw.d, (size_t)n * sizeof(double), stream);
        GaArgs a{};
        a.d = d; a.relaxation = relax; a.count = count; a.probs = d_gp.d;
        a.rel_prob = rel_prob.d; a.rng_prob = rng_prob.d; a.pri_prob = pri_prob.d; a.pin_prob = pin_prob.d;
        a.n_rel = n_rel; a.n_rng = n_rng; a.n_pri = n_pri; a.n_pin = n_pin;
        a.rel_base = rel_base.d; a.rel_to = rel_to.d; a.rel_t = rel_t.d; a.rel_R = rel_R.d; a.rel_kappa = rel_kappa.d; a.rel_tau = rel_tau.d;
        a.rng_a = rng_a.d; a.rng_b = rng_b.d; a.rng_dist = rng_dist.d; a.rng_prec = rng_prec.d;
        if (gen_src) {
            a.rel_base = gen_src->rel_base; a.rel_to = gen_src->rel_to; a.rel_t = gen_src->rel_t; a.rel_R = gen_src->rel_R;
            a.rel_kappa = gen_src->rel_kappa; a.rel_tau = gen_src->rel_tau;
            a.rng_a = gen_src->rng_a; a.rng_b = gen_src->rng_b; a.rng_dist = gen_src->rng_dist; a.rng_prec = gen_src->rng_prec;
        }
        a.pri_lm = pri_lm.d; a.pri_t = pri_t.d; a.pri_prec = pri_prec.d; a.pin_edge = d_pin.d;
        a.key = key0.d; a.idx = idx0.d; a.val = val.d; a.pad_key = (unsigned long long)n << 32;
        a.A_ptr = A_ptr.d; a.A_col = R.Ac.d; a.A_val = R.Av.d; a.b = R.braw.d;
        if (n_rel) hipLaunchKernelGGL(k_ga_rel, dim3((unsigned)((n_rel + 255) / 256)), dim3(256), 0, stream, a);
        if (n_pin) hipLaunchKernelGGL(k_ga_pin, dim3((unsigned)((n_pin + 255) / 256)), dim3(256), 0, stream, a);
        if (n_rng) hipLaunchKernelGGL(k_ga_rng, dim3((unsigned)((n_rng + 255) / 256)), dim3(256), 0, stream, a);
        if (n_pri) hipLaunchKernelGGL(k_ga_pri, dim3((unsigned)((n_pri + 255) / 256)), dim3(256), 0, stream, a);
        {   // the 64 slots behind the records: padding
            DevBuf<long long> used;
            used.alloc(1);
            const long long u = o;
            std::vector<long long> uv(1, u);
            used.upload(uv);
            hipLaunchKernelGGL(k_rec_pad, dim3(1), dim3(256), 0, stream, key0.d, idx0.d, (const long long*)used.d, rec_max, n);
        }
        HIP_CHECK(hipGetLastError());
        merge_records(n, rec_max, key0, idx0, val.d, nullptr, kQCol, R.qraw.d, R.pm);
        R.P_ptr = R.pm.ptr.d; R.P_col = R.pm.col.d; R.P_val = R.pm.o0.d;
    }

    // From the raw program to A, G1, G2, K (K0 / K1), q, b, 1/D, 1/E on the device; the host gets the row pointers, K's columns
    // and the norms back and lays out the tiles.
    void setup_on_device(HostSystem& h, const score_problem* probs, const score_graph* graphs) {
        PhaseTimer pt(st.verbose != 0);
        const int count = h.count;
        const int64_t n = h.n_tot, m = h.m_tot;
        fill_tab(h);
        const ProbTab tab = prob_tab();
        setup_tmp.dev = st.device;
        RawDev R;
        ArenaSwap swap(&setup_tmp);
        if (!graphs) raw_from_problems(h, probs, R);
        else raw_from_graphs(h, graphs, R);
        const int64_t pe = R.pe, ae = R.ae, sq = R.sq, n_stored = R.n_stored;
        nnzP_full = R.nnzP_full;
        // ---- persistent arrays whose sizes are known up front ----
        {
            ArenaSwap persist(&arena);
            if (!graphs) A_ptr.upload(R.Aptr);
            A_col.alloc((size_t)ae + 64); A_val.alloc((size_t)ae + 64);
            q.alloc((size_t)n); b.alloc((size_t)m); invD.alloc((size_t)n); invE.alloc((size_t)m); Dd.alloc((size_t)n); Ed.alloc((size_t)m);
            G1.ptr.alloc((size_t)n + 1); G2.ptr.alloc((size_t)n + 1); G2.split.alloc((size_t)n);
            // (G1's and, on the graph path, G2's entries end where the kernels say: everything behind them must read as
            //  (column 0, value 0) -- one block, one fill)
            ZeroGroup zg;
            zg.add(G1.col, (size_t)ae + 64); zg.add(G1.val, (size_t)ae + 64);
            if (!R.exact) { zg.add(G2.col, (size_t)(nnzP_full + ae) + 64); zg.add(G2.val, (size_t)(nnzP_full + ae) + 64); }
            zg.commit(stream);
            if (R.exact) { G2.col.alloc((size_t)(nnzP_full + ae) + 64); G2.val.alloc((size_t)(nnzP_full + ae) + 64); }
        }
        const int64_t g2_nnz = nnzP_full + ae;  // (graph path: an upper bound)
        // (A's padding: k_g_scale_a writes it)
        if (R.exact) {
            UploadBatch fills;
            fill_zero_async(G2.col.d + g2_nnz, 64 * sizeof(int32_t), stream);
            fill_zero_async(G2.val.d + g2_nnz, 64 * sizeof(double), stream);
        }
        DevBuf<int32_t> atp, arow;
        DevBuf<uint32_t> idx0, atpos, akey0, akey1;
        DevBuf<double> dsc, esc, norms;
        const int32_t* const Pp_d = R.P_ptr; const int32_t* const Pc_d = R.P_col; const double* const Pv_d = R.P_val;
        const int32_t* const Ac_d = R.Ac.d; const double* const Av_d = R.Av.d;
        pt.mark(graphs ? "  device setup: graphs up, assembler queued" : "  device setup: raw matrices up");
        // ---- A' position map: entries of A by column, in row order (a stable sort of the columns) ----
        arow.alloc((size_t)std::max<int64_t>(1, ae)); atp.alloc((size_t)n + 1); atpos.alloc((size_t)std::max<int64_t>(1, ae));
        {
            RuizInitArgs ia{};
            ia.D = Dd.d; ia.E = Ed.d; ia.n = n; ia.m = m; ia.A_ptr = A_ptr.d; ia.arow = arow.d;
            hipLaunchKernelGGL(k_ruiz_init, dim3((unsigned)std::max<int64_t>(1, (std::max(n, m) + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, ia);
        }
        DevBuf<unsigned char> scratch;
        if (ae) {
            idx0.alloc((size_t)ae); akey0.alloc((size_t)ae); akey1.alloc((size_t)ae);
            hipLaunchKernelGGL(k_iota, dim3((unsigned)((ae + 255) / 256)), dim3(256), 0, stream, idx0.d, ae);
            HIP_CHECK(hipMemcpyAsync(akey0.d, Ac_d, (size_t)ae * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            int bits = 1;
            while (((int64_t)1 << bits) < n) ++bits;
            size_t tb = 0;
            HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, akey0.d, akey1.d, idx0.d, atpos.d, (size_t)ae, 0, bits, stream));
            scratch.alloc(tb + 256);
            HIP_CHECK(rocprim::radix_sort_pairs((void*)scratch.d, tb, akey0.d, akey1.d, idx0.d, atpos.d, (size_t)ae, 0, bits, stream));
        }
        hipLaunchKernelGGL(k_lower_bounds, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)akey1.d, ae, n, atp.d);
        HIP_CHECK(hipGetLastError());
        // ---- Ruiz passes ----
        const int64_t ngroups = (int64_t)h.cone_row.size();
        dsc.alloc((size_t)n); esc.alloc((size_t)std::max<int64_t>(1, ngroups));
        DevBuf<double> cmax;
        DevBuf<int32_t> long_rows, n_long_rows;
        {
            ZeroGroup zg;
            zg.add(cmax, (size_t)n); zg.add(norms, (size_t)4 * count); zg.add(n_long_rows, 1);
            zg.commit(stream);
        }
        RzArgs rz{};
        rz.cmax = cmax.d; rz.acol_sorted = akey1.d; rz.nnzA = ae;
        rz.P_ptr = Pp_d; rz.P_col = Pc_d; rz.P_val = Pv_d; rz.A_ptr = A_ptr.d; rz.A_col = Ac_d; rz.A_val = Av_d;
        rz.atp = atp.d; rz.atpos = atpos.d; rz.arow = arow.d; rz.gstart = cone_row.d;
        rz.D = Dd.d; rz.E = Ed.d; rz.d = dsc.d; rz.e = esc.d; rz.n = n; rz.m = m; rz.ngroups = ngroups; rz.tab = tab;
        const unsigned ggrp = (unsigned)std::max<int64_t>(1, (ngroups + 255) / 256);
        const unsigned gapp = (unsigned)std::max<int64_t>(1, (std::max(n, ngroups) + 255) / 256);
        for (int it = 0; it < std::max(0, st.scale_iters); ++it) {
            const unsigned ga_blocks = (unsigned)((ae + 255) / 256), gg_blocks = ngroups ? ggrp : 0u;
            if (ga_blocks + gg_blocks) hipLaunchKernelGGL(k_rz_colsA, dim3(ga_blocks + gg_blocks), dim3(256), 0, stream, rz, (int)ga_blocks);
            hipLaunchKernelGGL(k_rz_cols, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, rz);
            hipLaunchKernelGGL(k_rz_apply, dim3(gapp), dim3(256), 0, stream, rz);
        }
        HIP_CHECK(hipGetLastError());
        pt.mark("  device setup: A' map + equilibration queued");
        // ---- G1, G2, the equilibrated A, q, b, reciprocal scales, norms ----
        DevBuf<long long> len1, len2;
        len1.alloc((size_t)n + 1); len2.alloc((size_t)n + 1);
        GDevArgs ga{};
        ga.n = n; ga.m = m; ga.nnzA = ae; ga.tab = tab;
        ga.P_ptr = Pp_d; ga.P_col = Pc_d; ga.P_val = Pv_d; ga.A_col = Ac_d; ga.A_val = Av_d;
        ga.atp = atp.d; ga.atpos = atpos.d; ga.arow = arow.d; ga.D = Dd.d; ga.E = Ed.d;
        ga.len1 = len1.d; ga.len2 = len2.d; ga.g1_ptr = G1.ptr.d; ga.g2_ptr = G2.ptr.d; ga.g2_split = G2.split.d;
        ga.oA_col = A_col.d; ga.oA_val = A_val.d; ga.g1_col = G1.col.d; ga.g1_val = G1.val.d; ga.g2_col = G2.col.d; ga.g2_val = G2.val.d;
        ga.q_raw = R.qraw.d; ga.b_raw = R.braw.d; ga.q = q.d; ga.b = b.d; ga.invD = invD.d; ga.invE = invE.d;
        const unsigned grow1 = (unsigned)((n + 1 + 255) / 256);
        hipLaunchKernelGGL(k_g_lengths, dim3(grow1), dim3(256), 0, stream, ga);
        {
            size_t tb = 0;
            HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, len1.d, len1.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
            DevBuf<unsigned char> sc2;
            sc2.alloc(tb + 256);
            HIP_CHECK(rocprim::exclusive_scan((void*)sc2.d, tb, len1.d, len1.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
            HIP_CHECK(rocprim::exclusive_scan((void*)sc2.d, tb, len2.d, len2.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
        }
        hipLaunchKernelGGL(k_g_ptrs, dim3(grow1), dim3(256), 0, stream, ga);
        hipLaunchKernelGGL(k_g_scale_a, dim3((unsigned)std::max<int64_t>(1, (std::max(ae, std::max(n, m)) + 255) / 256)), dim3(256), 0, stream, ga);
        // rows of G2 by length: eight lanes for the usual ones, a wavefront for the listed long ones (k_row_classify)
        const int64_t long_cap = (nnzP_full + ae) / kLongRowEntries + 1;
        long_rows.alloc((size_t)long_cap);
        hipLaunchKernelGGL(k_row_classify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const int32_t*)G2.ptr.d, n, long_rows.d, n_long_rows.d);
        const unsigned g8 = (unsigned)((n + 31) / 32), g64 = (unsigned)((long_cap + 3) / 4);
        ga.long_rows = long_rows.d; ga.n_long_rows = n_long_rows.d;
        hipLaunchKernelGGL(k_g_fill<8>, dim3(g8), dim3(256), 0, stream, ga);
        hipLaunchKernelGGL(k_g_fill<64>, dim3(g64), dim3(256), 0, stream, ga);
        hipLaunchKernelGGL(k_prob_norms, dim3(count > 8 ? 8u : 32u, (unsigned)count), dim3(256), 0, stream, tab, (const double*)R.qraw.d, (const double*)q.d, (const double*)R.braw.d,
                           (const double*)b.d, norms.d);
        HIP_CHECK(hipGetLastError());
        // ---- K = P + sigma I + rho A'A as K0 + rho K1 ----
        const int64_t rec_max = n_stored + pe + sq + 64;
        DevBuf<long long> kcnt;
        DevBuf<unsigned long long> key0;
        DevBuf<uint32_t> kidx0;
        DevBuf<double> v0, v1;
        kcnt.alloc((size_t)n + 1); key0.alloc((size_t)rec_max); kidx0.alloc((size_t)rec_max); v0.alloc((size_t)rec_max); v1.alloc((size_t)rec_max);
        KBuildArgs ka{};
        ka.n = n; ka.tab = tab; ka.sigma = h.sigma;
        ka.g2_ptr = G2.ptr.d; ka.g2_split = G2.split.d; ka.g2_col = G2.col.d; ka.g2_val = G2.val.d;
        ka.A_ptr = A_ptr.d; ka.A_col = A_col.d; ka.A_val = A_val.d;
        ka.rec_cnt = kcnt.d; ka.key = key0.d; ka.idx = kidx0.d; ka.v0 = v0.d; ka.v1 = v1.d;
        ka.long_rows = long_rows.d; ka.n_long_rows = n_long_rows.d;
        hipLaunchKernelGGL(k_kb_count<8>, dim3(g8), dim3(256), 0, stream, ka);
        hipLaunchKernelGGL(k_kb_count<64>, dim3(g64), dim3(256), 0, stream, ka);
        {
            size_t tb = 0;
            HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, kcnt.d, kcnt.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
            DevBuf<unsigned char> sc2;
            sc2.alloc(tb + 256);
            HIP_CHECK(rocprim::exclusive_scan((void*)sc2.d, tb, kcnt.d, kcnt.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
        }
        hipLaunchKernelGGL(k_kb_expand<8>, dim3(g8), dim3(256), 0, stream, ka);
        hipLaunchKernelGGL(k_kb_expand<64>, dim3(g64), dim3(256), 0, stream, ka);
        hipLaunchKernelGGL(k_rec_pad, dim3((unsigned)((rec_max + 255) / 256)), dim3(256), 0, stream, key0.d, kidx0.d, (const long long*)(kcnt.d + n), rec_max, n);
        MergeOut mo;
        merge_records(n, rec_max, key0, kidx0, v0.d, v1.d, -1, nullptr, mo, (const long long*)kcnt.d);  // (kcnt: the rows' record offsets)
        pt.mark("  device setup: G1 / G2 / A / K queued");
        // ---- first trip back: counts, row pointers, norms ----
        using Pinned = PinnedBlock;
        const size_t o_res = 0, o_norm = 16, o_kp = o_norm + (size_t)4 * count * sizeof(double);
        const size_t o_g1 = o_kp + ((size_t)n + 1) * sizeof(int32_t), o_g2 = o_g1 + ((size_t)n + 1) * sizeof(int32_t);
        Pinned back(o_g2 + ((size_t)n + 1) * sizeof(int32_t), st.device);
        char* hb = (char*)back.p;
        HIP_CHECK(hipMemcpyAsync(hb + o_res, mo.result.d, 2 * sizeof(long long), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(hb + o_norm, norms.d, (size_t)4 * count * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(hb + o_kp, mo.ptr.d, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(hb + o_g1, G1.ptr.d, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipMemcpyAsync(hb + o_g2, G2.ptr.d, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(sync_stream(stream));
        pt.mark("  device setup: kernels + row pointers back");
        const int64_t nnzK = ((const long long*)(hb + o_res))[0];
        const double* nm = (const double*)(hb + o_norm);
        h.qnorm_u.assign((size_t)count, 0.0); h.qnorm_s.assign((size_t)count, 0.0); h.bnorm_u.assign((size_t)count, 0.0); h.bnorm_s.assign((size_t)count, 0.0);
        for (int p = 0; p < count; ++p) {
            h.qnorm_u[(size_t)p] = nm[4 * p]; h.qnorm_s[(size_t)p] = nm[4 * p + 1]; h.bnorm_u[(size_t)p] = nm[4 * p + 2]; h.bnorm_s[(size_t)p] = nm[4 * p + 3];
        }
        h.K.ptr.assign((const int32_t*)(hb + o_kp), (const int32_t*)(hb + o_kp) + n + 1);
        h.G1.ptr.assign((const int32_t*)(hb + o_g1), (const int32_t*)(hb + o_g1) + n + 1);
        h.G2.ptr.assign((const int32_t*)(hb + o_g2), (const int32_t*)(hb + o_g2) + n + 1);
        h.A.ptr = std::move(R.Aptr);
        if (h.K.ptr[(size_t)n] != nnzK || (R.exact && h.G2.ptr[(size_t)n] != g2_nnz)) throw std::runtime_error("device setup: inconsistent entry counts");
        nnzP_full = h.G2.ptr[(size_t)n] - ae;
        g1_nnz = h.G1.ptr[(size_t)n];
        // ---- K's persistent arrays at their exact size; its columns to the host (band layout, tiles) ----
        {
            ArenaSwap persist(&arena);
            K.ptr.alloc((size_t)n + 1); K.col.alloc((size_t)nnzK + 64); K.val.alloc((size_t)nnzK + 64);
            K0d.alloc((size_t)nnzK + 64); K1d.alloc((size_t)nnzK + 64);
        }
        HIP_CHECK(hipMemcpyAsync(K.ptr.d, mo.ptr.d, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        // (the 64 padding entries behind the last one come along: k_rec_total has zeroed them in the merge's arrays)
        HIP_CHECK(hipMemcpyAsync(K.col.d, mo.col.d, ((size_t)nnzK + 64) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(K0d.d, mo.o0.d, ((size_t)nnzK + 64) * sizeof(double), hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipMemcpyAsync(K1d.d, mo.o1.d, ((size_t)nnzK + 64) * sizeof(double), hipMemcpyDeviceToDevice, stream));
        fill_zero_async(K.val.d, K.val.n * sizeof(double), stream);
        {   // K's columns for the host's band layout: they stay in the pinned block they arrive in until the layout job (or, without
            // one, init) copies them into h.K.col -- 8 MB of resize + memcpy on this thread were 0.5 ms of a 16-trial create
            kcols_pin = std::make_shared<PinnedBlock>((size_t)std::max<int64_t>(1, nnzK) * sizeof(int32_t), st.device);
            kcols_n = nnzK;
            HIP_CHECK(hipMemcpyAsync(kcols_pin->p, mo.col.d, (size_t)nnzK * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            h.G1.nrows = h.G2.nrows = h.K.nrows = h.K.ncols = n;
            HIP_CHECK(sync_stream(stream));
        }
        make_system_rowblocks(h);
        fill_kkt_bytes(h);
        pt.mark("  device setup: K columns back, tiles");
    }

    struct PinnedBlock {
        void* p = nullptr; size_t bytes = 0; int dev;
        PinnedBlock(size_t b, int d) : bytes(b), dev(d) { p = block_cache().take(bytes, dev, true); }
        ~PinnedBlock() { block_cache().give(p, bytes, dev, true); }
        PinnedBlock(const PinnedBlock&) = delete;
        PinnedBlock& operator=(const PinnedBlock&) = delete;
    };
    std::shared_ptr<PinnedBlock> kcols_pin;  // K's columns as they came back from the device setup (adopt_k_columns)
    int64_t kcols_n = 0;
    void adopt_k_columns(HostSystem& h) {
        if (!kcols_pin) return;
        const int32_t* src = (const int32_t*)kcols_pin->p;
        h.K.col.assign(src, src + kcols_n);
        kcols_pin.reset();
    }
    void init(HostSystem& h, const score_settings& s_, const score_problem* probs = nullptr, const score_graph* graphs = nullptr) {
        H = &h;
        st = s_;
        PhaseTimer pt(st.verbose != 0);
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
        if (st.device < 0 || st.device >= ndev) throw std::runtime_error("score_settings.device out of range");
        HIP_CHECK(hipSetDevice(st.device));
        arena.dev = st.device;
        stream = stream_pool().take(st.device);
        tl_copy_stream = stream;
        struct ArenaScope {  // buffers allocated during init come from this handle's arena
            explicit ArenaScope(DevArena* a) { tl_arena = a; }
            ~ArenaScope() { tl_arena = nullptr; }
        } arena_scope(&arena);
        struct StageScope {  // setup uploads through pinned staging (StageArena); one synchronisation when the setup is over
            StageArena a;
            hipStream_t st;
            StageScope(int dev, hipStream_t s) : st(s) {
                a.dev = dev;
                if (stage_limit_bytes() > 0) tl_stage = &a;
            }
            ~StageScope() {
                tl_stage = nullptr;
                (void)sync_stream(st);  // (every queued transfer has left its pinned slot)
            }
        } stage_scope(st.device, stream);
        HIP_CHECK(hipEventCreate(&ev0));
        HIP_CHECK(hipEventCreate(&ev1));
        {   // split long rows finished by polling (CsrDev::long_spin) only where the whole launch is resident at once: every
            // SpMV kernel fits two workgroups per CU (<= 216 registers), the products with the Newton matrix six (<= 82)
            int cus = 0;
            HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st.device));
            K.spin_max_tiles = G1.spin_max_tiles = G2.spin_max_tiles = cus * 3 / 2;
            Hm.spin_max_tiles = cus * 4;
        }
        pt.mark("device + stream");
        if (h.bs != 0 && h.bs != 3 && h.bs != 4 && h.bs != 1 && h.bs != 2)
            throw std::runtime_error("unsupported block size");
        struct JoinPolish {  // an exception below must not leave the builder running against a dying handle
            std::future<void>& f;
            ~JoinPolish() { if (f.valid()) f.wait(); }
        } join_polish{polish_build};
        // the Newton matrix pattern and its contribution lists only read the finished host system:
        // built on another thread while this one uploads (4.3 ms beside 2.4 ms of uploads / allocations)
        // The Newton matrix: its pattern and contribution lists are built on the device (score_polish_device.hpp) from the
        // matrices uploaded below -- unless its band view is asked for or SCORE_HOST_POLISH_BUILD is set: then on another
        // host thread while this one uploads.  Either way the structure check (per-cone data) runs on that thread.
        polish_on_device = st.polish && !band_h(h) && std::getenv("SCORE_HOST_POLISH_BUILD") == nullptr;
        if (h.device_setup && !probs && !graphs) throw std::runtime_error("device setup: the raw problems are missing");
        if (st.polish && !h.device_setup)  // (device setup: the per-cone structure comes from a kernel, init_polish)
            polish_build = std::async(std::launch::async, [this, &h] {
                if (polish_on_device) { Q = PolishData(); polish_structure(h, Q); }
                else build_polish(h, Q, st.verbose != 0, band_h(h));
            });
        std::future<BandLayout> band_layout_job;
        auto start_band_layout = [&] {
        if (band_k(h) && !h.chains.empty())  // the band view of K only reads the finished host system: laid out on a thread of its own
            band_layout_job = std::async(std::launch::async, [&h, this] {
                BuildScope scope;
                adopt_k_columns(h);  // (device setup: the columns are still in their pinned block)
                std::vector<char> use(h.chains.size());
                for (size_t ci = 0; ci < h.chains.size(); ++ci) use[ci] = h.chain_owner[ci] == (int32_t)ci;
                std::vector<RowSegment> sg;
                if (h.rep > 1) {
                    for (int p = 0; p < h.count; ++p) {
                        const int64_t nr = h.rep_n[(size_t)p];
                        sg.push_back(RowSegment{h.xoff[p], h.xoff[p] + nr, p, (int32_t)nr});
                        sg.push_back(RowSegment{h.xoff[p] + (int64_t)h.rep * nr, h.xoff[p + 1], p, 0});
                    }
                } else {
                    sg = plain_segments(h.xoff);
                }
                return build_band_layout(h.K, sg, band_runs(h.chains, use, h.bs, h.rep, h.rep_n, true), h.bs, h.count);
            });
        };
        if (!h.device_setup) start_band_layout();
        struct JoinBand {  // (an exception below must not leave the job running against a dying handle)
            std::future<BandLayout>& f;
            ~JoinBand() { if (f.valid()) f.wait(); }
        } join_band{band_layout_job};
        if (h.device_setup) {
            // everything matrix-shaped is built on the device from the raw problems (score_setup_device.hpp); the cone table
            // (its rows are the equilibration's groups) goes up first
            cone_row.upload(h.cone_row); cone_dim.upload(h.cone_dim); cone_type.upload(h.cone_type);
            setup_on_device(h, probs, graphs);
            start_band_layout();
            if (!band_layout_job.valid()) adopt_k_columns(h);  // (no layout job to do it)
            { UploadBatch ub; K.adopt_tiles(h.K, h.rbK); G1.adopt_tiles(h.G1, h.rbG1); G2.adopt_tiles(h.G2, h.rbG2); }
            pt.mark("  device setup");
        } else
        K.upload(h.K, h.rbK, nullptr, false);  // (values: K0 + rho K1, on the device -- derive_rho_data)
        // A single problem whose equilibration ran on the device: the equilibrated A, G1 = A' and G2 = [P | A'] are derived
        // there from the raw matrices and scales those passes left behind (k_derive_a / k_derive_g; 26 MB of uploads less
        // for the headline problem) -- a replicated problem only when its replicas' P values are bit-equal to replica 0's.
        derive_ag = !h.device_setup && h.count == 1 && h.m_tot > 0 && ruiz_dev.kept && ruiz_dev.k_n == h.n_tot && ruiz_dev.k_m == h.m_tot &&
                    ruiz_dev.k_nnzA == (int64_t)h.A.col.size() && ruiz_dev.k_rep == h.rep && (h.rep == 1 || h.rep_exact);
        struct DropKept {  // (the kept buffers go back when the setup is over, whatever happens -- once nothing reads them any more)
            RuizDevice& r;
            hipStream_t st;
            ~DropKept() {
                if (r.kept) (void)sync_stream(st);
                r.drop();
            }
        } drop_kept{ruiz_dev, stream};
        if (!h.device_setup) {
        G1.upload(h.G1, h.rbG1, nullptr, !derive_ag, !derive_ag);
        G2.upload(h.G2, h.rbG2, &h.g2_split, !derive_ag, !derive_ag);
        }
        pt.mark("  uploads: K, G1, G2");
        // replicated problems (HostSystem::rep): K and G1 = A' hold replica 0's rows; K's operands repeat with the
        // block's replica stride, G1's are the consecutive tail rows of a cone
        K.rep = h.rep; K.rs_in = 0;
        G1.rep = h.rep; G1.rs_in = 1;
        K.unroll = (h.rep > 1) ? h.tile_nnz / kThreads : kUnroll;
        G1.unroll = (h.rep > 1) ? kUnroll / 2 : kUnroll;  // (tiles of at most kTileNnz / 2 nonzeros, see build_system)
        {
            std::vector<int32_t> vf, ve, vp;
            for (int p = 0; p < h.count; ++p)
                for (int64_t r = h.xoff[p]; r < h.xoff[p + 1]; r += kThreads) {
                    vf.push_back((int32_t)r); ve.push_back((int32_t)std::min<int64_t>(r + kThreads, h.xoff[p + 1])); vp.push_back(p);
                }
            n_vblocks = (int)vf.size();
            UploadBatch ub;
            vb_first.upload(vf); vb_end.upload(ve); vb_prob.upload(vp);
        }
        if (!h.device_setup) A_ptr.upload(h.A.ptr);
        // padded like the SpMV matrices: the cone kernel clamps its unconditional loads
        if (h.device_setup) {
        } else if (derive_ag) {
            const size_t nz = h.A.col.size();
            A_col.alloc(nz + 64); A_val.alloc(nz + 64);
            { UploadBatch fills; fill_zero_async(A_col.d + nz, 64 * sizeof(int32_t), stream); fill_zero_async(A_val.d + nz, 64 * sizeof(double), stream); }
            DeriveArgs da{};
            da.n = h.n_tot; da.m = h.m_tot; da.nnzA = (int64_t)nz; da.rep = h.rep; da.nr = h.rep > 1 ? h.rep_n[0] : 0;
            da.P_ptr = ruiz_dev.Pp.d; da.P_col = ruiz_dev.Pc.d; da.P_val = ruiz_dev.Pv.d;
            da.A_ptr = ruiz_dev.Ap.d; da.A_col = ruiz_dev.Ac.d; da.A_val = ruiz_dev.Av.d;
            da.atp = ruiz_dev.dat.d; da.atpos = ruiz_dev.dpos.d; da.arow = ruiz_dev.drow.d; da.D = ruiz_dev.dD.d; da.E = ruiz_dev.dE.d;
            da.oA_col = A_col.d; da.oA_val = A_val.d;
            da.g1_ptr = G1.ptr.d; da.g1_col = G1.col.d; da.g1_val = G1.val.d;
            da.g2_ptr = G2.ptr.d; da.g2_split = G2.split.d; da.g2_col = G2.col.d; da.g2_val = G2.val.d;
            if (nz) hipLaunchKernelGGL(k_derive_a, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, stream, da);
            hipLaunchKernelGGL(k_derive_g, dim3((unsigned)((h.n_tot + 3) / 4)), dim3(256), 0, stream, da);
            HIP_CHECK(hipGetLastError());
        } else {
            A_col.upload_padded(h.A.col, 64); A_val.upload_padded(h.A.val, 64);
        }
        if (!h.device_setup) { q.upload(h.q); b.upload(h.b); }
        if (h.device_setup) {
        } else if (derive_ag) {  // (D and E are on the device: their reciprocals too)
            invD.alloc(h.D.size()); invE.alloc(h.E.size());
            const int64_t nm = std::max<int64_t>((int64_t)h.D.size(), (int64_t)h.E.size());
            hipLaunchKernelGGL(k_derive_inv, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, stream, (const double*)ruiz_dev.dD.d, (const double*)ruiz_dev.dE.d,
                               invD.d, invE.d, (int64_t)h.D.size(), (int64_t)h.E.size());
            HIP_CHECK(hipGetLastError());
        } else {
            std::vector<double> iD(h.D.size()), iE(h.E.size());
            parallel_ranges((int64_t)iD.size(), 32768, [&](int, int64_t i0, int64_t i1) { for (int64_t i = i0; i < i1; ++i) iD[(size_t)i] = 1.0 / h.D[(size_t)i]; });
            parallel_ranges((int64_t)iE.size(), 32768, [&](int, int64_t i0, int64_t i1) { for (int64_t i = i0; i < i1; ++i) iE[(size_t)i] = 1.0 / h.E[(size_t)i]; });
            invD.upload(iD); invE.upload(iE);
        }
        pt.mark("  uploads: A, q, b, 1/D, 1/E");
        if (!h.device_setup) { cone_row.upload(h.cone_row); cone_dim.upload(h.cone_dim); cone_type.upload(h.cone_type); }
        { UploadBatch ub; cone_block_first.upload(h.cone_block_first); cone_block_prob.upload(h.cone_block_prob); }
        {   // the cone tables (row pointers, the entries of the small cones again by cone index): from A on the device
            const size_t nc = h.cone_row.size();
            cone_meta.alloc(2 * nc); cone_cols.alloc(8 * nc); cone_vals.alloc(8 * nc);
            if (nc) {
                ConeTabArgs ca{};
                ca.ncones = (int64_t)nc; ca.cone_row = cone_row.d; ca.cone_dim = cone_dim.d; ca.cone_type = cone_type.d;
                ca.A_ptr = A_ptr.d; ca.A_col = A_col.d; ca.A_val = A_val.d;
                ca.meta = cone_meta.d; ca.cols = cone_cols.d; ca.vals = cone_vals.d;
                hipLaunchKernelGGL(k_cone_tables, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, ca);
                HIP_CHECK(hipGetLastError());
            }
        }
        n_cone_blocks = (int)h.cone_block_prob.size();
        pt.mark("  uploads: cone records");
        {   // (the chain tables below: one run of uploads, nothing launched in between -- UploadBatch)
        UploadBatch ub_tables;
        {
            std::vector<int2> lg;
            for (int bl = 0; bl < n_cone_blocks; ++bl)
                for (int c = h.cone_block_first[bl]; c < h.cone_block_first[bl + 1]; ++c)
                    if (h.cone_dim[c] > kWaveCone) lg.push_back(make_int2(c, h.cone_block_prob[bl]));
            n_large_cones = (int)lg.size();
            if (n_large_cones) cone_large.upload(lg);
        }
        node_col.upload(h.node_col); diag_cols.upload(h.diag_cols);
        prec_work.upload(h.prec_work); chains.upload(h.chains); levels.upload(h.levels);
        fac_rangeK.upload(h.fac_range); fac_rangeH.upload(h.fac_range_H);
        factor_work.upload(h.factor_work);
        if (h.rep > 1) { chainsH.upload(h.chainsH); levelsH.upload(h.levelsH); }
        else { chainsH.view(chains.d, chains.n); levelsH.view(levels.d, levels.n); }
        {   // the records k_prec_pre reads (PrecRecord), for the factors of K and for those of the Newton matrix
            auto build = [&](const std::vector<ChainDesc>& cs, const std::vector<ChainLevelDesc>& ls) {
                std::vector<PrecRecord> rec(h.prec_work.size());
                std::memset((void*)rec.data(), 0, rec.size() * sizeof(PrecRecord));
                for (size_t w = 0; w < rec.size(); ++w) {
                    rec[w].wk = h.prec_work[w];
                    if (rec[w].wk.kind != 0) continue;
                    rec[w].ch = cs[(size_t)rec[w].wk.index];
                    for (int l = 0; l < std::min<int>(rec[w].ch.n_levels, kRecLevels); ++l) rec[w].lv[l] = ls[(size_t)rec[w].ch.level_begin + l];
                }
                // update helpers (PrecArgs::split_update): a single problem's chains occupy a fraction of the CUs for the
                // whole launch, and a third of what each of them pulls through its CU is the xt / kx update.  Extra
                // records after the problem's own hand that update, slice by slice, to workgroups on the idle CUs.
                // (A batch fills the chip with chains: there the update stays fused.)
                PrecRecord hr;
                std::memset((void*)&hr, 0, sizeof(hr));
                for (int i = 0; i < n_help; ++i) {
                    const int64_t e0 = (int64_t)i * kHelpEntries;
                    hr.wk = PrecWork{2, (int32_t)e0, (int32_t)std::min<int64_t>(kHelpEntries, h.n_tot - e0), 0};
                    rec.push_back(hr);
                }
                return rec;
            };
            {
                int cus = 0;
                HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st.device));
                const int want = (int)((h.n_tot + kHelpEntries - 1) / kHelpEntries);
                const bool on = h.count == 1 && (int)h.prec_work.size() + want <= cus;  // the whole launch resident at once
                n_help = on ? want : 0;
            }
            prec_rec.upload(build(h.chains, h.levels));
            if (h.rep > 1 && st.polish) prec_recH.upload(build(h.chainsH, h.levelsH));
            else prec_recH.view(prec_rec.d, prec_rec.n);
        }
        n_prec = (int)h.prec_work.size();
        active_part_ptr = h.prec_part_ptr;
        {   // split chain kernel: only when the whole launch is resident at once (one workgroup per CU)
            int cus = 0;
            HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, st.device));
            const bool off = st.chain_split <= 0 || h.rep > 1;
            if (!off && cus > 0) build_split_system(h, cus, split);
            if (split.active && split.stage_rel.size() > 0) {
                for (const auto& pl : split.plans)
                    if (pl.n_stage > kWaveThreads * kWaveStage || (pl.lv[0].n + 1) * 3 > kWaveThreads * kWaveVec) split.active = false;
            }
            if (split.active) {
                n_prec = (int)split.work.size();
                active_part_ptr = split.part_ptr;
                split_work.upload(split.work); split_items.upload(split.items); split_plans.upload(split.plans);
                split_stage.upload(split.stage_rel.empty() ? std::vector<int32_t>(1, -1) : split.stage_rel);
                split_xbuf.alloc((size_t)std::max(1, split.n_slots) * kSplitMaxParts * kSplitSlotDoubles); split_xbuf.zero(stream);
                split_xflag.alloc((size_t)std::max(1, split.n_slots) * kSplitMaxParts); split_xflag.zero(stream);
                split_epoch.alloc(split.work.size()); split_epoch.zero(stream);
                // more than half of a CU's LDS: at most one of these workgroups per CU (the hand-off between the
                // parts of a chain is sized and measured for that)
                split_lds = std::max<size_t>(split.max_lds_doubles * sizeof(double), (size_t)84 * 1024);
                HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_wave<PREC_INIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
                HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_wave<PREC_STEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
                int khz = 0;
                HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, st.device));
                split_poll_limit = (unsigned long long)std::max(1, khz) * 50ull;  // 50 ms
                if (split.max_lds_doubles * sizeof(double) > (size_t)150 * 1024) split.active = false;
            }
            if (!split.active) { n_prec = (int)h.prec_work.size(); active_part_ptr = h.prec_part_ptr; }
        }
        prec_part_ptr.upload(active_part_ptr);
        }
        if (st.verbose)
            std::fprintf(stderr, "[score setup] chain preconditioner: %s, %d work items (%zu chains)\n",
                         split.active ? "split (one wavefront per chain part)" : "one workgroup per chain", n_prec, h.chains.size());
        pt.mark("uploads");
        // chain vectors in LDS: level 0 (N nodes) when it fits, always the coarse levels
        int max_nodes = 0, max_all = 0;
        for (const auto& ch : h.chains) {
            max_nodes = std::max(max_nodes, ch.scratch_nodes);
            max_all = std::max(max_all, ch.scratch_nodes + ch.N);
        }
        const size_t lds_all = (16 + (size_t)max_all * std::max(1, h.bs)) * sizeof(double);
        const size_t lds_up = (16 + (size_t)max_nodes * std::max(1, h.bs)) * sizeof(double);
        prec_lds0 = lds_all <= 144 * 1024;
        prec_lds = prec_lds0 ? lds_all : lds_up;
        // k_prec_pre (level 0 in registers, coarser levels staged into LDS): every chain needs
        // <= 256 level-0 runs, its vector in one chunk of loads, coarse factors that fit the
        // staging registers, and everything within the LDS budget
        prec_pre = h.bs >= 1 && h.bs <= 4 && !h.chains.empty();
        size_t lds_pre = 0;
        const int pre_chunk = h.bs >= 4 ? 8 : kPrecChunk;          // PreTile<BS>::CH
        const size_t deep_esz = h.bs >= 4 ? sizeof(float) : sizeof(double);  // coarse-level factors in LDS (k_prec_pre: LT)
        for (const auto& ch : h.chains) {
            const ChainLevelDesc* lv = &h.levels[ch.level_begin];
            int64_t deep = 0;
            if (ch.n_levels >= 2) {
                const ChainLevelDesc& Lz = lv[ch.n_levels - 1];
                deep = Lz.offB + (int64_t)2 * h.bs * h.bs * Lz.N - lv[1].offR;
            }
            const int ng = std::max(8 * h.bs * h.bs, 32);
            // (k_prec_pre computes a node's column as col0 + node * stride: chains with irregular columns take k_prec)
            if (ch.col_stride == 0 || ch.n_levels > kRecLevels || lv[0].nruns > kPreRunLanes || (ch.n_levels >= 2 && lv[1].N > kPrecThreads - kPreRunLanes) || (int64_t)ch.N * h.bs > (int64_t)pre_chunk * kPrecThreads ||
                deep > (int64_t)ng * (kPrecThreads - kPreRunLanes))
                prec_pre = false;
            const ChainLevelDesc& Lend = lv[ch.n_levels - 1];
            const size_t vec_doubles = (size_t)Lend.lds_off + (size_t)Lend.N * h.bs + 1;
            lds_pre = std::max(lds_pre, (16 + vec_doubles + (size_t)(lv[0].nruns + 1) * h.bs) * sizeof(double) + (((size_t)deep + 1) & ~(size_t)1) * deep_esz);
        }
        if (lds_pre > 158 * 1024) prec_pre = false;  // 160 KiB per CU, minus the static record and slack
        prec_pre_lds = prec_pre ? lds_pre : 0;
        // register-resident coarse levels (k_prec_pre<.., float, true>): LDS holds the vectors only
        prec_reg = prec_pre && h.deep_ok && h.bs <= 3 && st.fac_fp32 != 0;
        if (prec_reg) {
            size_t lds_reg = 0;
            for (const auto& ch : h.chains) {
                const ChainLevelDesc* lv = &h.levels[ch.level_begin];
                const ChainLevelDesc& Lend = lv[ch.n_levels - 1];
                lds_reg = std::max(lds_reg, (16 + (size_t)Lend.lds_off + (size_t)Lend.N * h.bs + 1 + (size_t)(lv[0].nruns + 1) * h.bs) * sizeof(double));
            }
            prec_reg_lds = lds_reg;
            deep_map.upload(h.deep_map);
        }
        {   // does any launch fall back to the streaming kernel (k_prec)?  4 x 4 blocks: every factor set kept in double
            const bool fallback = !prec_pre || (h.bs >= 4 && st.fac_fp32 == 0);
            if (fallback && prec_lds > 144 * 1024) throw std::runtime_error("chain too long: more than 129 segments of 1023 nodes (132 k) for the segmented chain solver, and beyond what the streaming kernel keeps in LDS");
        }
        if (n_prec_chains(h)) {
            if (h.bs <= 1) allow_big_lds<1>(); else if (h.bs == 2) allow_big_lds<2>();
            else if (h.bs == 3) allow_big_lds<3>(); else allow_big_lds<4>();
        }
        if (st.polish) init_polish_build(h);
        pt.mark("polish: structure (before the band view is waited for)");
        if (band_layout_job.valid()) {  // band view of K (score_band.hpp), laid out beside everything above
            Kb.upload(band_layout_job.get());
            if (Kb.on)
                for (int p = 0; p < h.count; ++p) h.kkt_bytes[(size_t)p] = Kb.L.bytes[(size_t)p] + 16.0 * (double)(h.xoff[p + 1] - h.xoff[p]);
            if (st.verbose)
                std::fprintf(stderr, "[score setup] band view of K: %s (%d band + %d csr + %d diag tiles, %d slots per row)\n", Kb.on ? "on" : "off",
                             Kb.L.n_band, Kb.L.n_csr, Kb.L.n_diag, Kb.L.S);
            pt.mark("band view of K (wait + upload)");
        }
        kblk_part_ptr.upload(Kb.on ? Kb.L.part_ptr : h.rbK.part_ptr);
        {   // the iterates and partial sums a reset zeroes: ONE block (a reset is one fill instead of fifteen -- each a 4 us
            // dispatch, on every solve)
            auto pad = [](size_t c) { return (std::max<size_t>(1, c) + 31) & ~(size_t)31; };  // (256-byte aligned pieces)
            const size_t nm = pad((size_t)(h.n_tot + h.m_tot)), nn = pad((size_t)h.n_tot), mm = pad((size_t)h.m_tot), cc = pad((size_t)h.count);
            const size_t kb = pad((size_t)kblocks()), np_ = pad((size_t)n_prec);
            iter_block.alloc(2 * nm + mm + 6 * nn + cc + kb + 4 * np_);
            double* o = iter_block.d;
            auto take = [&](DevBuf<double>& b, size_t count, size_t padded) { b.view(o, count); o += padded; };
            take(xtu, (size_t)(h.n_tot + h.m_tot), nm); take(xy, (size_t)(h.n_tot + h.m_tot), nm); take(s, (size_t)h.m_tot, mm);
            take(r, (size_t)h.n_tot, nn); take(z, (size_t)h.n_tot, nn); take(p, (size_t)h.n_tot, nn); take(p2, (size_t)h.n_tot, nn);
            take(w, (size_t)h.n_tot, nn); take(kx, (size_t)h.n_tot, nn); take(step, (size_t)h.count, cc);
            take(pw_part, (size_t)kblocks(), kb); take(rz_part0, (size_t)n_prec, np_); take(rz_part1, (size_t)n_prec, np_);
            take(rz_meas0, (size_t)n_prec, np_); take(rz_meas1, (size_t)n_prec, np_);
        }
        cg_iters = st.cg_iters;
        std::vector<int32_t> dz(h.count, 0);
        done.upload(dz);
        if (!h.device_setup) { K0d.upload_padded(h.K0, 64); K1d.upload_padded(h.K1, 64); }
        {   // positions of the chain blocks and the Jacobi diagonals in K's value array: looked up on the device (binary
            // search per block entry; on the host this was 0.5 ms of a headline create and 2.4 of an 8-trial handle's)
            const int b2 = h.bs * h.bs;
            kposd.alloc(h.node_col.size() * (size_t)b2); kposs.alloc(h.node_col.size() * (size_t)b2); kdiagpos.alloc(h.diag_cols.size());
            DevBuf<int32_t> prevc, drow;
            prevc.upload(h.node_prev_owned); drow.upload(h.diag_row0);
            HPosArgs pa{};
            pa.Hptr = K.ptr.d; pa.Hcol = K.col.d; pa.node_col = node_col.d; pa.prev_col = prevc.d;
            pa.n_nodes = (int64_t)h.node_col.size(); pa.bs = h.bs; pa.pos_diag = kposd.d; pa.pos_sub = kposs.d;
            pa.diag_cols = drow.d; pa.n_diag = (int64_t)h.diag_cols.size(); pa.diag_pos = kdiagpos.d;
            const int64_t npos = std::max<int64_t>(pa.n_nodes * b2, pa.n_diag);
            if (npos > 0) hipLaunchKernelGGL(k_hb_positions, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, stream, pa);
            HIP_CHECK(hipGetLastError());
        }
        join_init(h);
        // (separator slots of the spike region are never written, nor used: zeroed once, with the float copy, in one fill)
        ZeroGroup zfac;
        zfac.add(fac, h.fac_doubles);
        use_fac32 = st.fac_fp32 != 0;
        // 4 x 4 blocks (3-D problems): the LDS-resident chain kernel only exists for the 4-byte stream, and the streaming
        // kernel is three times slower (22 / 30 us against 66 / 74 us per application on 1000-pose chains) -- the Newton
        // factors follow the ADMM ones there (same Newton and PCG counts on the 3-D BASELINE-sized problems)
        // Long chains (every chain >= 256 nodes: the BASELINE sizes) take the 4-byte stream for the Newton factors as well:
        // same Newton and PCG counts there, and the register-resident chain kernel serves the Newton PCG too (headline
        // default solve 6.1 -> 5.7 ms).  Short chains with stiff pinned-pose terms keep double Newton factors (+10 % PCG
        // iterations with floats on the 144-graph sweep of round 2).
        int min_chain = 1 << 30;
        for (const auto& ch : h.chains) min_chain = std::min(min_chain, (int)ch.N);
        newton_fac32 = st.fac_fp32 >= 2 || (st.fac_fp32 == 1 && prec_pre && (h.bs >= 4 || min_chain >= 256));
        if (use_fac32) zfac.add(fac32, h.fac_doubles);
        zfac.commit(stream);
        if (use_fac32 && prec_reg) deepK.alloc((size_t)std::max<int64_t>(1, h.deep_floats));
        dinv.alloc(h.dinv.size()); rho.upload(h.rho);
        q_work.alloc((size_t)std::max<int64_t>(1, h.scratch_nodes) * 2 * std::max(1, h.bs * h.bs));
        plan_factor_lds();
        derive_rho_data(false);
        pt.mark("allocations + rho data (device)");
        if (st.polish) init_polish(h);
        link_init(h, probs, graphs);
        pt.mark("polish setup");
        {   // the report arena (see `rep`) and the control block (see `ctl`)
            const size_t n_pres = (size_t)std::max(1, n_cone_blocks) * kPartStride, n_dres = (size_t)G2.nblocks * kPartStride;
            const size_t n_int = (2 * (size_t)h.count + 1) / 2;  // 2 * count int32
            const size_t n_meas = 2 * (size_t)std::max(1, n_prec);
            const size_t total = n_pres + n_dres + n_fpart + n_gd + n_meas + 2 * n_int + 1;
            h_rep_bytes = total * sizeof(double);
            h_rep = (double*)block_cache().take(h_rep_bytes, st.device, true);
            std::memset(h_rep, 0, total * sizeof(double));
            double* d_rep = nullptr;
            HIP_CHECK(hipHostGetDevicePointer((void**)&d_rep, h_rep, 0));
            rep.view(d_rep, total);
            rep_dres_off = n_pres;
            size_t o = 0;
            pres_part.view(d_rep + o, n_pres); h_pres = h_rep + o; o += n_pres;
            dres_part.view(d_rep + o, n_dres); h_dres = h_rep + o; o += n_dres;
            q_fpart.view(d_rep + o, n_fpart); h_newton = h_rep + o; o += n_fpart;
            q_gd.view(d_rep + o, n_gd); h_gd = h_rep + o; o += n_gd;
            h_meas = h_rep + o; d_meas = d_rep + o; o += n_meas;
            h_gate = (int32_t*)(h_rep + o); d_gate_host = (int32_t*)(d_rep + o); o += n_int;
            h_gate_live = (int32_t*)(h_rep + o); d_gate_live = (int32_t*)(d_rep + o); o += n_int;  // (written by pcg_gate as the PCG runs)
            h_seq = (unsigned long long*)(h_rep + o); d_seq = (unsigned long long*)(d_rep + o);
            // device-resident words the kernels read: gate flags and counts, control block
            q_pcgdone.alloc(2 * (size_t)h.count);
            q_pcgdone.zero(stream);
            q_gate_used.view(q_pcgdone.d + h.count, h.count);
            ctl.alloc(2 * (size_t)h.count + (3 * (size_t)h.count + 1) / 2);  // [step | tol2 | skip, reref, fskip]
            q_step.view(ctl.d, h.count);
            q_gate_tol2.view(ctl.d + h.count, h.count);
            q_skip.view((int32_t*)(ctl.d + 2 * (size_t)h.count), h.count);
            q_reref.view((int32_t*)(ctl.d + 2 * (size_t)h.count) + h.count, h.count);
            q_fskip.view((int32_t*)(ctl.d + 2 * (size_t)h.count) + 2 * (size_t)h.count, h.count);
        }
        reset();
        pt.mark("reset");
        HIP_CHECK(sync_stream(stream));
        setup_tmp.release_all();  // (nothing in flight reads the setup's scratch any more)
        for (auto& pb : setup_pinned) block_cache().give(pb.first, pb.second, st.device, true);
        setup_pinned.clear();
        pt.mark("staged uploads: drain");
    }

    // Everything that depends on the penalties, on the device and in stream order: K = K0 + rho K1 on
    // the fixed pattern, the nested-dissection factors of the chains of K (the same k_factor the Newton
    // polish uses for its Hessian) and the reciprocal Jacobi diagonal.
    void derive_rho_data(bool upload_penalties) {
        const HostSystem& h = *H;
        if (upload_penalties) {
            double* v = (double*)next_ring_slot(h.count * sizeof(double));
            for (int p = 0; p < h.count; ++p) v[p] = h.rho[p];
            fetch_words((int32_t*)rho.d, (const char*)v, 2 * h.count);
        }
        hipLaunchKernelGGL(k_kval, dim3(K.nblocks), dim3(kThreads), 0, stream, K.dev(), (const double*)K0d.d, (const double*)K1d.d,
                           (const double*)rho.d, K.val.d, (const int32_t*)nullptr);
        if (Kb.on) {
            const int64_t nnz = (int64_t)h.K.col.size();
            hipLaunchKernelGGL(k_band_pack, dim3((unsigned)((nnz + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, (const int32_t*)Kb.dst.d,
                               (const double*)K.val.d, Kb.V.d, nnz);
        }
        if (!h.factor_work.empty()) {
            FactorArgs fa{};
            fa.work = factor_work.d; fa.chains = chains.d; fa.levels = levels.d; fa.Hval = K.val.d;
            fa.pos_diag = kposd.d; fa.pos_sub = kposs.d; fa.fac = fac.d; fa.work_mat = q_work.d; fa.skip = nullptr;
            fa.diag_pos = kdiagpos.d; fa.dinv = dinv.d;
            launch_factor(fa, (int)h.factor_work.size(), false);
        }
        HIP_CHECK(hipGetLastError());
    }
    // ---- segmented long chains (score_join.hpp) ----
    // tables of the long chains + the positions of the separators' blocks in K; the Newton matrix's come with init_polish
    void join_init(const HostSystem& h) {
        n_join_items = (int)h.join_items.size(); n_join_chains = (int)h.join_chains.size(); n_join_seps = (int)h.join_sep_col.size();
        if (!n_join_items) return;
        for (const JoinChain& jc : h.join_chains)
            if (jc.n_seg - 1 > kJoinMaxSeps) throw std::runtime_error("internal: a segmented chain with more separators than the join kernel holds (build_system keeps such chains whole)");
        std::optional<UploadBatch> ub_join;
        ub_join.emplace();
        join_jc.upload(h.join_chains); join_items.upload(h.join_items);
        join_sep_col.upload(h.join_sep_col); join_sep_diag.upload(h.join_sep_diag);
        // pseudo-nodes for the position look-up (k_hb_positions): [separator | prev = last node of the segment before it] and
        // [first node of the segment after it | prev = separator]
        std::vector<int32_t> pc(2 * (size_t)n_join_seps), pp(2 * (size_t)n_join_seps);
        for (const JoinChain& jc : h.join_chains)
            for (int sg = 0; sg + 1 < jc.n_seg; ++sg) {
                const ChainDesc& cl = h.chains[(size_t)jc.first_chain + sg];
                const ChainDesc& cr = h.chains[(size_t)jc.first_chain + sg + 1];
                const size_t sp = (size_t)jc.sep_begin + sg;
                const int32_t b = h.join_sep_col[sp];
                pc[2 * sp] = b; pp[2 * sp] = h.node_col[(size_t)cl.node_begin + cl.N - 1];
                pc[2 * sp + 1] = h.node_col[(size_t)cr.node_begin]; pp[2 * sp + 1] = b;
            }
        join_pcol.upload(pc); join_pprev.upload(pp);
        ub_join.reset();
        const size_t b2 = (size_t)h.bs * h.bs, nb = 2 * (size_t)h.bs;
        ZeroGroup zj;
        zj.add(join_W_K, nb * (size_t)h.n_tot); zj.add(join_rhs, (size_t)h.n_tot); zj.add(join_data_K, 5 * b2 * (size_t)n_join_seps);
        zj.add(join_zero, (size_t)h.count); zj.add(join_zb, (size_t)h.bs * n_join_seps);
        zj.commit(stream);
        join_tmp_p.alloc((size_t)h.n_tot); join_tmp_rz.alloc(h.prec_work.size() + 4096);  // (every workgroup of a chain-kernel launch has a slot: work items + update helpers)
        join_positions(K.ptr.d, K.col.d, join_posd_K, join_poss_K);
        if (st.verbose) std::fprintf(stderr, "[score setup] long chains: %d in %d segments of <= %d nodes (second level: score_join.hpp)\n",
                                     n_join_chains, n_join_items, seg_max_nodes());
    }
    void join_positions(const int32_t* ptr, const int32_t* col, DevBuf<int32_t>& posd, DevBuf<int32_t>& poss) {
        const size_t b2 = (size_t)H->bs * H->bs;
        posd.alloc(2 * (size_t)n_join_seps * b2); poss.alloc(2 * (size_t)n_join_seps * b2);
        HPosArgs pa{};
        pa.Hptr = ptr; pa.Hcol = col; pa.node_col = join_pcol.d; pa.prev_col = join_pprev.d;
        pa.n_nodes = 2 * (int64_t)n_join_seps; pa.bs = H->bs; pa.pos_diag = posd.d; pa.pos_sub = poss.d;
        pa.diag_cols = nullptr; pa.n_diag = 0; pa.diag_pos = nullptr;
        hipLaunchKernelGGL(k_hb_positions, dim3((unsigned)((pa.n_nodes * (int64_t)b2 + 255) / 256)), dim3(256), 0, stream, pa);
        HIP_CHECK(hipGetLastError());
    }
    void join_init_newton(const HostSystem& h) {
        if (!n_join_items) return;
        const size_t b2 = (size_t)h.bs * h.bs, nb = 2 * (size_t)h.bs;
        ZeroGroup zj;
        zj.add(join_W_H, nb * (size_t)h.n_tot); zj.add(join_data_H, 5 * b2 * (size_t)n_join_seps);
        zj.commit(stream);
        join_positions(Hm.ptr.d, Hm.col.d, join_posd_H, join_poss_H);
    }
    JoinArgs join_args(bool newton_set) {
        JoinArgs ja{};
        ja.jc = join_jc.d; ja.items = join_items.d; ja.chains = newton_set ? chainsH.d : chains.d; ja.node_col = node_col.d;
        ja.sep_col = join_sep_col.d; ja.sep_nodes = join_pcol.d; ja.sep_prev = join_pprev.d; ja.done = join_zero.d;
        ja.use_owner = (!newton_set && H->rep > 1) ? 1 : 0;
        ja.W = newton_set ? join_W_H.d : join_W_K.d; ja.n_tot = H->n_tot;
        ja.data = newton_set ? join_data_H.d : join_data_K.d;
        ja.val = newton_set ? Hm.val.d : K.val.d;
        ja.pos_diag = newton_set ? join_posd_H.d : join_posd_K.d; ja.pos_sub = newton_set ? join_poss_H.d : join_poss_K.d;
        ja.rhs = join_rhs.d; ja.zb = join_zb.d;
        ja.sep_diag = join_sep_diag.d; ja.dinv = newton_set ? q_dinv.d : dinv.d; ja.n_sep_entries = (int)H->join_sep_diag.size();
        return ja;
    }
    template <int BS>
    void join_refresh_bs(bool newton_set) {
        JoinArgs ja = join_args(newton_set);
        hipLaunchKernelGGL(k_join_dinv, dim3((unsigned)((ja.n_sep_entries + kJoinThreads - 1) / kJoinThreads)), dim3(kJoinThreads), 0, stream, ja);
        // the spikes: 2 BS applications of the chain kernel alone to the coupling columns
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = newton_set ? chainsH.d : chains.d; pa.levels = newton_set ? levelsH.d : levels.d;
        pa.rec = newton_set ? prec_recH.d : prec_rec.d; pa.fac = newton_set ? q_fac.d : fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = newton_set ? q_dinv.d : dinv.d; pa.done = join_zero.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = newton_set ? q_hblk_part.d : kblk_part_ptr.d;
        pa.uni = uni_for(newton_set ? hblocks() : kblocks());
        pa.r = join_rhs.d; pa.r_in = join_rhs.d; pa.p = join_tmp_p.d; pa.w = w.d; pa.xt = join_tmp_p.d; pa.kx = join_tmp_p.d;
        pa.pw_part = nullptr; pa.rz_in = nullptr; pa.rz_out = join_tmp_rz.d;
        join_suspend = true;
        for (int c = 0; c < 2 * BS; ++c) {
            ja.column = c;
            hipLaunchKernelGGL(k_join_rhs<BS>, dim3((unsigned)((n_join_items * BS + kJoinThreads - 1) / kJoinThreads)), dim3(kJoinThreads), 0, stream, ja, n_join_items);
            pa.z = ja.W + (size_t)c * (size_t)H->n_tot;
            launch_prec<PREC_INIT>(pa);
        }
        join_suspend = false;
        hipLaunchKernelGGL(k_join_schur<BS>, dim3((unsigned)((n_join_chains + 63) / 64)), dim3(64), 0, stream, ja, n_join_chains);
    }
    void join_refresh(bool newton_set) {
        if (!n_join_items) return;
        switch (H->bs) {
            case 1: join_refresh_bs<1>(newton_set); break;
            case 2: join_refresh_bs<2>(newton_set); break;
            case 3: join_refresh_bs<3>(newton_set); break;
            default: join_refresh_bs<4>(newton_set); break;
        }
    }
    template <int BS, int MODE>
    void join_apply_bs(const PrecArgs& pa, bool newton_set) {
        JoinArgs ja = join_args(newton_set);
        ja.done = pa.done;
        ja.r = (MODE == PREC_INIT) ? pa.r_in : pa.r;
        ja.z = pa.z; ja.p = pa.p; ja.rz_out = pa.rz_out;
        unsigned nv = 1;
        if (MODE == PREC_INIT && pa.n_vec > 1) {  // (score_link.hpp's refresh: every round's vector in one launch)
            nv = (unsigned)pa.n_vec;
            ja.n_vec = pa.n_vec; ja.vec_stride = pa.vec_stride; ja.zb_stride = (long long)H->bs * n_join_seps; ja.rz_stride = (long long)n_prec;
            ja.zb = link_zb.d;
        }
        hipLaunchKernelGGL(k_join_solve<BS>, dim3((unsigned)n_join_chains, nv), dim3(64), 0, stream, ja);
        hipLaunchKernelGGL((k_join_apply<BS, MODE>), dim3((unsigned)n_join_items, nv), dim3(kJoinThreads), 0, stream, ja);
    }
    template <int MODE>
    void join_apply(const PrecArgs& pa, bool newton_set) {
        switch (H->bs) {
            case 1: join_apply_bs<1, MODE>(pa, newton_set); break;
            case 2: join_apply_bs<2, MODE>(pa, newton_set); break;
            case 3: join_apply_bs<3, MODE>(pa, newton_set); break;
            default: join_apply_bs<4, MODE>(pa, newton_set); break;
        }
    }

    // ---- loop closures inside the preconditioners (score_link.hpp): Woodbury correction of the chain solve, for the ADMM
    //      loop's K (set 0) and for the Newton matrix (set 1) -- one plan, the sets' own positions, columns Z and matrices Q ----
    LinkPlan link_plan;
    int n_link_items = 0, n_link_probs = 0, n_link_u = 0, link_rounds = 0, link_max_u = 0;
    bool link_suspend = false;
    bool link_set_on[2] = {false, false};
    DevBuf<LinkProb> link_probs;
    DevBuf<LinkItem> link_items;
    DevBuf<int32_t> link_ucol, link_uround, link_usuper, link_pcol[2], link_pshift[2], link_pos[2], link_zero, link_status[2];
    DevBuf<uint8_t> link_mask;
    DevBuf<double> link_Qt[2], link_Zr[2], link_t, link_rhs, link_tmp_p, link_tmp_rz, link_zb;
    void link_init(const HostSystem& h, const score_problem* probs, const score_graph* graphs) {
        n_link_items = n_link_probs = n_link_u = link_rounds = 0;
        link_set_on[0] = link_set_on[1] = false;
        if (h.chains.empty() || split.active || std::getenv("SCORE_NO_LINKS") != nullptr) return;
        std::vector<int32_t> pairs;
        if (graphs) find_link_pairs_graphs(h, graphs, pairs);
        else if (probs) find_link_pairs_P(h, probs, pairs);
        if (pairs.empty()) return;
        make_link_plan(h, pairs, link_plan);
        const LinkPlan& L = link_plan;
        if (st.verbose) std::fprintf(stderr, "[score setup] loop closures: %d node pairs outside the chains, %d inside the preconditioners (%d unknowns in %d groups, %d chains, %d rounds)\n",
                                     L.pairs_total, L.pairs_used, (int)L.ucol.size(), (int)L.probs.size(), (int)L.items.size(), L.rounds);
        if (L.empty()) return;
        n_link_items = (int)L.items.size(); n_link_probs = (int)L.probs.size(); n_link_u = (int)L.ucol.size(); link_rounds = L.rounds;
        link_max_u = 0;
        for (const LinkProb& P : L.probs) link_max_u = std::max(link_max_u, (int)P.n_u);
        std::optional<UploadBatch> ub_link;
        ub_link.emplace();
        link_probs.upload(L.probs); link_items.upload(L.items);
        link_ucol.upload(L.ucol); link_uround.upload(L.uround); link_usuper.upload(L.usuper); link_mask.upload(L.mask);
        const bool sets[2] = {true, st.polish != 0 && Q.available};
        {   // where an unknown's row is looked up: K of a replicated problem holds replica 0's rows, the Newton matrix every row
            std::vector<int32_t> prob_of((size_t)n_link_u, 0);
            for (const LinkProb& P : L.probs)
                for (int a = 0; a < P.n_u; ++a) prob_of[(size_t)P.u_begin + a] = P.prob;
            for (int set = 0; set < 2; ++set) {
                if (!sets[set]) continue;
                std::vector<int32_t> pc((size_t)n_link_u), ps((size_t)n_link_u, 0);
                for (int u = 0; u < n_link_u; ++u) pc[(size_t)u] = set == 0 ? link_owner_col(h, prob_of[(size_t)u], L.ucol[(size_t)u], &ps[(size_t)u]) : L.ucol[(size_t)u];
                link_pcol[set].upload(pc); link_pshift[set].upload(ps);
            }
        }
        ub_link.reset();
        ZeroGroup zl;
        for (int set = 0; set < 2; ++set) {
            if (!sets[set]) continue;
            link_pos[set].alloc(L.mask.size());
            zl.add(link_status[set], (size_t)n_link_probs); zl.add(link_Qt[set], L.mask.size()); zl.add(link_Zr[set], (size_t)link_rounds * (size_t)h.n_tot);
        }
        zl.add(link_t, (size_t)n_link_u);
        zl.add(link_rhs, (size_t)link_rounds * (size_t)h.n_tot); zl.add(link_zero, (size_t)h.count);
        zl.commit(stream);
        // (one launch applies the chain kernel to every round's right-hand side: a vector of n_tot per round)
        link_tmp_p.alloc((size_t)link_rounds * (size_t)h.n_tot); link_tmp_rz.alloc((size_t)link_rounds * (h.prec_work.size() + 4096));
        if (n_join_seps) link_zb.alloc((size_t)link_rounds * (size_t)h.bs * (size_t)n_join_seps);
        HIP_CHECK(hipFuncSetAttribute((const void*)k_link_cap<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kLinkMaxU * kLinkMaxU * (int)sizeof(double)));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_link_cap<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kLinkMaxU * kLinkMaxU * (int)sizeof(double)));
        for (int set = 0; set < 2; ++set) {
            if (!sets[set]) continue;
            link_set_on[set] = true;
            LinkArgs la = link_args(set == 1);
            hipLaunchKernelGGL(k_link_positions, dim3(16), dim3(kLinkThreads), 0, stream, la, n_link_probs);
        }
        LinkArgs la = link_args(false);
        hipLaunchKernelGGL(k_link_rhs, dim3((unsigned)((n_link_u + kLinkThreads - 1) / kLinkThreads)), dim3(kLinkThreads), 0, stream, la);
        HIP_CHECK(hipGetLastError());
        link_refresh(false);  // (K's chains were factored before the plan existed; the Newton set's follow its first factorisation)
    }
    LinkArgs link_args(bool newton_set) {
        const int set = newton_set ? 1 : 0;
        LinkArgs la{};
        la.probs = link_probs.d; la.items = link_items.d; la.ucol = link_ucol.d; la.uround = link_uround.d; la.usuper = link_usuper.d;
        la.mask = link_mask.d; la.pcol = link_pcol[set].d; la.pshift = link_pshift[set].d; la.pos = link_pos[set].d;
        la.Qt = link_Qt[set].d; la.t = link_t.d; la.Zr = link_Zr[set].d; la.rhs = link_rhs.d;
        la.n_tot = H->n_tot; la.rounds = link_rounds; la.n_u_total = n_link_u;
        if (newton_set) { la.Hptr = Hm.ptr.d; la.Hcol = Hm.col.d; la.Hval = Hm.val.d; }
        else { la.Hptr = K.ptr.d; la.Hcol = K.col.d; la.Hval = K.val.d; }
        la.chains = newton_set ? chainsH.d : chains.d; la.node_col = node_col.d; la.done = link_zero.d; la.status = link_status[set].d;
        return la;
    }
    // after every factorisation of a set's chains: the columns Z = T^-1 U (one application of the chain kernel -- second level
    // included -- to every round's right-hand side at once) and Q = (I + G Z[U,:])^-1 G
    void link_refresh(bool newton_set) {
        if (!n_link_items || !link_set_on[newton_set ? 1 : 0]) return;
        LinkArgs la = link_args(newton_set);
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = newton_set ? chainsH.d : chains.d; pa.levels = newton_set ? levelsH.d : levels.d;
        pa.rec = newton_set ? prec_recH.d : prec_rec.d; pa.fac = newton_set ? q_fac.d : fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = newton_set ? q_dinv.d : dinv.d; pa.done = link_zero.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = newton_set ? q_hblk_part.d : kblk_part_ptr.d;
        pa.uni = uni_for(newton_set ? hblocks() : kblocks());
        pa.r = link_rhs.d; pa.r_in = link_rhs.d; pa.p = link_tmp_p.d; pa.w = w.d; pa.xt = link_tmp_p.d; pa.kx = link_tmp_p.d;
        pa.pw_part = nullptr; pa.rz_in = nullptr; pa.rz_out = link_tmp_rz.d;
        // (the right-hand sides -- unit vectors, round r's in vector r -- were written once, at link_init: the chain kernel does
        //  not touch r_in)
        pa.z = la.Zr;
        link_suspend = true;
        if (link_rounds == 1) launch_prec<PREC_INIT>(pa);
        else {
            pa.n_vec = link_rounds; pa.vec_stride = (long long)H->n_tot;
            launch_prec<PREC_INIT>(pa);
        }
        link_suspend = false;
        // (up to 48 unknowns: Gauss-Jordan on one wavefront, no block barriers; beyond: four wavefronts)
        if (link_max_u <= 48) hipLaunchKernelGGL(k_link_cap<true>, dim3((unsigned)n_link_probs), dim3(256), (size_t)2 * link_max_u * link_max_u * sizeof(double), stream, la);
        else hipLaunchKernelGGL(k_link_cap<false>, dim3((unsigned)n_link_probs), dim3(256), (size_t)2 * link_max_u * link_max_u * sizeof(double), stream, la);
    }
    template <int BS, int MODE>
    void link_apply_bs(const PrecArgs& pa, bool newton_set) {
        LinkArgs la = link_args(newton_set);
        la.done = pa.done;
        la.r = (MODE == PREC_INIT) ? pa.r_in : pa.r;
        la.z = pa.z; la.p = pa.p; la.rz_out = pa.rz_out;
        if (link_plan.max_items <= kLinkGroupItems) {  // (a workgroup per group: t and the group's chains in one launch)
            hipLaunchKernelGGL((k_link_group<BS, MODE>), dim3((unsigned)n_link_probs), dim3(kLinkApplyThreads), 0, stream, la);
            return;
        }
        hipLaunchKernelGGL(k_link_solve, dim3((unsigned)n_link_probs), dim3(128), 0, stream, la);
        hipLaunchKernelGGL((k_link_apply<BS, MODE>), dim3((unsigned)n_link_items), dim3(kLinkApplyThreads), 0, stream, la);
    }
    template <int MODE>
    void link_apply(const PrecArgs& pa, bool newton_set) {
        switch (H->bs) {
            case 1: link_apply_bs<1, MODE>(pa, newton_set); break;
            case 2: link_apply_bs<2, MODE>(pa, newton_set); break;
            case 3: link_apply_bs<3, MODE>(pa, newton_set); break;
            default: link_apply_bs<4, MODE>(pa, newton_set); break;
        }
    }

    int n_prec_items() const { return (int)H->prec_work.size(); }
    // k_factor keeps the level-to-level matrices in LDS when the longest chain fits
    int factor_lds_wmat = 0;
    size_t factor_lds_bytes = 0;
    void plan_factor_lds() {
        const HostSystem& h = *H;
        const int b2 = std::max(1, h.bs * h.bs);
        int max_scr = 0, max_runs = 0;
        for (const auto& ch : h.chains) {
            max_scr = std::max(max_scr, ch.scratch_nodes);
            max_runs = std::max(max_runs, h.levels[ch.level_begin].nruns);
        }
        const size_t wm = (size_t)max_scr * 2 * b2, vx = (size_t)(max_runs + 2) * b2;
        factor_lds_wmat = 0; factor_lds_bytes = 0;
        if (!h.chains.empty() && max_runs <= kThreads && (wm + vx) * sizeof(double) <= (size_t)150 * 1024) {
            factor_lds_wmat = (int)std::max<size_t>(wm, 1);
            factor_lds_bytes = (wm + vx + 2) * sizeof(double);
            HIP_CHECK(hipFuncSetAttribute((const void*)k_factor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_factor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_factor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_factor<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        }
    }
    void launch_factor_kernels(const FactorArgs& fa, int np) {
        const int bs = H->bs;
        if (bs <= 1) hipLaunchKernelGGL(k_factor<1>, dim3(np), dim3(kThreads), factor_lds_bytes, stream, fa);
        else if (bs == 2) hipLaunchKernelGGL(k_factor<2>, dim3(np), dim3(kThreads), factor_lds_bytes, stream, fa);
        else if (bs == 3) hipLaunchKernelGGL(k_factor<3>, dim3(np), dim3(kThreads), factor_lds_bytes, stream, fa);
        else hipLaunchKernelGGL(k_factor<4>, dim3(np), dim3(kThreads), factor_lds_bytes, stream, fa);
    }
    // newton_set: the factors of the Newton matrix (every chain its own: chainsH / levelsH / q_fac)
    void launch_factor(FactorArgs fa, int np, bool newton_set) {
        if (np == 0) return;
        fa.lds_wmat = factor_lds_wmat;
        const int bs = H->bs;
        if (np > 65535) throw std::runtime_error("too many preconditioner work items for one handle (65535): split the batch");
        const int64_t nf = (int64_t)(newton_set ? H->fac_doubles_H : H->fac_doubles);
        // When every application of this factor set goes through k_prec_pre<.., float> (the 4-byte stream), k_factor writes the
        // float copy itself and nobody reads the doubles: no rounding launch (33 us a time on the headline problem, seven times
        // per default solve).  The streaming kernel reads `fac`: then the factors are rounded in place as before.
        const bool want32 = (newton_set ? newton_fac32 : use_fac32) && nf > 0;
        const bool direct32 = want32 && prec_pre && !split.active;
        fa.fac32 = direct32 ? (newton_set ? q_fac32.d : fac32.d) : nullptr;
        launch_factor_kernels(fa, np);
        if (want32) {
            float* shadow = newton_set ? q_fac32.d : fac32.d;
            if (!direct32)
            // (chain by chain, honouring the launch's skip flags: the Newton polish re-factors only the problems whose
            //  active set has moved, and rounding the whole array again cost as much as re-factoring them)
            hipLaunchKernelGGL(k_fac_round_items, dim3(16, (unsigned)np), dim3(kThreads), 0, stream, fa.work,
                               (const int64_t*)(newton_set ? fac_rangeH.d : fac_rangeK.d), fa.skip, fa.fac, shadow);
            if (prec_reg)  // ... and the lane-major copy of the coarse levels the register-resident chain kernel loads
                hipLaunchKernelGGL(k_deep_pack, dim3((unsigned)(deep_padded_slots(bs) / 4), (unsigned)np), dim3(kThreads), 0, stream, fa.work, fa.chains,
                                   fa.levels, (const int32_t*)deep_map.d, (const float*)shadow, newton_set ? deepH.d : deepK.d, fa.skip, bs * bs);
        }
        join_refresh(newton_set);  // (segmented long chains: separators' inverse diagonals, spikes, Schur factors)
        link_refresh(newton_set);  // (loop closures: the Woodbury columns and the capacitance matrix, score_link.hpp)
    }

    ConeArgs cone_args(const double* gathered) {
        ConeArgs a{};
        a.A_ptr = A_ptr.d; a.A_col = A_col.d; a.A_val = A_val.d;
        a.cone_row = cone_row.d; a.cone_dim = cone_dim.d; a.cone_type = cone_type.d; a.cone_meta = cone_meta.d;
        a.cone_cols = (const int4*)cone_cols.d; a.cone_vals = (const double2*)cone_vals.d;
        a.uniform_cones = (H->count == 1) ? (int)H->cone_row.size() : 0;
        a.block_first = cone_block_first.d; a.block_prob = cone_block_prob.d;
        a.done = done.d; a.rho = rho.d; a.b = b.d; a.xt = gathered;
        a.s = s.d; a.y = xy.d + H->n_tot; a.u = xtu.d + H->n_tot;
        a.s_in = a.s; a.y_in = a.y;
        a.alpha_relax = st.alpha; a.invE = invE.d; a.pres_part = pres_part.d;
        a.apply_alpha = 0; a.pfin = p.d; a.pw_in = pw_part.d; a.rz_in = rz_part0.d;
        a.prec_part_ptr = prec_part_ptr.d; a.kblk_part_ptr = kblk_part_ptr.d; a.step_out = step.d;
        a.uni = uni_for(kblocks());
        a.skip_large = n_large_cones > 0 ? 1 : 0;
        return a;
    }

    void upload_rho(const HostSystem& h) {
        (void)h;
        derive_rho_data(true);
        {   // K changed: the carried product kx = K xt is recomputed once
            SpmvArgs a = spmv_args(K, xtu.d);
            a.p = xtu.d; a.w = kx.d;
            launch_spmv<MODE_KP>(K, a);
            HIP_CHECK(hipGetLastError());
        }
        if (n_cone_blocks) {
            hipLaunchKernelGGL(k_refresh_u, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
            HIP_CHECK(hipGetLastError());
        }
    }

    // Stream-ordered uploads of small per-problem arrays without a host synchronisation and without
    // the copy engine: the words go through a ring of host-mapped pinned slots and a one-workgroup
    // kernel fetches them (every wait_published() drains the stream long before the ring wraps).
    char* next_ring_slot(size_t bytes) {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (need > ring_slot_bytes) {
            HIP_CHECK(sync_stream(stream));
            block_cache().give(h_ring, h_ring_bytes, st.device, true);
            h_ring = nullptr;
            ring_slot_bytes = std::max<size_t>(need, 256);
            h_ring_bytes = kFlagSlots * ring_slot_bytes;
            h_ring = (char*)block_cache().take(h_ring_bytes, st.device, true);
            HIP_CHECK(hipHostGetDevicePointer((void**)&d_ring, h_ring, 0));
            ring_used = 0;
        }
        if (++ring_used >= kFlagSlots) {  // about to reuse a slot a queued fetch may not have read yet
            HIP_CHECK(sync_stream(stream));
            ring_used = 1;
        }
        flag_slot = (flag_slot + 1) % kFlagSlots;
        return h_ring + (size_t)flag_slot * ring_slot_bytes;
    }
    void fetch_words(int32_t* dev_dst, const char* slot, int n_words) {
        const int32_t* src = (const int32_t*)(d_ring + (slot - h_ring));
        hipLaunchKernelGGL(k_fetch, dim3(1), dim3(kThreads), 0, stream, src, dev_dst, n_words);
    }
    void set_done(const std::vector<int>& d) {
        int32_t* v = (int32_t*)next_ring_slot(d.size() * sizeof(int32_t));
        for (size_t i = 0; i < d.size(); ++i) v[i] = d[i];
        fetch_words(done.d, (const char*)v, (int)d.size());
    }
    // Publish: push up to three small device arrays into the host-mapped arena and raise the
    // sequence number; wait: spin on it (falls back to a stream synchronisation, which also surfaces
    // device errors, when it does not arrive within 2 s).
    unsigned long long publish(const void* s0 = nullptr, void* d0 = nullptr, size_t w0 = 0, const void* s1 = nullptr, void* d1 = nullptr,
                               size_t w1 = 0, const void* s2 = nullptr, void* d2 = nullptr, size_t w2 = 0) {
        PushArgs a{};
        a.src[0] = (const unsigned long long*)s0; a.dst[0] = (unsigned long long*)d0; a.n[0] = (int)w0;
        a.src[1] = (const unsigned long long*)s1; a.dst[1] = (unsigned long long*)d1; a.n[1] = (int)w1;
        a.src[2] = (const unsigned long long*)s2; a.dst[2] = (unsigned long long*)d2; a.n[2] = (int)w2;
        a.flag = d_seq; a.seq = ++seq_next;
        hipLaunchKernelGGL(k_push, dim3(1), dim3(kThreads), 0, stream, a);
        return a.seq;
    }
    // The host's side of a publish: a short spin with the CPU's pause hint (a result that is about to land is picked up within
    // a fraction of a microsecond), then yields (the waiting thread gives its core to whoever can run: with several ranks per
    // node and several driver threads per rank the host has fewer cores than waiters -- a bare spin there burns exactly the
    // CPU quota the other ranks' setup needs), then short sleeps; a stream synchronisation, which also surfaces device errors,
    // when nothing arrives within 2 s.
    void wait_published(unsigned long long seq) {
        HIP_CHECK(hipGetLastError());
        struct WaitTimer {
            HipBackend* b; double t0;
            ~WaitTimer() { b->wait_ms += now_ms() - t0; b->waits += 1; }
        } wait_timer{this, now_ms()};
        const auto t0 = std::chrono::steady_clock::now();
        auto t_spin0 = t0;
        unsigned spins = 0;
        int phase = 0;  // 0 spin, 1 yield (alone) / economy sleeps, 2 sleep
        long long slept_ns = 0;
        while (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) < seq) {
            if (phase == 0) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                if ((++spins & 63) != 0) continue;
            } else if (phase == 1) {
                if (economy_waits()) {
                    const auto s0 = std::chrono::steady_clock::now();
                    economy_sleep();
                    slept_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - s0).count();
                } else {
                    sched_yield();
                    if ((++spins & 15) != 0) continue;
                }
            } else {
                const auto s0 = std::chrono::steady_clock::now();
                struct timespec ts = {0, 50000};  // 50 us
                nanosleep(&ts, nullptr);
                slept_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - s0).count();
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            const double spin_us = economy_waits() ? 10.0 : 30.0;
            if (phase == 0 && us > spin_us) phase = 1;
            else if (phase == 1 && us > 2000.0 && !economy_waits()) phase = 2;
            else if (us > 2e6) {
                release_prequeued();  // (a fetch waiting for the host's words would keep the stream from draining)
                HIP_CHECK(sync_stream(stream));
                if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) < seq) throw std::runtime_error("device did not publish its results");
                break;
            }
        }
        {
            const long long total_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_spin0).count();
            wait_stats().spin_ns.fetch_add(std::max(0LL, total_ns - slept_ns), std::memory_order_relaxed);  // (sleeps account for themselves)
            wait_stats().waits.fetch_add(1, std::memory_order_relaxed);
        }
        ring_used = pre_slot ? 1 : 0;  // everything queued before the publish has run: those ring slots are free again
    }

    DevBuf<double> iter_block;  // xtu | xy | s | r | z | p | p2 | w | kx | step | pw_part | rz_part0/1 | rz_meas0/1
    void reset() {
        const double t0 = st.verbose ? now_ms() : 0.0;
        if (st.verbose) HIP_CHECK(sync_stream(stream));
        const double t1 = st.verbose ? now_ms() : 0.0;
        iter_block.zero(stream);
        const double t2 = st.verbose ? now_ms() : 0.0;
        if (st.verbose) HIP_CHECK(sync_stream(stream));
        const double t3 = st.verbose ? now_ms() : 0.0;
        if (n_cone_blocks) {
            hipLaunchKernelGGL(k_refresh_u, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(sync_stream(stream));
        if (st.verbose) std::fprintf(stderr, "[score] reset: %.3f ms waiting for work queued earlier, fill of %.1f MB: %.3f ms call + %.3f ms wait, refresh %.3f ms\n", t1 - t0,
                                     (double)iter_block.n * 8e-6, t2 - t1, t3 - t2, now_ms() - t3);
    }

    // Launch on the handle's stream.  Inside score_time_iteration (tev != nullptr, slot >= 0) the
    // launch carries a start/stop event pair: the runtime then reports the dispatch's own begin/end
    // timestamps (what a profiler's kernel trace shows) without inserting a command between two
    // kernels of the loop.
    hipEvent_t* tev = nullptr;
    template <class Kern, class... Args>
    void launch_on_stream(Kern kernel, dim3 grid, dim3 block, size_t lds_bytes, int slot, Args... args) {
        if (tev && slot >= 0)
            hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds_bytes, stream, tev[2 * slot], tev[2 * slot + 1], 0, args...);
        else
            hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, args...);
    }

    template <int MODE>
    void launch_prec(const PrecArgs& pa_in, int slot = -1) {
        if (n_prec == 0) return;
        PrecArgs pa = pa_in;
        const bool newton_set = (pa.fac == q_fac.d) && q_fac.d;
        pa.fac32 = newton_set ? q_fac32.d : fac32.d;  // the float copy of whichever factor set is applied
        pa.deep = newton_set ? deepH.d : deepK.d;
        const bool use_fac32 = newton_set ? newton_fac32 : this->use_fac32;
        if (split.active) {
            if (pa.n_vec > 1) throw std::runtime_error("internal: several right-hand sides through the split chain kernel");
            WaveArgs wa{};
            wa.p = pa;
            wa.p.work = split_work.d;
            wa.items = split_items.d; wa.plans = split_plans.d; wa.stage_rel = split_stage.d;
            wa.xbuf = split_xbuf.d; wa.xflag = split_xflag.d; wa.epoch = split_epoch.d; wa.poll_limit = split_poll_limit;
            launch_on_stream(k_prec_wave<MODE>, dim3(n_prec), dim3(kWaveThreads), split_lds, slot, wa);
        } else {
            const int bs = H->bs;
            if (bs <= 1) launch_prec_bs<1, MODE>(pa, slot, use_fac32);
            else if (bs == 2) launch_prec_bs<2, MODE>(pa, slot, use_fac32);
            else if (bs == 3) launch_prec_bs<3, MODE>(pa, slot, use_fac32);
            else launch_prec_bs<4, MODE>(pa, slot, use_fac32);
        }
        // segmented long chains: the second level (score_join.hpp) after every application of the chain kernel
        if (n_join_items && !join_suspend && !pa.debug_skip) join_apply<MODE>(pa, newton_set);
        // loop closures (score_link.hpp): the Woodbury correction of the Newton set's chain solve
        if (n_link_items && link_set_on[newton_set ? 1 : 0] && !link_suspend && !join_suspend && !pa.debug_skip) link_apply<MODE>(pa, newton_set);
    }
    template <int BS, int MODE>
    void launch_prec_bs(const PrecArgs& pa_in, int slot, bool use_fac32) {
        // (the k_prec_pre launches of a STEP carry the update helpers of a single-problem handle, see the records)
        PrecArgs pa = pa_in;
        const bool help = MODE == PREC_STEP && n_help > 0 && pa.rec != nullptr && !pa.debug_skip;
        const int n_prec = this->n_prec + (help ? n_help : 0);
        pa.split_update = (help && MODE == PREC_STEP) ? 1 : 0;
        const unsigned nv = (MODE == PREC_INIT && pa.n_vec > 1) ? (unsigned)pa.n_vec : 1u;  // (several right-hand sides: PrecArgs::n_vec)
        if (nv == 1) pa.n_vec = 0;
        // k_prec_pre (level 0 in registers, coarse levels in LDS) when every chain fits; 4 x 4 blocks (3-D problems) only
        // with the 4-byte factor stream (score_settings.fac_fp32), otherwise the streaming kernel
        if constexpr (BS <= 3) {
            if (prec_reg && use_fac32) {  // coarse-level factors in registers, vectors only in LDS
                launch_on_stream((k_prec_pre<BS, MODE, float, true>), dim3(n_prec, nv), dim3(kPrecThreads), prec_reg_lds, slot, pa);
                return;
            }
        }
        if (prec_pre && use_fac32) {
            launch_on_stream((k_prec_pre<BS, MODE, float>), dim3(n_prec, nv), dim3(kPrecThreads), prec_pre_lds, slot, pa);
            return;
        }
        if constexpr (BS <= 3) {
            if (prec_pre) {
                launch_on_stream((k_prec_pre<BS, MODE, double>), dim3(n_prec, nv), dim3(kPrecThreads), prec_pre_lds, slot, pa);
                return;
            }
        }
        pa.split_update = 0;  // (the streaming kernel reads the plain work list: no helpers)
        if (prec_lds0) launch_on_stream((k_prec<BS, 3, MODE, true>), dim3(this->n_prec, nv), dim3(kPrecThreads), prec_lds, slot, pa);
        else launch_on_stream((k_prec<BS, 3, MODE, false>), dim3(this->n_prec, nv), dim3(kPrecThreads), prec_lds, slot, pa);
    }

    // The attribute is per kernel function, i.e. shared by every handle of the process: always
    // raise it to the same ceiling, never to what this handle happens to need (a later, smaller
    // handle would otherwise lower it under a live larger one).
    template <int BS>
    void allow_big_lds() {
        constexpr int kCeil = 158 * 1024;
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec<BS, 3, PREC_INIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec<BS, 3, PREC_STEP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec<BS, 3, PREC_INIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec<BS, 3, PREC_STEP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        if constexpr (BS <= 3) {
            HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_INIT, float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_STEP, float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_INIT, double>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
            HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_STEP, double>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        }
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_INIT, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
        HIP_CHECK(hipFuncSetAttribute((const void*)k_prec_pre<BS, PREC_STEP, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kCeil));
    }

    // SpMV launch: matrices of a replicated problem (K, G1) run with rep right-hand sides per stored row
    // XCD-aware tile order (SpmvArgs::xcd_chunk): returns the grid size.  Measured: KKT SpMV of a 16-problem batch
    // 48.0 -> 45.6 us (3.06 -> 3.23 TB/s), single problem 8.2 -> 7.8 us; kpb and rhs gain 2-3 %.
    unsigned xcd_grid(SpmvArgs& a, int nblocks) const {
        a.n_tiles = nblocks;
        if (nblocks < 16) { a.xcd_chunk = 0; return (unsigned)nblocks; }
        a.xcd_chunk = (nblocks + 7) / 8;
        return (unsigned)(8 * a.xcd_chunk);
    }
    // partial-sum ranges by value for single-problem handles (UniRanges, score_kernels.hpp); kblocks = row blocks of the
    // matrix whose p'w partials the launch reads (K in the ADMM loop, H in the Newton PCG)
    static constexpr bool uni_ranges = true;
    UniRanges uni_for(int kblocks) const {
        UniRanges u{};
        u.on = (uni_ranges && H->count == 1 && active_part_ptr.size() == 2) ? 1 : 0;
        u.l0 = 0; u.l1 = u.on ? active_part_ptr[1] : 0;
        u.k0 = 0; u.k1 = kblocks;
        return u;
    }
    // K / H product over a band view: the view's tile tables replace the source matrix's
    // (the band tiles stage their operand window in LDS: band_tile<.., LDSW = true>; the per-lane global loads of round 4 lost
    //  their A/B -- profiles/r05_band_lds_ab.txt -- and are no longer instantiated)
    template <int MODE, int NR>
    void launch_band_s(const BandBufs& Bv, const SpmvArgs& a, unsigned grid, int slot) {
        if (Bv.L.S == 8) launch_on_stream((k_spmv_band<MODE, NR, 4, true>), dim3(grid), dim3(kThreads), 0, slot, a);
        else if (Bv.L.S == 10) launch_on_stream((k_spmv_band<MODE, NR, 5, true>), dim3(grid), dim3(kThreads), 0, slot, a);
        else launch_on_stream((k_spmv_band<MODE, NR, 6, true>), dim3(grid), dim3(kThreads), 0, slot, a);
    }
    template <int MODE>
    void launch_band(const CsrBufs& M, const BandBufs& Bv, const SpmvArgs& a_in, int slot = -1) {
        SpmvArgs a = a_in;
        a.M.blk_meta = Bv.meta.d; a.M.blk_prob = Bv.prob.d; a.M.blk_rs = Bv.rs.d; a.M.nblocks = Bv.nblocks;
        a.M.blk_long = Bv.lng.d; a.M.long_part = Bv.long_part.d; a.M.long_cnt = Bv.long_cnt.d;
        a.M.long_spin = (Bv.nblocks <= M.spin_max_tiles && long_spin_enabled()) ? 1 : 0;
        a.B = Bv.dev();
        const unsigned grid = xcd_grid(a, Bv.nblocks);
        if (M.rep == 2) launch_band_s<MODE, 2>(Bv, a, grid, slot);
        else if (M.rep == 3) launch_band_s<MODE, 3>(Bv, a, grid, slot);
        else launch_band_s<MODE, 1>(Bv, a, grid, slot);
    }
    // the products with the Newton matrix (plain rows)
    template <int MODE>
    void launch_h(const SpmvArgs& a_in, int slot = -1) {
        if (Hb.on) { launch_band<MODE>(Hm, Hb, a_in, slot); return; }
        SpmvArgs a = a_in;
        const unsigned grid = xcd_grid(a, Hm.nblocks);
        launch_on_stream(k_spmv<MODE>, dim3(grid), dim3(kThreads), 0, slot, a);
    }
    // ---- probe of the Newton PCG's launches (score_debug_get "newton_probe_arm" / "newton_probe"): the next polish
    //      binds start / stop events to its chain-kernel STEPs and H products (as score_time_iteration does for the ADMM
    //      loop) and reports the mean dispatch duration of those that did work (launches queued beyond a solve's
    //      convergence are no-ops and are left out) ----
    static constexpr int kProbeCap = 1024;
    bool np_armed = false;
    std::vector<hipEvent_t> np_ev;
    struct ProbeSlot { int kind, newton_it, step; bool real; };
    std::vector<ProbeSlot> np_slots;
    int np_newton_it = 0;
    double np_out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int probe_slot(int kind, int step) {
        if (!np_armed || (int)np_slots.size() >= kProbeCap) return -1;
        np_slots.push_back(ProbeSlot{kind, np_newton_it, step, false});
        tev = np_ev.data();
        return (int)np_slots.size() - 1;
    }
    void probe_mark(int used) {  // the PCG steps of the current Newton iteration that did work
        for (auto& sl : np_slots)
            if (sl.newton_it == np_newton_it && sl.step < used) sl.real = true;
    }
    void probe_collect() {
        if (!np_armed) return;
        double sum[2] = {0, 0};
        int cnt[2] = {0, 0};
        for (size_t i = 0; i < np_slots.size(); ++i) {
            if (!np_slots[i].real) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, np_ev[2 * i], np_ev[2 * i + 1]) != hipSuccess) continue;
            sum[np_slots[i].kind] += 1e3 * (double)ms;
            cnt[np_slots[i].kind] += 1;
        }
        const HostSystem& h = *H;
        // algorithmic bytes: H product (KPB: + p, z, w_old in, p out), chain STEP of the Newton set (every chain its own factors)
        double hbytes = 0.0;
        if (Hb.on) { for (double b : Hb.L.bytes) hbytes += b; }
        else hbytes = 12.0 * (double)hm_nnz + 4.0 * (double)(h.n_tot + 1);
        hbytes += 40.0 * (double)h.n_tot;
        const double fbytes = (newton_fac32 ? 4.0 : 8.0) * (double)h.fac_doubles_H;
        np_out[0] = cnt[0]; np_out[1] = cnt[0] ? sum[0] / cnt[0] : 0.0; np_out[2] = hbytes;
        np_out[3] = cnt[1]; np_out[4] = cnt[1] ? sum[1] / cnt[1] : 0.0; np_out[5] = fbytes;
        np_out[6] = (double)hblocks(); np_out[7] = (double)n_prec;
        for (hipEvent_t e : np_ev) (void)hipEventDestroy(e);
        np_ev.clear(); np_slots.clear();
        np_armed = false;
    }
    template <int MODE>
    void launch_spmv(const CsrBufs& M, const SpmvArgs& a_in, int slot = -1) {
        static_assert(MODE == MODE_RHS || MODE == MODE_KP || MODE == MODE_KPB, "the residual / gradient modes run on plain rows (G2, H)");
        if constexpr (MODE != MODE_RHS) {
            if (&M == &K && Kb.on) { launch_band<MODE>(K, Kb, a_in, slot); return; }
        }
        SpmvArgs a = a_in;
        const unsigned grid = xcd_grid(a, M.nblocks);
        const bool half = (M.unroll == kUnroll / 2);
        if (M.rep == 2 && half) { launch_on_stream(k_spmv<MODE, 2, kUnroll / 2>, dim3(grid), dim3(kThreads), 0, slot, a); return; }
        if (M.rep == 3 && half) { launch_on_stream(k_spmv<MODE, 3, kUnroll / 2>, dim3(grid), dim3(kThreads), 0, slot, a); return; }
        if (M.rep == 2) { launch_on_stream(k_spmv<MODE, 2>, dim3(grid), dim3(kThreads), 0, slot, a); return; }
        if (M.rep == 3) { launch_on_stream(k_spmv<MODE, 3>, dim3(grid), dim3(kThreads), 0, slot, a); return; }
        launch_on_stream(k_spmv<MODE, 1>, dim3(grid), dim3(kThreads), 0, slot, a);
    }

    SpmvArgs spmv_args(const CsrBufs& M, const double* xin) {
        SpmvArgs a{};
        a.M = M.dev(); a.xin = xin; a.done = done.d; a.rs_in = M.rs_in;
        a.n_tiles = M.nblocks;  // (a launch through xcd_grid restates it)
        a.x = xy.d; a.q = q.d; a.kx = kx.d; a.r = r.d; a.sigma = H->sigma;
        a.p = p.d; a.w = w.d; a.pw_part = pw_part.d;
        a.prec_part_ptr = prec_part_ptr.d; a.kblk_part_ptr = kblk_part_ptr.d;
        a.uni = uni_for(kblocks());
        a.apply_update = 0; a.pfin = p.d; a.wfin = w.d; a.xt_rw = xtu.d; a.kx_rw = kx.d; a.x_rw = xy.d;
        a.alpha_relax = st.alpha; a.step_in = step.d;
        a.invD = invD.d; a.dres_part = dres_part.d;
        return a;
    }

    // w = K p
    void launch_kp(const double* pdir, unsigned long long* ts = nullptr, int slot = -1) {
        SpmvArgs a = spmv_args(K, pdir);
        a.p = pdir; a.tstamp = ts;
        launch_spmv<MODE_KP>(K, a, slot);
    }
    // p_new = z + beta p_old ; w = K p_new
    void launch_kpb(const double* p_old, double* p_new, const double* rz_new, const double* rz_old,
                    unsigned long long* ts = nullptr, int slot = -1) {
        SpmvArgs a = spmv_args(K, p_old);
        a.tstamp = ts;
        a.p = p_old; a.z = z.d; a.p_out = p_new; a.rz_new = rz_new; a.rz_old = rz_old;
        launch_spmv<MODE_KPB>(K, a, slot);
    }

    void set_cg_iters(int k) {
        cg_iters = std::max(1, k);
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    }

    // sqrt(r'z_final / r'z_initial) of the last measured KKT solve, per problem
    void cg_reduction(std::vector<double>& out) {
        const HostSystem& h = *H;
        out.assign(h.count, 0.0);
        if (n_prec == 0) return;
        // (pushed into the host-mapped arena by the last residuals() call)
        const double* a = h_meas;
        const double* b2 = h_meas + n_prec;
        for (int pi = 0; pi < h.count; ++pi) {
            double s0 = 0, s1 = 0;
            for (int i = active_part_ptr[pi]; i < active_part_ptr[pi + 1]; ++i) { s0 += a[i]; s1 += b2[i]; }
            out[pi] = s0 > 0 ? std::sqrt(std::max(0.0, s1) / s0) : 0.0;
        }
    }

    // enqueue one ADMM iteration on `stream`; a measuring iteration additionally
    // leaves r'z before and after the PCG sweep in rz_meas0 / rz_meas1
    // The end-of-PCG update of an iteration (xt += a p, kx += a w, x relaxed) is applied by
    // the NEXT iteration's right-hand-side kernel on its own rows, and on the fly by the cone
    // kernel's gather; only a measuring iteration (the last of a launch graph) finalises it
    // itself, so the first iteration of a graph has nothing pending (`first`).
    // `ts` (score_time_iteration only): device slots, ts_stride per kernel of the iteration (KernelStamp)
    size_t ts_stride = 0;
    void enqueue_iteration(bool measure, bool first, unsigned long long* ts = nullptr) {
        auto slot = [&](int k) -> unsigned long long* { return ts ? ts + ts_stride * k : nullptr; };
        {
            SpmvArgs ra = spmv_args(G1, xtu.d);
            ra.tstamp = slot(0);
            ra.apply_update = first ? 0 : 1;
            ra.pfin = last_p;
            launch_spmv<MODE_RHS>(G1, ra, 0);
        }
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = chains.d; pa.levels = levels.d; pa.rec = prec_rec.d; pa.fac = fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = dinv.d; pa.done = done.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = kblk_part_ptr.d; pa.uni = uni_for(kblocks());
        pa.r = r.d; pa.r_in = r.d; pa.z = z.d; pa.p = p.d; pa.w = w.d; pa.xt = xtu.d; pa.kx = kx.d;
        pa.pw_part = pw_part.d;
        double* rz_cur = measure ? rz_meas0.d : rz_part0.d;
        double* p_cur = p.d;
        double* p_oth = p2.d;
        pa.p = p_cur; pa.rz_in = nullptr; pa.rz_out = rz_cur;
        pa.tstamp = slot(1);
        launch_prec<PREC_INIT>(pa, 1);
        launch_kp(p_cur, slot(2), 2);
        for (int j = 2; j <= cg_iters; ++j) {
            double* rz_nxt = (rz_cur == rz_part0.d) ? rz_part1.d : rz_part0.d;
            pa.p = p_cur; pa.rz_in = rz_cur; pa.rz_out = rz_nxt;
            pa.tstamp = (j == 2) ? slot(3) : nullptr;
            launch_prec<PREC_STEP>(pa, (j == 2) ? 3 : -1);
            launch_kpb(p_cur, p_oth, rz_nxt, rz_cur, (j == 2) ? slot(4) : nullptr, (j == 2) ? 4 : -1);
            std::swap(p_cur, p_oth);
            rz_cur = rz_nxt;
        }
        dbg_p = p_cur;
        ConeArgs ca = cone_args(xtu.d);
        if (measure) {
            pa.tstamp = nullptr;
            pa.p = p_cur; pa.rz_in = rz_cur; pa.rz_out = rz_meas1.d;
            launch_prec<PREC_STEP>(pa);  // also applies xt += a p, kx += a w, r -= a w
            VecArgs va{};
            va.first_row = vb_first.d; va.end_row = vb_end.d; va.blk_prob = vb_prob.d; va.done = done.d;
            va.prec_part_ptr = prec_part_ptr.d; va.kblk_part_ptr = kblk_part_ptr.d;
            va.pw_part = pw_part.d; va.p = p_cur; va.w = w.d; va.kx = kx.d; va.xt = xtu.d; va.x = xy.d;
            va.alpha_relax = st.alpha; va.rz_old = rz_cur; va.apply_alpha = 0;
            hipLaunchKernelGGL(k_xupdate, dim3(n_vblocks), dim3(kThreads), 0, stream, va);
        } else {
            ca.apply_alpha = 1; ca.pfin = p_cur; ca.rz_in = rz_cur;
            last_rz = rz_cur;  // what the next iteration's right-hand-side kernel has to apply
            last_p = p_cur;
        }
        ca.tstamp = slot(5);
        if (n_cone_blocks) {
            unsigned cgrid = (unsigned)n_cone_blocks;
            if (n_cone_blocks >= 16) { ca.xcd_chunk = (n_cone_blocks + 7) / 8; ca.n_blocks = n_cone_blocks; cgrid = 8u * (unsigned)ca.xcd_chunk; }
            launch_on_stream(k_cone, dim3(cgrid), dim3(kThreads), 0, 5, ca);
            ca.xcd_chunk = 0;
        }
        if (n_large_cones) {  // (general conic programs only: a SCORE model has none)
            ca.tstamp = nullptr;
            hipLaunchKernelGGL(k_cone_wave, dim3((unsigned)((n_large_cones + 3) / 4)), dim3(kThreads), 0, stream, ca,
                               (const int2*)cone_large.d, n_large_cones);
        }
    }

    // in-loop duration of the six kernels of an iteration (see score_time_iteration):
    //   us[0..5]   device wall clock, first workgroup in -> last workgroup out
    //   us[6..11]  begin -> end of the dispatch as the runtime records it (start/stop events bound to
    //              the launch itself: the interval rocprofv3 --kernel-trace reports); 0 when with_events == 0
    // The two are taken in separate passes over the same iterations (the per-launch events make
    // the runtime wait for each dispatch's completion signal, which the plain pass does not).
    void time_iteration(int warmup, int iters, double* us, int with_events) {
        if (cg_iters != 2) throw std::runtime_error("score_time_iteration: needs cg_iters == 2");
        iters = std::max(1, iters);
        int khz = 0;
        HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, st.device));
        if (khz <= 0) throw std::runtime_error("score_time_iteration: no wall clock rate");
        for (int k = 0; k < 12; ++k) us[k] = 0.0;
        const int maxb = std::max(std::max(G1.nblocks, kblocks()) + 8, std::max(n_prec + n_help, n_cone_blocks + 8));  // (grids: XCD rounding, update helpers)
        ts_stride = (size_t)2 * maxb;
        const size_t per_iter = 6 * ts_stride, nslot = per_iter * iters;
        {
            std::vector<unsigned long long> hts(nslot);
            for (size_t i = 0; i < nslot; i += 2) { hts[i] = ~0ull; hts[i + 1] = 0ull; }
            DevBuf<unsigned long long> dts;  // (tl_arena is null outside init: a plain hipMalloc on the handle's device)
            dts.alloc(nslot);
            HIP_CHECK(hipMemcpyAsync(dts.d, hts.data(), nslot * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            HIP_CHECK(sync_stream(stream));
            for (int i = 0; i < warmup; ++i) enqueue_iteration(false, i == 0);
            for (int i = 0; i < iters; ++i) enqueue_iteration(false, warmup == 0 && i == 0, dts.d + per_iter * i);
            HIP_CHECK(hipMemcpyAsync(hts.data(), dts.d, nslot * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(sync_stream(stream));
            HIP_CHECK(hipGetLastError());
            for (int i = 0; i < iters; ++i)
                for (int k = 0; k < 6; ++k) {
                    const unsigned long long* p = &hts[per_iter * i + ts_stride * k];
                    unsigned long long t0 = ~0ull, t1 = 0ull;
                    for (int b = 0; b < maxb; ++b) { t0 = std::min(t0, p[2 * b]); t1 = std::max(t1, p[2 * b + 1]); }
                    if (t1 > t0) us[k] += (double)(t1 - t0) * 1e3 / (double)khz / iters;
                }
            if (trace_on("stamps")) {
                // per-workgroup timeline of the last timed iteration: kernel, workgroup, entry and exit in us after the
                // iteration's first entry (stdout; profiles/scripts/r04_timeline.py draws it)
                const unsigned long long* base = &hts[per_iter * (size_t)(iters - 1)];
                unsigned long long t00 = ~0ull;
                for (size_t j = 0; j < per_iter; j += 2) t00 = std::min(t00, base[j]);
                for (int k = 0; k < 6; ++k)
                    for (int b = 0; b < maxb; ++b) {
                        const unsigned long long a0 = base[ts_stride * k + 2 * (size_t)b], a1 = base[ts_stride * k + 2 * (size_t)b + 1];
                        if (a0 == ~0ull || a1 == 0ull) continue;
                        std::printf("STAMP %d %d %.3f %.3f\n", k, b, (double)(a0 - t00) * 1e3 / (double)khz, (double)(a1 - t00) * 1e3 / (double)khz);
                    }
            }
        }
        if (!with_events) return;
        std::vector<hipEvent_t> evs((size_t)12 * iters, nullptr);
        struct EvFree {
            std::vector<hipEvent_t>& v;
            ~EvFree() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
        } ev_free{evs};
        for (auto& e : evs) HIP_CHECK(hipEventCreate(&e));
        for (int i = 0; i < warmup; ++i) enqueue_iteration(false, false);
        for (int i = 0; i < iters; ++i) {
            tev = evs.data() + (size_t)12 * i;
            enqueue_iteration(false, false);
        }
        tev = nullptr;
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipGetLastError());
        for (int i = 0; i < iters; ++i)
            for (int k = 0; k < 6; ++k) {
                if (k == 5 && !n_cone_blocks) continue;
                float ms = 0.f;
                HIP_CHECK(hipEventElapsedTime(&ms, evs[(size_t)12 * i + 2 * k], evs[(size_t)12 * i + 2 * k + 1]));
                us[6 + k] += 1e3 * (double)ms / iters;
            }
    }

    void build_graph(int iters) {
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        hipGraph_t g = nullptr;
        PhaseTimer pt(st.verbose != 0);
        HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) enqueue_iteration(i == iters - 1, i == 0);
        HIP_CHECK(hipStreamEndCapture(stream, &g));
        HIP_CHECK(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        graph_iters = iters;
        pt.mark("graph capture + instantiate");
    }

    // Launch graphs pay for long blocks (the ADMM loop alone: 25 iterations = 150 launches replayed per convergence test);
    // the product default runs ONE short block (polish_warmup = 6 iterations) before the Newton polish: capturing and
    // instantiating a graph for it costs more than its 36 direct launches (0.2-0.3 ms per handle), and -- measured, round 5,
    // profiles/r05_graph_after_effect.txt -- a process that has destroyed a graph executable runs every LATER lock-step batch
    // 15 % slower (64 config-5 trials: 4130-4220 -> 3520-3580 problems/s after one default solve of any handle with a graph,
    // 4130 after the same solve without; the runtime's doing, not ours).  Blocks below kGraphMinIters are launched directly.
    static constexpr int graph_min_iters() { return 16; }
    void run(int iters) {
        if (st.use_graph && iters >= graph_min_iters()) {
            if (!graph_exec || graph_iters != iters) build_graph(iters);
            HIP_CHECK(hipGraphLaunch(graph_exec, stream));
        } else {
            for (int i = 0; i < iters; ++i) enqueue_iteration(i == iters - 1, i == 0);
            HIP_CHECK(hipGetLastError());
        }
    }

    void residuals(std::vector<ResidualSums>& R) {
        const HostSystem& h = *H;
        if (n_cone_blocks)
            hipLaunchKernelGGL(k_pres, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xy.d));
        hipLaunchKernelGGL(k_spmv<MODE_DRES>, dim3(G2.nblocks), dim3(kThreads), 0, stream, spmv_args(G2, xy.d));
        // the partials land in host-mapped memory as the kernels write them; the r'z measurements of the
        // last iteration of the launch graph are pushed there, then the sequence number
        wait_published(publish(rz_meas0.d, d_meas, (size_t)n_prec, rz_meas1.d, d_meas + n_prec, (size_t)n_prec));
        for (int pi = 0; pi < h.count; ++pi) {
            ResidualSums a;
            for (int bl = h.cone_part_ptr[pi]; bl < h.cone_part_ptr[pi + 1]; ++bl) {
                const double* o = h_pres + (size_t)bl * kPartStride;
                a.rp_u = (o[0] != o[0]) ? o[0] : std::max(a.rp_u, o[0]);
                a.ax_u = std::max(a.ax_u, o[1]); a.s_u = std::max(a.s_u, o[2]);
                a.rp_s = (o[3] != o[3]) ? o[3] : std::max(a.rp_s, o[3]);
                a.ax_s = std::max(a.ax_s, o[4]); a.s_s = std::max(a.s_s, o[5]);
                a.by += o[6];
                a.sy_yrp += o[7];
                if (a.rp_u != a.rp_u) break;
            }
            for (int bl = h.rbG2.part_ptr[pi]; bl < h.rbG2.part_ptr[pi + 1]; ++bl) {
                const double* o = h_dres + (size_t)bl * kPartStride;
                a.rd_u = (o[0] != o[0]) ? o[0] : std::max(a.rd_u, o[0]);
                a.px_u = std::max(a.px_u, o[1]); a.aty_u = std::max(a.aty_u, o[2]);
                a.rd_s = (o[3] != o[3]) ? o[3] : std::max(a.rd_s, o[3]);
                a.px_s = std::max(a.px_s, o[4]); a.aty_s = std::max(a.aty_s, o[5]);
                a.xPx += o[6]; a.qx += o[7]; a.xrd += o[8];
                if (a.rd_u != a.rd_u) break;
            }
            R[pi] = a;
        }
    }

    // score_read_estimates: poses (rounded, homogeneous), the relaxation's blocks, landmarks, range variables from the solution
    // on the device (k_read_estimates); one staging block, one wait
    DevBuf<EstProb> est_probs;
    DevBuf<int32_t> est_pose_off, est_lm_off, est_rng_off, est_rng_a, est_rng_b;
    DevBuf<double> est_rng_dist;
    bool est_up = false;
    void read_estimates(const HostSystem& h, const EstLayout& L, int qcqp_dirs, double* poses, double* relaxed, double* lms, double* rng, int32_t* degenerate) {
        const int d = L.d, D1 = d + 1, count = (int)L.probs.size();
        if (!est_up) {  // (the handle's arena: lives as long as the handle)
            tl_copy_stream = stream;
            ArenaSwap persist(&arena);
            std::vector<int32_t> po((size_t)count + 1), lo((size_t)count + 1), ro((size_t)count + 1);
            for (int p = 0; p < count; ++p) { po[(size_t)p] = L.probs[(size_t)p].pose_off; lo[(size_t)p] = L.probs[(size_t)p].lm_off; ro[(size_t)p] = L.probs[(size_t)p].rng_off; }
            po[(size_t)count] = (int32_t)L.n_pose; lo[(size_t)count] = (int32_t)L.n_lm; ro[(size_t)count] = (int32_t)L.n_rng;
            std::vector<EstProb> pr = L.probs;
            est_probs.upload(pr); est_pose_off.upload(po); est_lm_off.upload(lo); est_rng_off.upload(ro);
            est_rng_a.upload(L.rng_a); est_rng_b.upload(L.rng_b); est_rng_dist.upload(L.rng_dist);
            est_up = true;
        }
        const int rw = (L.relaxation != 0 || qcqp_dirs) ? d : 1;
        const size_t n_T = (size_t)L.n_pose * D1 * D1, n_B = (size_t)L.n_pose * d * D1, n_L = (size_t)L.n_lm * d, n_R = (size_t)L.n_rng * rw;
        const size_t n_dbl = n_T + n_B + n_L + n_R, bytes_d = n_dbl * sizeof(double), bytes = bytes_d + (size_t)L.n_pose * sizeof(int32_t);
        size_t got_d = std::max<size_t>(bytes, 64), got_h = got_d;
        char* dv = (char*)block_cache().take(got_d, st.device, false);
        char* hv = (char*)block_cache().take(got_h, st.device, true);
        EstArgs a{};
        a.d = d; a.relaxation = L.relaxation; a.count = count; a.qcqp_dirs = qcqp_dirs ? 1 : 0;
        a.probs = est_probs.d; a.pose_off = est_pose_off.d; a.lm_off = est_lm_off.d; a.rng_off = est_rng_off.d;
        a.n_pose = L.n_pose; a.n_lm = L.n_lm; a.n_rng = L.n_rng;
        a.x = xy.d; a.D = h.device_setup ? Dd.d : nullptr;
        a.rng_a = est_rng_a.d; a.rng_b = est_rng_b.d; a.rng_dist = est_rng_dist.d;
        a.poses = (double*)dv; a.relaxed = a.poses + n_T; a.lms = a.relaxed + n_B; a.rng = a.lms + n_L; a.degenerate = (int32_t*)(dv + bytes_d);
        hipError_t e = hipSuccess;
        DevBuf<double> Dtmp;
        if (!h.device_setup) {  // (host setup: the scales live on the host)
            Dtmp.alloc((size_t)h.n_tot);
            staged_h2d(Dtmp.d, h.D.data(), (size_t)h.n_tot * sizeof(double), stream);
            a.D = Dtmp.d;
        }
        const int64_t nmax = std::max(L.n_pose, std::max(L.n_lm, L.n_rng));
        if (e == hipSuccess && nmax > 0) hipLaunchKernelGGL(k_read_estimates, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, stream, a);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(hv, dv, bytes, hipMemcpyDeviceToHost, stream);
        const hipError_t e2 = sync_stream(stream);
        if (e == hipSuccess && e2 == hipSuccess) {
            if (poses) std::memcpy(poses, hv, n_T * sizeof(double));
            if (relaxed) std::memcpy(relaxed, hv + n_T * sizeof(double), n_B * sizeof(double));
            if (lms) std::memcpy(lms, hv + (n_T + n_B) * sizeof(double), n_L * sizeof(double));
            if (rng) std::memcpy(rng, hv + (n_T + n_B + n_L) * sizeof(double), n_R * sizeof(double));
            if (degenerate) std::memcpy(degenerate, hv + bytes_d, (size_t)L.n_pose * sizeof(int32_t));
        }
        block_cache().give(dv, got_d, st.device, false);
        block_cache().give(hv, got_h, st.device, true);
        HIP_CHECK(e); HIP_CHECK(e2);
    }

    // Device -> caller's arrays THROUGH PINNED MEMORY of the library's own.  A hipMemcpy into pageable memory registers the
    // caller's pages with the driver for the transfer; when the caller later frees them (NumPy arrays of the previous solve,
    // temporaries of a model construction) the address-space change evicts every queue of the process until the driver has
    // restored them -- measured, round 5: the first kernel after such a free started 14-37 ms late (20 x 5000 poses: 12 MB of
    // x, y, s per solve; solves 47 / 46 / 21 / 21 ms), whichever handle or thread it belonged to.
    void copy_out(const std::vector<std::pair<double*, std::pair<const double*, size_t>>>& parts) {  // {host dst, {device src, count}}
        size_t total = 0;
        for (const auto& pt : parts) if (pt.first) total += pt.second.second;
        if (!total) return;
        size_t got = total * sizeof(double);
        double* pin = (double*)block_cache().take(got, st.device, true);
        size_t off = 0;
        hipError_t err = hipSuccess;
        for (const auto& pt : parts)
            if (pt.first && pt.second.second) {
                const hipError_t e = hipMemcpyAsync(pin + off, pt.second.first, pt.second.second * sizeof(double), hipMemcpyDeviceToHost, stream);
                if (e != hipSuccess) err = e;
                off += pt.second.second;
            }
        const hipError_t es = sync_stream(stream);
        if (err == hipSuccess && es == hipSuccess) {
            std::vector<std::pair<double*, std::pair<const double*, size_t>>> hp;
            off = 0;
            for (const auto& pt : parts)
                if (pt.first && pt.second.second) { hp.push_back({pt.first, {pin + off, pt.second.second}}); off += pt.second.second; }
            for (const auto& c : hp) {
                const size_t cnt = c.second.second;
                parallel_ranges((int64_t)cnt, (int64_t)1 << 17, [&](int, int64_t a, int64_t b) {
                    std::memcpy(c.first + a, c.second.first + a, (size_t)(b - a) * sizeof(double));
                });
            }
        }
        block_cache().give(pin, got, st.device, true);
        HIP_CHECK(err); HIP_CHECK(es);
    }
    void download(const HostSystem& h, double* x, double* y, double* s_out) {
        const int64_t n = h.n_tot, m = h.m_tot;
        if (h.device_setup) {  // the scales live on the device: x = xhat D, y = yhat E, s = shat / E there, then one copy each
            size_t got = (size_t)(n + 2 * m) * sizeof(double);
            double* ob = (double*)block_cache().take(got, st.device, false);
            hipLaunchKernelGGL(k_unscale, dim3((unsigned)std::max<int64_t>(1, (std::max(n, m) + 255) / 256)), dim3(256), 0, stream, (const double*)xy.d, (const double*)s.d,
                               (const double*)Dd.d, (const double*)Ed.d, ob, n, m);
            try {
                copy_out({{x, {ob, (size_t)n}}, {y, {ob + n, (size_t)m}}, {s_out, {ob + n + m, (size_t)m}}});
            } catch (...) {
                block_cache().give(ob, got, st.device, false);
                throw;
            }
            block_cache().give(ob, got, st.device, false);
            return;
        }
        std::vector<double> hx((size_t)(n + m)), hs((size_t)m);
        copy_out({{hx.data(), {xy.d, hx.size()}}, {m ? hs.data() : nullptr, {s.d, hs.size()}}});
        if (x) for (int64_t i = 0; i < n; ++i) x[i] = hx[i] * h.D[i];
        if (y) for (int64_t i = 0; i < m; ++i) y[i] = hx[n + i] * h.E[i];
        if (s_out) for (int64_t i = 0; i < m; ++i) s_out[i] = hs[i] / h.E[i];
    }

    int64_t get_vec(const char* name, double* out, int64_t len) {
        const HostSystem& h = *H;
        const double* src = nullptr;
        int64_t sz = 0;
        bool host = false;
        std::string nm(name);
        if (nm == "xt") { src = xtu.d; sz = h.n_tot; }
        else if (nm == "u") { src = xtu.d + h.n_tot; sz = h.m_tot; }
        else if (nm == "x") { src = xy.d; sz = h.n_tot; }
        else if (nm == "y") { src = xy.d + h.n_tot; sz = h.m_tot; }
        else if (nm == "s") { src = s.d; sz = h.m_tot; }
        else if (nm == "r") { src = r.d; sz = h.n_tot; }
        else if (nm == "z") { src = z.d; sz = h.n_tot; }
        else if (nm == "p") { src = dbg_p ? dbg_p : ((cg_iters % 2 == 1) ? p.d : p2.d); sz = h.n_tot; }  // last PCG direction
        else if (nm == "w") { src = w.d; sz = h.n_tot; }
        else if (nm == "kx") { src = kx.d; sz = h.n_tot; }
        else if (nm == "D" && h.device_setup) { src = Dd.d; sz = h.n_tot; }
        else if (nm == "E" && h.device_setup) { src = Ed.d; sz = h.m_tot; }
        else if (nm == "D") { src = h.D.data(); sz = h.n_tot; host = true; }
        else if (nm == "E") { src = h.E.data(); sz = h.m_tot; host = true; }
        else if (nm == "K0" || nm == "K1" || nm == "Aval" || nm == "G1val" || nm == "G2val" || nm == "qs" || nm == "bs" || nm == "invD" || nm == "invE") {
            // the setup's value arrays as the kernels read them (tests: device setup against host setup, bit for bit)
            const int64_t nk = (int64_t)h.K.col.size(), na = (int64_t)h.A.ptr[(size_t)h.m_tot];
            const int64_t n1 = h.device_setup ? g1_nnz : (int64_t)h.G1.col.size(), n2 = (int64_t)h.G2.ptr[(size_t)h.n_tot];
            if (nm == "K0") { src = K0d.d; sz = nk; } else if (nm == "K1") { src = K1d.d; sz = nk; }
            else if (nm == "Aval") { src = A_val.d; sz = na; } else if (nm == "G1val") { src = G1.val.d; sz = n1; }
            else if (nm == "G2val") { src = G2.val.d; sz = n2; } else if (nm == "qs") { src = q.d; sz = h.n_tot; }
            else if (nm == "bs") { src = b.d; sz = h.m_tot; } else if (nm == "invD") { src = invD.d; sz = h.n_tot; }
            else { src = invE.d; sz = h.m_tot; }
        }
        else if (nm == "Acol" || nm == "Aptr" || nm == "G1col" || nm == "G1ptr" || nm == "G2col" || nm == "G2ptr" || nm == "G2split" || nm == "Kcol_dev" || nm == "Kptr_dev") {
            const int64_t nk = (int64_t)h.K.col.size(), na = (int64_t)h.A.ptr[(size_t)h.m_tot];
            const int64_t n1 = h.device_setup ? g1_nnz : (int64_t)h.G1.col.size(), n2 = (int64_t)h.G2.ptr[(size_t)h.n_tot];
            const int32_t* isrc = nullptr;
            if (nm == "Acol") { isrc = A_col.d; sz = na; } else if (nm == "Aptr") { isrc = A_ptr.d; sz = h.m_tot + 1; }
            else if (nm == "G1col") { isrc = G1.col.d; sz = n1; } else if (nm == "G1ptr") { isrc = G1.ptr.d; sz = h.n_tot + 1; }
            else if (nm == "G2col") { isrc = G2.col.d; sz = n2; } else if (nm == "G2ptr") { isrc = G2.ptr.d; sz = h.n_tot + 1; }
            else if (nm == "G2split") { isrc = G2.split.d; sz = h.n_tot; } else if (nm == "Kcol_dev") { isrc = K.col.d; sz = nk; }
            else { isrc = K.ptr.d; sz = h.n_tot + 1; }
            if (out && len > 0) {
                std::vector<int32_t> tmp((size_t)std::min(len, sz));
                HIP_CHECK(sync_stream(stream));
                staged_d2h(tmp.data(), isrc, tmp.size() * sizeof(int32_t), stream);
                HIP_CHECK(sync_stream(stream));
                for (size_t i = 0; i < tmp.size(); ++i) out[i] = (double)tmp[i];
            }
            return sz;
        }
        else if (nm == "setup_scalars") {  // per problem: |q|_inf, |b|_inf unscaled and scaled, kkt_bytes
            sz = 5 * (int64_t)h.count;
            if (out && len >= sz)
                for (int p = 0; p < h.count; ++p) {
                    out[5 * p] = h.qnorm_u[(size_t)p]; out[5 * p + 1] = h.qnorm_s[(size_t)p]; out[5 * p + 2] = h.bnorm_u[(size_t)p];
                    out[5 * p + 3] = h.bnorm_s[(size_t)p]; out[5 * p + 4] = h.kkt_bytes[(size_t)p];
                }
            return sz;
        }
        else if (nm == "device_setup") {
            if (out && len > 0) out[0] = h.device_setup ? 1.0 : 0.0;
            return 1;
        }
        else if (nm == "Kval") { src = K.val.d; sz = (int64_t)h.K.col.size(); }
        else if (nm == "rep") {  // [replicas the kernels run with (1 = general problem), nnz of the stored K, of the stored A']
            const double v[3] = {(double)h.rep, (double)h.K.col.size(), (double)(h.device_setup ? (size_t)g1_nnz : h.G1.col.size())};
            if (out && len > 0) std::memcpy(out, v, sizeof(double) * (size_t)std::min<int64_t>(len, 3));
            return 3;
        }
        else if (nm == "links") {  // [node pairs outside the chains, pairs inside the preconditioner, unknowns, affected chains, rounds, problems whose capacitance matrix was singular]
            double v[6] = {(double)link_plan.pairs_total, (double)link_plan.pairs_used, (double)n_link_u, (double)n_link_items, (double)link_rounds, 0.0};
            if (n_link_probs && out) {
                std::vector<int32_t> stt((size_t)n_link_probs);
                HIP_CHECK(sync_stream(stream));
                for (int set = 0; set < 2; ++set) {
                    if (!link_set_on[set]) continue;
                    HIP_CHECK(hipMemcpy(stt.data(), link_status[set].d, sizeof(int32_t) * (size_t)n_link_probs, hipMemcpyDeviceToHost));
                    for (int32_t x : stt) v[5] += x;
                }
            }
            if (out && len > 0) std::memcpy(out, v, sizeof(double) * (size_t)std::min<int64_t>(len, 6));
            return 6;
        }
        else if (nm == "link_pairs") {  // first columns of the node pairs inside the Newton preconditioner (two per pair)
            const std::vector<int32_t>& pc = link_plan.pair_cols;
            const int64_t np_ = n_link_items ? (int64_t)pc.size() : 0;
            if (out) for (int64_t i = 0; i < np_ && i < len; ++i) out[i] = (double)pc[(size_t)i];
            return np_;
        }
        else if (nm == "fac") { src = fac.d; sz = (int64_t)h.fac_doubles; }
        else if (nm == "newton_probe_arm") {  // the next polish times its PCG launches (see probe_slot)
            if (!Q.available) return -1;
            if (out && len > 0) {
                for (hipEvent_t e : np_ev) (void)hipEventDestroy(e);
                np_ev.assign((size_t)2 * kProbeCap, nullptr);
                for (auto& e : np_ev)
                    if (hipEventCreate(&e) != hipSuccess) return -2;
                np_slots.clear();
                np_armed = true;
                out[0] = 1.0;
            }
            return 1;
        }
        else if ((nm == "ag_device_check" || nm == "polish_build_check") && h.device_setup) return -1;  // (no host matrices to compare with: SCORE_HOST_SETUP=1)
        else if (nm == "ag_device_check") {
            // the equilibrated A, G1, G2 on the device against the host arrays: [derived on the device (0/1), mismatching
            // columns of A, max |A value difference|, the same for G1, for G2]
            if (out && len >= 7) {
                auto down_i = [&](const int32_t* d, size_t cnt) {
                    std::vector<int32_t> v(cnt);
                    if (cnt) staged_d2h(v.data(), d, cnt * sizeof(int32_t), stream);
                    HIP_CHECK(sync_stream(stream));
                    return v;
                };
                auto down_d = [&](const double* d, size_t cnt) {
                    std::vector<double> v(cnt);
                    if (cnt) staged_d2h(v.data(), d, cnt * sizeof(double), stream);
                    HIP_CHECK(sync_stream(stream));
                    return v;
                };
                auto mism = [](const std::vector<int32_t>& x, const std::vector<int32_t>& y) {
                    double c = 0;
                    for (size_t i = 0; i < y.size(); ++i) c += x[i] != y[i];
                    return c;
                };
                auto maxd = [](const std::vector<double>& x, const std::vector<double>& y) {
                    double c = 0;
                    for (size_t i = 0; i < y.size(); ++i) c = std::max(c, std::fabs(x[i] - y[i]));
                    return c;
                };
                out[0] = derive_ag ? 1.0 : 0.0;
                out[1] = mism(down_i(A_col.d, h.A.col.size()), h.A.col); out[2] = maxd(down_d(A_val.d, h.A.val.size()), h.A.val);
                out[3] = mism(down_i(G1.col.d, h.G1.col.size()), h.G1.col); out[4] = maxd(down_d(G1.val.d, h.G1.val.size()), h.G1.val);
                out[5] = mism(down_i(G2.col.d, h.G2.col.size()), h.G2.col); out[6] = maxd(down_d(G2.val.d, h.G2.val.size()), h.G2.val);
            }
            return 7;
        }
        else if (nm == "polish_build_check") {
            // the Newton matrix built on the device against the host loop's (score_polish_host.hpp): [built on the device (0/1),
            // entries device, entries host, mismatching row pointers, columns, max |P-on-pattern difference|, mismatching
            // list pointers, cones, block indices, max |coefficient difference|, mismatching chain positions (diagonal,
            // sub-diagonal), Jacobi positions, long entries]
            if (!Q.available) return -1;
            if (out && len >= 14) {
                PolishData R;
                build_polish(h, R, false, false);
                auto down_i = [&](const int32_t* d, size_t cnt) {
                    std::vector<int32_t> v(cnt);
                    if (cnt) staged_d2h(v.data(), d, cnt * sizeof(int32_t), stream);
                    HIP_CHECK(sync_stream(stream));
                    return v;
                };
                auto down_d = [&](const double* d, size_t cnt) {
                    std::vector<double> v(cnt);
                    if (cnt) staged_d2h(v.data(), d, cnt * sizeof(double), stream);
                    HIP_CHECK(sync_stream(stream));
                    return v;
                };
                auto mism = [](const std::vector<int32_t>& x, const std::vector<int32_t>& y) {
                    double c = (double)(x.size() > y.size() ? x.size() - y.size() : y.size() - x.size());
                    for (size_t i = 0; i < std::min(x.size(), y.size()); ++i) c += x[i] != y[i];
                    return c;
                };
                auto maxd = [](const std::vector<double>& x, const std::vector<double>& y) {
                    double c = x.size() == y.size() ? 0.0 : 1e300;
                    for (size_t i = 0; i < std::min(x.size(), y.size()); ++i) c = std::max(c, std::fabs(x[i] - y[i]));
                    return c;
                };
                const size_t nz = (size_t)hm_nnz, nc = R.ccone.size();
                out[0] = polish_on_device ? 1.0 : 0.0;
                out[1] = (double)hm_nnz; out[2] = (double)R.Hm.col.size();
                out[3] = mism(down_i(Hm.ptr.d, (size_t)h.n_tot + 1), R.Hm.ptr);
                out[4] = mism(down_i(Hm.col.d, nz), R.Hm.col);
                out[5] = maxd(down_d(q_Pon.d, nz), R.Pon);
                out[6] = mism(down_i(q_cptr.d, nz + 1), R.cptr);
                out[7] = mism(down_i(q_ccone.d, nc), R.ccone);
                out[8] = mism(down_i(q_cab.d, nc), R.cab);
                out[9] = maxd(down_d(q_ccoef.d, nc), R.ccoef);
                out[10] = mism(down_i(q_posd.d, R.pos_diag.size()), R.pos_diag);
                out[11] = mism(down_i(q_poss.d, R.pos_sub.size()), R.pos_sub);
                out[12] = mism(down_i(q_diagpos.d, R.diag_pos.size()), R.diag_pos);
                out[13] = mism(Q.long_ent, R.long_ent) + mism(Q.long_prob, R.long_prob);
            }
            return 14;
        }
        else if (nm == "newton_probe") {
            // [H products timed, mean us, bytes per launch, chain STEPs timed, mean us, factor bytes per launch, H tiles, prec work items]
            if (out && len > 0) std::memcpy(out, np_out, sizeof(double) * (size_t)std::min<int64_t>(len, 8));
            return 8;
        }
        // ---- kernel-level checks of the Newton polish (tests/test_gpu_parity.py) ----
        else if (nm == "polish_assemble_at_x") {
            // evaluate F, gradient, generalised Hessian and its chain factors at the current ADMM
            // iterate x (what the first Newton iteration does); returns F
            if (!Q.available) return -1;
            if (out && len > 0) {
                try {
                    c_step.assign(h.count, 1.0); c_tol2.assign(h.count, 0.0); c_skip.assign(h.count, 0);
                    std::vector<char> all(h.count, 1);
                    NewtonVecArgs va{};
                    va.n = h.n_tot; va.is_head = q_ishead.d; va.g = q_g.d; va.part = q_gd.d;
                    HIP_CHECK(hipMemsetAsync(q_g.d, 0, q_g.n * sizeof(double), stream));
                    va.u = xy.d; va.delta = xy.d; va.step = 0.0; va.out = q_X0.d;
                    hipLaunchKernelGGL(k_newton_trial, dim3((unsigned)((h.n_tot + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, va);
                    upload_skip(all);
                    std::vector<double> F(h.count), gn(h.count);
                    newton_eval_batch(q_X0.d, all, F, gn);
                    newton_hessian(q_skip.d);
                    HIP_CHECK(sync_stream(stream));
                    out[0] = F[0];
                } catch (const std::exception&) { return -2; }
            }
            return 1;
        }
        else if (nm == "polish_prec_of_negg") {
            // z = M^-1 (-g) with the Newton preconditioner as factored on the device (k_factor), applied by
            // the chain kernel the PCG uses
            if (!Q.available) return -1;
            if (out && len > 0) {
                PrecArgs pa{};
                pa.work = prec_work.d; pa.chains = chainsH.d; pa.levels = levelsH.d; pa.rec = prec_recH.d; pa.fac = q_fac.d;
                pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = q_dinv.d; pa.done = q_skip.d;
                pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = q_hblk_part.d; pa.uni = uni_for(hblocks());
                pa.r = r.d; pa.r_in = q_negg.d; pa.z = z.d; pa.p = p.d; pa.w = w.d; pa.xt = q_delta.d; pa.kx = q_dummy.d;
                pa.pw_part = q_pw.d; pa.rz_in = nullptr; pa.rz_out = rz_part0.d;
                launch_prec<PREC_INIT>(pa);
            }
            src = z.d; sz = h.n_tot;
        }
        else if (nm == "polish_g") { src = q_g.d; sz = Q.available ? h.n_tot : 0; }
        else if (nm == "Hval") { src = Hm.val.d; sz = Q.available ? hm_nnz : 0; }
        else if (nm == "Hcol" || nm == "Hptr" || nm == "is_head" || nm == "chain_of_col") {
            if (!Q.available) return -1;
            std::vector<double> tmp;
            if (nm == "is_head" && h.device_setup) {  // (made on the device: the host never held it)
                std::vector<int32_t> ih((size_t)h.n_tot);
                staged_d2h(ih.data(), q_ishead.d, sizeof(int32_t) * ih.size(), stream);
                HIP_CHECK(sync_stream(stream));
                tmp.assign(ih.begin(), ih.end());
            } else
            if (nm == "Hcol" && polish_on_device) {  // (built on the device: the host never held it)
                std::vector<int32_t> hc((size_t)hm_nnz);
                staged_d2h(hc.data(), Hm.col.d, sizeof(int32_t) * hc.size(), stream);
                HIP_CHECK(sync_stream(stream));
                tmp.assign(hc.begin(), hc.end());
            }
            else if (nm == "Hcol") tmp.assign(Q.Hm.col.begin(), Q.Hm.col.end());
            else if (nm == "Hptr") tmp.assign(Q.Hm.ptr.begin(), Q.Hm.ptr.end());
            else if (nm == "is_head") tmp.assign(Q.is_head.begin(), Q.is_head.end());
            else {  // per column: chain node index (global numbering over all chains) or -1
                tmp.assign(h.n_tot, -1.0);
                for (size_t ci = 0; ci < h.chains.size(); ++ci)
                    for (int i = 0; i < h.chains[ci].N; ++i)
                        for (int c = 0; c < h.bs; ++c) tmp[h.node_col[h.chains[ci].node_begin + i] + c] = (double)(h.chains[ci].node_begin + i) + 1e-3 * 0;
                // chain boundaries: encode the chain id in a second pass through "chain_id_of_col"
            }
            sz = (int64_t)tmp.size();
            if (out && len > 0) std::memcpy(out, tmp.data(), sizeof(double) * (size_t)std::min(len, sz));
            return sz;
        }
        else if (nm == "Kcol" || nm == "Kptr") {  // pattern of the K the kernels stream (a replicated problem: replica 0 + tail rows)
            std::vector<double> tmp;
            if (nm == "Kcol") tmp.assign(h.K.col.begin(), h.K.col.end());
            else tmp.assign(h.K.ptr.begin(), h.K.ptr.end());
            sz = (int64_t)tmp.size();
            if (out && len > 0) std::memcpy(out, tmp.data(), sizeof(double) * (size_t)std::min(len, sz));
            return sz;
        }
        else if (nm == "chain_id_of_col") {
            std::vector<double> tmp(h.n_tot, -1.0);
            for (size_t ci = 0; ci < h.chains.size(); ++ci)
                for (int i = 0; i < h.chains[ci].N; ++i)
                    for (int c = 0; c < h.bs; ++c) tmp[h.node_col[h.chains[ci].node_begin + i] + c] = (double)ci;
            sz = (int64_t)tmp.size();
            if (out && len > 0) std::memcpy(out, tmp.data(), sizeof(double) * (size_t)std::min(len, sz));
            return sz;
        }
        else return -1;
        if (out && len > 0) {
            const size_t bytes = sizeof(double) * (size_t)std::min(len, sz);
            if (host) std::memcpy(out, src, bytes);
            else {
                if (sync_stream(stream) != hipSuccess) return -2;
                try { staged_d2h(out, src, bytes, stream); } catch (const std::exception&) { return -2; }
            }
        }
        return sz;
    }

    // ---- linear mode: K x = rhs by the chain-preconditioned PCG of the ADMM loop (k_prec_pre / k_prec +
    //      k_spmv), K factored on the device (k_factor), termination on the device (pcg_gate):
    //      r'M^-1 r <= rel_tol^2 r0'M^-1 r0.  The host looks at one flag per chunk of iterations. ----
    DevBuf<double> lin_rhs, lin_tol2, lin_ref;
    DevBuf<int32_t> lin_flag;  // [gate flag | STEPs executed] per problem
    static constexpr int kDirectEvery = 32;  // PCG: a direct product w = K p every so many recursive ones
    void linear_buffers(const HostSystem& h) {
        if (h.m_tot != 0 || h.count != 1) throw std::runtime_error("linear mode: one unconstrained pattern per handle");
        if (!lin_flag.d) {
            lin_rhs.alloc((size_t)h.n_tot); lin_tol2.alloc(1); lin_ref.alloc(1); lin_flag.alloc(2);
        }
    }
    bool linear_solve(const HostSystem& h, const double* rhs, double* x, double rel_tol, int max_iters, int* used_out) {
        linear_buffers(h);
        const size_t n = (size_t)h.n_tot;
        // values -> K0 (K1 = 0)
        staged_h2d(K0d.d, h.K0.data(), h.K0.size() * sizeof(double), stream);
        staged_h2d(lin_rhs.d, rhs, n * sizeof(double), stream);
        const bool ok = linear_solve_core(h, lin_rhs.d, rel_tol, max_iters, used_out);
        staged_d2h(x, xtu.d, n * sizeof(double), stream);
        return ok;
    }
    // K0d holds the values and rhs_dev the right-hand side, both on the device; the solution is left in xtu
    bool linear_solve_core(const HostSystem& h, const double* rhs_dev, double rel_tol, int max_iters, int* used_out) {
        linear_buffers(h);
        derive_rho_data(false);  // K = K0 and its chain factors / Jacobi inverses on the device
        const double tol2 = rel_tol * rel_tol;
        HIP_CHECK(hipMemcpyAsync(lin_tol2.d, &tol2, sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipMemsetAsync(lin_flag.d, 0, 2 * sizeof(int32_t), stream));
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = chains.d; pa.levels = levels.d; pa.rec = prec_rec.d; pa.fac = fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = dinv.d; pa.done = lin_flag.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = kblk_part_ptr.d; pa.uni = uni_for(kblocks());
        pa.r = r.d; pa.r_in = rhs_dev; pa.z = z.d; pa.w = w.d; pa.xt = xtu.d; pa.kx = kx.d; pa.pw_part = pw_part.d;
        pa.gate_used = lin_flag.d + 1;
        pa.early_done = 1;
        double* rz_cur = rz_part0.d;
        double* p_cur = p.d;
        double* p_oth = p2.d;
        pa.p = p_cur; pa.rz_in = nullptr; pa.rz_out = rz_cur;
        launch_prec<PREC_INIT>(pa);  // z = M^-1 rhs, p = z
        {
            SpmvArgs a = spmv_args(K, p_cur);
            a.p = p_cur; a.done = lin_flag.d;
            launch_spmv<MODE_KP>(K, a);
        }
        pa.gate_flag = lin_flag.d; pa.gate_tol2 = lin_tol2.d; pa.gate_ref = lin_ref.d;
        int32_t state[2] = {0, 0};
        int queued = 0;
        bool first = true;
        while (!state[0] && queued < max_iters) {
            const int chunk = std::min(max_iters - queued, queued == 0 ? 16 : 32);
            for (int j = 0; j < chunk; ++j) {
                double* rz_nxt = (rz_cur == rz_part0.d) ? rz_part1.d : rz_part0.d;
                pa.p = p_cur; pa.rz_in = rz_cur; pa.rz_out = rz_nxt;
                pa.gate_first = first ? 1 : 0;
                pa.r_in = first ? rhs_dev : r.d;
                pa.xt_zero = first ? 1 : 0;
                launch_prec<PREC_STEP>(pa);  // x += a p ; r -= a w ; z = M^-1 r   (or the gate fires)
                SpmvArgs a = spmv_args(K, p_cur);
                a.p = p_cur; a.z = z.d; a.p_out = p_oth; a.rz_new = rz_nxt; a.rz_old = rz_cur; a.done = lin_flag.d;
                a.early_done = 1;
                launch_spmv<MODE_KPB>(K, a);
                std::swap(p_cur, p_oth);
                if ((queued + j + 1) % kDirectEvery == 0) {
                    // the recurrence w = K z + beta w_old accumulates rounding over a long solve: every kDirectEvery-th
                    // product is recomputed directly, w = K p (and its p'w)
                    SpmvArgs d = spmv_args(K, p_cur);
                    d.p = p_cur; d.done = lin_flag.d; d.early_done = 1;
                    launch_spmv<MODE_KP>(K, d);
                }
                rz_cur = rz_nxt;
                first = false;
            }
            queued += chunk;
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(state, lin_flag.d, sizeof(state), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(sync_stream(stream));
        }
        if (used_out) *used_out = state[1];
        return state[0] != 0;
    }

    // The Newton matrix's pattern, P on it and the contribution lists, on the device (score_polish_device.hpp); the host keeps
    // the row pointers (tiles, per-problem entry ranges).  false: the program's cones are not laid out the way the kernels
    // assume -- the caller builds on the host.
    bool build_polish_on_device(const HostSystem& h) {
        const int T = Q.T, D1 = T + 1, bs = h.bs, b2 = bs * bs;
        const int64_t n = h.n_tot;
        PhaseTimer pt(st.verbose != 0);
        for (size_t k = 0; k < h.cone_row.size(); ++k)
            if ((int64_t)h.cone_row[k] != (int64_t)k * D1) return false;
        int64_t con_max = 0;
        const int64_t rec_max = polish_record_bound(h, T, &con_max, h.device_setup ? nnzP_full : -1);
        if (rec_max + 64 >= ((int64_t)1 << 31)) return false;
        const int long_max = 1 << 16;
        // what stays: pattern, P on it, lists, positions (the handle's arena)
        Hm.ptr.alloc((size_t)n + 1);
        {
            ZeroGroup zg;
            zg.add(Hm.col, (size_t)rec_max + 64); zg.add(Hm.val, (size_t)rec_max + 64);
            zg.commit(stream);
        }
        q_Pon.alloc((size_t)rec_max); q_cptr.alloc((size_t)rec_max + 1);
        q_ccone.alloc((size_t)std::max<int64_t>(1, con_max)); q_cab.alloc((size_t)std::max<int64_t>(1, con_max)); q_ccoef.alloc((size_t)std::max<int64_t>(1, con_max));
        q_posd.alloc(h.node_col.size() * (size_t)b2); q_poss.alloc(h.node_col.size() * (size_t)b2); q_diagpos.alloc(h.diag_cols.size());
        std::vector<long long> res(3, 0);
        std::vector<int32_t> longs((size_t)long_max);
        Q.Hm.nrows = Q.Hm.ncols = n;
        Q.Hm.ptr.assign((size_t)n + 1, 0);
        {
            // scratch of the build: its own arena, back to the block cache when the build is over
            DevArena tmp;
            tmp.dev = st.device;
            struct Swap {
                DevArena* keep;
                explicit Swap(DevArena* a) : keep(tl_arena) { tl_arena = a; }
                ~Swap() { tl_arena = keep; }
            } swap(&tmp);
            try {
            DevBuf<long long> cnt, off, result;
            DevBuf<unsigned long long> key0, key1, flag, flag_s;
            DevBuf<uint32_t> idx0, idx1;
            DevBuf<int32_t> rcone, rab, hrow, long_ent, prev_col;
            DevBuf<double> rcoef;
            cnt.alloc((size_t)n + 1); off.alloc((size_t)n + 1); result.alloc(3);
            key0.alloc((size_t)rec_max); key1.alloc((size_t)rec_max); idx0.alloc((size_t)rec_max); idx1.alloc((size_t)rec_max);
            flag.alloc((size_t)rec_max); flag_s.alloc((size_t)rec_max);
            rcone.alloc((size_t)rec_max); rab.alloc((size_t)rec_max); rcoef.alloc((size_t)rec_max); hrow.alloc((size_t)rec_max);
            long_ent.alloc((size_t)long_max);
            std::optional<UploadBatch> fills;  // (counters, the sort's segment list, the predecessor table: one fill launch, one transfer)
            fills.emplace();
            fill_zero_async(result.d, 3 * sizeof(long long), stream);
            {
                std::vector<int32_t> pc(h.node_col.size(), -1);  // column of the chain predecessor
                for (const auto& ch : h.chains)
                    for (int i = 1; i < ch.N; ++i) pc[(size_t)ch.node_begin + i] = h.node_col[(size_t)ch.node_begin + i - 1];
                prev_col.upload(pc);
            }
            HBuildArgs a{};
            a.n = n; a.T = T;
            a.g2_ptr = G2.ptr.d; a.g2_split = G2.split.d; a.g2_col = G2.col.d; a.g2_val = G2.val.d;
            a.A_ptr = A_ptr.d; a.A_col = A_col.d; a.A_val = A_val.d; a.is_head = q_ishead.d;
            a.rec_max = rec_max; a.key = key0.d; a.idx = idx0.d; a.rcone = rcone.d; a.rab = rab.d; a.rcoef = rcoef.d;
            const unsigned grec = (unsigned)((rec_max + 255) / 256);
            // (eight lanes per row; the listed long rows -- landmark rows -- a wavefront each)
            const int64_t long_cap = rec_max / kLongRowEntries + 1;
            DevBuf<int32_t> long_rows, n_long_rows;
            long_rows.alloc((size_t)long_cap); n_long_rows.alloc(1);
            fill_zero_async(n_long_rows.d, sizeof(int32_t), stream);
            int bits = 1;
            while (((int64_t)1 << bits) <= n) ++bits;  // (the sentinel row n sorts last)
            RowSort rs;  // (every row's records sorted where they lie: sort_rows)
            sort_rows_plan(rs, n, rec_max, bits, flag.d, flag_s.d);
            fills.reset();
            hipLaunchKernelGGL(k_row_classify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const int32_t*)G2.ptr.d, n, long_rows.d, n_long_rows.d);
            const unsigned g8 = (unsigned)((n + 31) / 32), g64 = (unsigned)((long_cap + 3) / 4);
            a.long_rows = long_rows.d; a.n_long_rows = n_long_rows.d;
            a.rec_cnt = cnt.d;
            hipLaunchKernelGGL(k_hb_count<8>, dim3(g8), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(k_hb_count<64>, dim3(g64), dim3(256), 0, stream, a);
            size_t tb = 0, tb2 = rs.bytes, tb3 = 0;
            HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, cnt.d, off.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
            HIP_CHECK(rocprim::inclusive_scan(nullptr, tb3, flag.d, flag_s.d, (size_t)rec_max, rocprim::plus<unsigned long long>(), stream));
            DevBuf<unsigned char> scratch;
            scratch.alloc(std::max(tb, std::max(tb2, tb3)) + 256);
            HIP_CHECK(rocprim::exclusive_scan((void*)scratch.d, tb, cnt.d, off.d, (long long)0, (size_t)n + 1, rocprim::plus<long long>(), stream));
            a.rec_cnt = off.d;
            hipLaunchKernelGGL(k_hb_expand<8>, dim3(g8), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(k_hb_expand<64>, dim3(g64), dim3(256), 0, stream, a);
            hipLaunchKernelGGL(k_hb_pad, dim3(grec), dim3(256), 0, stream, a);
            sort_rows(rs, n, rec_max, bits, key0.d, key1.d, idx1.d, flag.d, flag_s.d, off.d, (void*)scratch.d);
            HScatterArgs sa{};
            sa.n = n; sa.rec_max = rec_max; sa.key = key1.d; sa.idx = idx1.d; sa.rcone = rcone.d; sa.rab = rab.d; sa.rcoef = rcoef.d;
            sa.flag = flag.d; sa.Hcol = Hm.col.d; sa.Hrow = hrow.d; sa.Pon = q_Pon.d; sa.cptr = q_cptr.d;
            sa.ccone = q_ccone.d; sa.cab = q_cab.d; sa.ccoef = q_ccoef.d; sa.Hptr = Hm.ptr.d; sa.result = result.d;
            sa.long_ent = long_ent.d; sa.long_max = long_max; sa.diag_reg = kPolishDiagReg;
            hipLaunchKernelGGL(k_hb_flags, dim3(grec), dim3(256), 0, stream, sa);
            HIP_CHECK(rocprim::inclusive_scan((void*)scratch.d, tb3, flag.d, flag_s.d, (size_t)rec_max, rocprim::plus<unsigned long long>(), stream));
            sa.flag = flag_s.d;
            hipLaunchKernelGGL(k_hb_scatter, dim3(grec), dim3(256), 0, stream, sa);
            hipLaunchKernelGGL(k_hb_rows, dim3(grec), dim3(256), 0, stream, sa);
            HPosArgs pa{};
            pa.Hptr = Hm.ptr.d; pa.Hcol = Hm.col.d; pa.node_col = node_col.d; pa.prev_col = prev_col.d;
            pa.n_nodes = (int64_t)h.node_col.size(); pa.bs = bs; pa.pos_diag = q_posd.d; pa.pos_sub = q_poss.d;
            pa.diag_cols = diag_cols.d; pa.n_diag = (int64_t)h.diag_cols.size(); pa.diag_pos = q_diagpos.d;
            const int64_t npos = std::max<int64_t>(pa.n_nodes * b2, pa.n_diag);
            if (npos > 0) hipLaunchKernelGGL(k_hb_positions, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, stream, pa);
            HIP_CHECK(hipGetLastError());
            pt.mark("    polish (device): buffers + launches");
            HIP_CHECK(hipMemcpyAsync(res.data(), result.d, 3 * sizeof(long long), hipMemcpyDeviceToHost, stream));
            staged_d2h(Q.Hm.ptr.data(), Hm.ptr.d, ((size_t)n + 1) * sizeof(int32_t), stream);
            staged_d2h(longs.data(), long_ent.d, (size_t)long_max * sizeof(int32_t), stream);
            HIP_CHECK(sync_stream(stream));
            pt.mark("    polish (device): kernels + row pointers back");
            } catch (...) {
                (void)sync_stream(stream);  // (nothing in flight may touch the scratch once it goes back to the cache)
                throw;
            }
        }
        if (res[2] > long_max) return false;  // (more long entries than the device list holds: the host loop has no such limit)
        hm_nnz = res[0];
        longs.resize((size_t)res[2]);
        std::sort(longs.begin(), longs.end());
        Q.long_ent = longs;
        Q.long_prob.assign(longs.size(), 0);
        {
            int pr = 0;
            for (size_t x = 0; x < longs.size(); ++x) {
                while (pr + 1 < h.count && (int64_t)longs[x] >= (int64_t)Q.Hm.ptr[(size_t)h.xoff[pr + 1]]) ++pr;
                Q.long_prob[x] = pr;
            }
        }
        Q.rbH = make_rowblocks(Q.Hm, plain_segments(h.xoff), h.count);
        Hm.adopt_tiles(Q.Hm, Q.rbH);
        pt.mark("    polish (device): tiles");
        Q.available = true;
        return true;
    }

    // Two parts.  init_polish_build: the structure of the Newton system (pattern, contribution lists, cone data) -- on the device it
    // ends in a round trip, so init() starts it BEFORE it waits for the host's band layout of K (which needs none of it; the wait
    // was 0.6 ms of a 4.2 ms headline create).  init_polish: buffers, tiles, band view of H, second level -- after K's data.
    bool polish_built_on_device = false;
    void init_polish_build(const HostSystem& h) {
        polish_built_on_device = false;
        PhaseTimer pt(st.verbose != 0);
        if (polish_build.valid()) polish_build.get();  // (rethrows what build_polish threw)
        pt.mark("  polish: host structures (wait)");
        bool on_device = false;
        if (h.device_setup) {
            // the per-cone structure (polish_structure) from the device matrices: what the host can see -- uniform second-order
            // cones of 2-4 rows -- is checked here, the head-variable conditions by k_polish_structure
            Q = PolishData();
            const size_t nc = h.cone_row.size();
            const int T = nc ? h.cone_dim[0] - 1 : 0;
            bool ok = nc > 0 && h.m_tot > 0 && T >= 1 && T <= kPolishMaxTail;
            for (size_t k = 0; k < nc && ok; ++k) ok = h.cone_type[k] == 1 && h.cone_dim[k] - 1 == T;
            if (!ok) return;
            Q.T = T;
            q_head.alloc(nc); q_aabs.alloc(nc); q_ck.alloc(nc); q_theta.alloc(nc); q_xstar.alloc(nc);
            DevBuf<int32_t> bad;
            {
                ZeroGroup zg;
                zg.add(q_ishead, (size_t)h.n_tot); zg.add(bad, 1);
                zg.commit(stream);
            }
            PStructArgs pa{};
            pa.ncones = (int64_t)nc; pa.n = h.n_tot; pa.T = T;
            pa.cone_row = cone_row.d; pa.cone_dim = cone_dim.d; pa.cone_type = cone_type.d;
            pa.A_ptr = A_ptr.d; pa.A_col = A_col.d; pa.A_val = A_val.d; pa.b = b.d; pa.q = q.d;
            pa.g2_ptr = G2.ptr.d; pa.g2_split = G2.split.d; pa.g2_col = G2.col.d; pa.g2_val = G2.val.d;
            pa.head_col = q_head.d; pa.is_head = q_ishead.d; pa.a_abs = q_aabs.d; pa.ck = q_ck.d; pa.theta = q_theta.d; pa.xstar = q_xstar.d;
            pa.bad = bad.d;
            hipLaunchKernelGGL(k_polish_structure, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, pa);
            std::vector<int32_t> bad_h(1, 0);
            HIP_CHECK(hipMemcpyAsync(bad_h.data(), bad.d, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            try {
                on_device = build_polish_on_device(h);  // (its synchronisation brings the verdict back too)
            } catch (const std::exception& e) {
                if (std::strstr(e.what(), "hipMalloc") == nullptr) throw;
                (void)hipGetLastError();
                on_device = false;
            }
            pt.mark("  polish: structure + pattern + lists on the device");
            HIP_CHECK(sync_stream(stream));
            if (!on_device || bad_h[0] != 0) {  // (not the SCORE structure, or a build the device declined: ADMM alone)
                Q.available = false;
                return;
            }
        } else
        if (polish_on_device && Q.T > 0) {
            q_head.upload(Q.head_col); q_ishead.upload(Q.is_head); q_aabs.upload(Q.a_abs); q_ck.upload(Q.ck);
            q_theta.upload(Q.theta); q_xstar.upload(Q.xstar);
            try {
                on_device = build_polish_on_device(h);
            } catch (const std::exception& e) {
                // the scratch of the device build (~60 bytes per record, sized by an upper bound) did not fit: the host loop
                // needs none of it.  Anything else is a real failure.
                if (std::strstr(e.what(), "hipMalloc") == nullptr) throw;
                (void)hipGetLastError();
                on_device = false;
            }
            pt.mark("  polish: pattern + lists on the device");
            if (!on_device) {  // (cones not laid out row after row, more than 2^32 records: the host loop takes over)
                polish_on_device = false;
                build_polish(h, Q, st.verbose != 0, false);
            }
        }
        polish_built_on_device = on_device;
    }
    void init_polish(const HostSystem& h) {
        PhaseTimer pt(st.verbose != 0);
        const bool on_device = polish_built_on_device;
        if (!Q.available) return;
        if (!on_device) {
            hm_nnz = (int64_t)Q.Hm.col.size();
            UploadBatch ub;
            Hm.upload(Q.Hm, Q.rbH, nullptr, false);
            q_Pon.upload(Q.Pon); q_ccoef.upload(Q.ccoef); q_cptr.upload(Q.cptr); q_ccone.upload(Q.ccone); q_cab.upload(Q.cab);
            q_head.upload(Q.head_col); q_ishead.upload(Q.is_head); q_aabs.upload(Q.a_abs); q_ck.upload(Q.ck);
            q_theta.upload(Q.theta); q_xstar.upload(Q.xstar);
            q_posd.upload(Q.pos_diag); q_poss.upload(Q.pos_sub); q_diagpos.upload(Q.diag_pos);
        }
        pt.mark("  polish: uploads");
        std::optional<UploadBatch> ub_polish;  // (tables and zeroed buffers only from here to join_init_newton: nothing is launched)
        ub_polish.emplace();
        Hb.upload(std::move(Q.band));
        if (st.verbose)
            std::fprintf(stderr, "[score setup] band view of H: %s (%d band + %d csr + %d diag tiles, %d slots per row)\n", Hb.on ? "on" : "off",
                         Hb.L.n_band, Hb.L.n_csr, Hb.L.n_diag, Hb.L.S);
        q_hblk_part.upload(Hb.on ? Hb.L.part_ptr : Q.rbH.part_ptr);
        {   // entry range of every problem in H (k_hassemble runs problem by problem)
            std::vector<int64_t> ep((size_t)h.count + 1);
            q_ent_max = 0;
            for (int p = 0; p <= h.count; ++p) ep[(size_t)p] = Q.Hm.ptr[(size_t)h.xoff[p]];
            for (int p = 0; p < h.count; ++p) q_ent_max = std::max(q_ent_max, ep[(size_t)p + 1] - ep[(size_t)p]);
            q_entpart.upload(ep);
        }
        n_long = (int)Q.long_ent.size();  // (found by build_polish while it lays the contribution lists out)
        q_long.upload(Q.long_ent); q_long_prob.upload(Q.long_prob);
        const size_t nc = h.cone_row.size();
        q_Bbuf.alloc(nc * Q.T * Q.T);
        ZeroGroup zq;
        zq.add(q_act, nc);
        n_fpart = 2 * std::max<size_t>((nc + kThreads - 1) / kThreads, (size_t)n_cone_blocks);  // F partials, then active-set flips
        q_X0.alloc(h.n_tot + h.m_tot); q_X1.alloc(h.n_tot + h.m_tot);
        q_g.alloc(h.n_tot); q_delta.alloc(h.n_tot); q_dummy.alloc(h.n_tot); q_negg.alloc(h.n_tot);
        q_dinv.alloc(h.dinv.size());
        zq.add(q_fac, h.fac_doubles_H);  // separator slots of the spike region are never written (nor used)
        if (newton_fac32) zq.add(q_fac32, h.fac_doubles_H);
        zq.commit(stream);
        if (newton_fac32 && prec_reg) deepH.alloc((size_t)std::max<int64_t>(1, h.deep_floats_H));
        n_gd = std::max<size_t>((h.n_tot + kThreads - 1) / kThreads, (size_t)Hm.nblocks);
        q_pw.alloc(hblocks());
        {
            q_gate_ref.alloc(h.count);
            std::vector<int64_t> sb(2 * h.count), se(2 * h.count);
            for (int p = 0; p < h.count; ++p) {
                sb[2 * p] = h.xoff[p]; se[2 * p] = h.xoff[p + 1];
                sb[2 * p + 1] = h.n_tot + h.roff[p]; se[2 * p + 1] = h.n_tot + h.roff[p + 1];
            }
            q_seg_begin.upload(sb); q_seg_end.upload(se);
        }
        ub_polish.reset();
        join_init_newton(h);
    }

    PolishArgs polish_args(double* X) {
        PolishArgs a{};
        const HostSystem& h = *H;
        a.ncones = (int)h.cone_row.size(); a.T = Q.T;
        a.cone_row = cone_row.d; a.head_col = q_head.d; a.a_abs = q_aabs.d; a.ck = q_ck.d; a.theta = q_theta.d; a.xstar = q_xstar.d;
        a.A_ptr = A_ptr.d; a.A_col = A_col.d; a.A_val = A_val.d; a.b = b.d;
        a.u = X; a.nu = X + h.n_tot; a.Bbuf = q_Bbuf.d; a.fpart = q_fpart.d; a.n_tot = h.n_tot;
        a.act = q_act.d; a.flip_part = q_fpart.d + n_fpart / 2; a.reref = q_reref.d;
        return a;
    }

    // Any failure inside the polish (NaN, HIP error) leaves the ADMM state untouched -- the Newton
    // loop only writes scratch vectors until its final hand-over -- and ADMM simply continues.
    bool polish_available() const { return Q.available; }
    int newton_limit = 0;  // > 0: at most this many Newton iterations per polish call (intermediate iterates)
    void set_newton_limit(int k) { newton_limit = k; }

    bool polish(const HostSystem& h, const score_settings& s_, const std::vector<int>& done_host, int* newton_iters,
                int* cg_used, const std::vector<double>* dual_scale) {
        try {
            return polish_lockstep(h, s_, done_host, newton_iters, cg_used, dual_scale);
        } catch (const std::exception& e) {
            if (st.verbose) std::fprintf(stderr, "[score] polish abandoned: %s\n", e.what());
            release_prequeued();  // (a fetch waiting for the host's words would keep the stream from draining)
            (void)sync_stream(stream);
            (void)hipGetLastError();
            return false;
        }
    }

    // Generalised Hessian from the cone blocks of the last evaluation (q_Bbuf), the Jacobi diagonal
    // and the chain factors of the Newton preconditioner (device-side factorisation)
    void launch_hassemble() {
        const HostSystem& h = *H;
        HAsmArgs ha{};
        ha.nnz = hm_nnz; ha.Pon = q_Pon.d; ha.cptr = q_cptr.d; ha.ccone = q_ccone.d; ha.cab = q_cab.d;
        ha.ccoef = q_ccoef.d; ha.Bbuf = q_Bbuf.d; ha.T2 = Q.T * Q.T; ha.Hval = Hm.val.d;
        ha.dst = Hb.on ? Hb.dst.d : nullptr; ha.V = Hb.on ? Hb.V.d : nullptr;
        ha.ndiag = (int)h.diag_cols.size(); ha.diag_pos = q_diagpos.d; ha.dinv = q_dinv.d;
        ha.ent_part = q_entpart.d; ha.skip = q_skip.d;  // (the live mask: a frozen problem's matrix is not read any more)
        const int base_blocks = (int)((q_ent_max + kThreads - 1) / kThreads);
        hipLaunchKernelGGL(k_hassemble, dim3((unsigned)((int64_t)base_blocks * h.count + n_long)), dim3(kThreads), 0, stream, ha, (const int32_t*)q_long.d,
                           (const int32_t*)q_long_prob.d, base_blocks, h.count);
        HIP_CHECK(hipGetLastError());
    }
    void newton_hessian(const int32_t* skip = nullptr, bool refactor = true) {
        if (!hassemble_queued) launch_hassemble();  // (queued ahead by prequeue_control() otherwise)
        hassemble_queued = false;
        if (n_prec_items() && refactor) {  // chain factors and the reciprocal Jacobi diagonal, one launch
            FactorArgs fa{};
            fa.work = prec_work.d; fa.chains = chainsH.d; fa.levels = levelsH.d; fa.Hval = Hm.val.d;
            fa.pos_diag = q_posd.d; fa.pos_sub = q_poss.d; fa.fac = q_fac.d; fa.work_mat = q_work.d; fa.skip = skip;
            fa.diag_pos = q_diagpos.d; fa.dinv = q_dinv.d;
            launch_factor(fa, n_prec_items(), true);
        }
    }

    // ---- Newton polish, all problems of the handle in lock-step (a single problem is a batch of
    //      one): every live problem takes its Newton step through the same launches; per-problem
    //      F, |g|, step length and state live on the host, the kernels read per-problem step lengths
    //      and skip flags from device arrays.  The PCG solves terminate ON THE DEVICE (pcg_gate in
    //      score_kernels.hpp): the host queues as many PCG iterations as the previous Newton step
    //      needed plus a margin, then the trial point and its evaluation, and synchronises ONCE per
    //      Newton iteration to read F, |g|, g'delta and the PCG flags together. ----
    BatchTables batch_tables() const {
        BatchTables bt{};
        bt.cone_block_first = cone_block_first.d; bt.cone_block_prob = cone_block_prob.d;
        bt.row_first = Hm.first_row.d; bt.row_prob = Hm.blk_prob.d;
        bt.skip = q_skip.d; bt.step = q_step.d;
        return bt;
    }
    // one upload of the per-problem control block [step | tol2 | skip, reref, fskip]
    //   reref: the problem's chain factors are recomputed in this Newton iteration (PolishArgs::reref);
    //   fskip: ... are not (the skip flags of the factor kernels).  Both return to "no" after one upload.
    std::vector<double> c_step, c_tol2;
    std::vector<int32_t> c_skip, c_reref;
    void fill_control(char* v) {
        const size_t c = c_skip.size();
        std::memcpy(v, c_step.data(), c * sizeof(double));
        std::memcpy(v + c * sizeof(double), c_tol2.data(), c * sizeof(double));
        int32_t* w = (int32_t*)(v + 2 * c * sizeof(double));
        std::memcpy(w, c_skip.data(), c * sizeof(int32_t));
        for (size_t i = 0; i < c; ++i) { w[c + i] = c_reref[i]; w[2 * c + i] = c_reref[i] ? 0 : 1; }
    }
    // `consume`: the upload that opens a Newton iteration -- if prequeue_control() queued its fetch (and the Hessian assembly
    // behind it) before the host's wait, the words go into that slot and its flag is raised: no launch.  Any other upload
    // finds a pending slot released with "skip everything" first (the assembly behind it then does nothing).
    void upload_control(bool consume = false) {
        const size_t c = c_skip.size();
        if (c_reref.size() != c) c_reref.assign(c, 0);
        if (pre_slot) {
            if (consume) {
                fill_control(pre_slot);
                raise_prequeued();
                hassemble_queued = true;
                std::fill(c_reref.begin(), c_reref.end(), 0);
                return;
            }
            release_prequeued();
        }
        char* v = next_ring_slot(ctl.n * sizeof(double));
        fill_control(v);
        fetch_words((int32_t*)ctl.d, v, (int)(4 * c + 3 * c));
        std::fill(c_reref.begin(), c_reref.end(), 0);
    }
    // ---- the next Newton iteration's control fetch + Hessian assembly, queued before the host waits for this one's results:
    //      the device goes on a PCIe round trip after the host has decided instead of a launch latency after it (the assembly
    //      covers the launches that follow).  Between prequeue_control() and the raise / release the host calls nothing that
    //      waits for the stream: wait_published() polls memory (its fall-back releases first), next_ring_slot() is not called.
    char* pre_slot = nullptr;
    unsigned long long pre_expect = 0;
    bool hassemble_queued = false;
    size_t pre_flag_offset() const { return (ctl.n * sizeof(double) + 7) & ~(size_t)7; }
    void prequeue_control() {
        if (pre_slot || std::getenv("SCORE_NO_PREQUEUE") != nullptr) return;  // (read at every call: the GPU test switches it in-process)
        const size_t c = c_skip.size();
        char* v = next_ring_slot(pre_flag_offset() + 8);
        __atomic_store_n((unsigned long long*)(v + pre_flag_offset()), 0ull, __ATOMIC_RELEASE);
        pre_slot = v;
        pre_expect = ++pre_seq_next;
        const char* d = d_ring + (v - h_ring);
        hipLaunchKernelGGL(k_fetch_wait, dim3(1), dim3(kThreads), 0, stream, (const int32_t*)d, (int32_t*)ctl.d, (int)(7 * c),
                           (const unsigned long long*)(d + pre_flag_offset()), pre_expect);
        launch_hassemble();
    }
    unsigned long long pre_seq_next = 0;
    void raise_prequeued() {
        __atomic_store_n((unsigned long long*)(pre_slot + pre_flag_offset()), pre_expect, __ATOMIC_RELEASE);
        pre_slot = nullptr;
    }
    void release_prequeued() {  // "skip everything": the queued assembly finds no problem to work on
        if (!pre_slot) return;
        const size_t c = c_skip.size();
        double* v = (double*)pre_slot;
        for (size_t i = 0; i < c; ++i) { v[i] = 1.0; v[c + i] = 0.0; }
        int32_t* w = (int32_t*)(pre_slot + 2 * c * sizeof(double));
        for (size_t i = 0; i < c; ++i) { w[i] = 1; w[c + i] = 0; w[2 * c + i] = 1; }
        raise_prequeued();
        hassemble_queued = false;
    }
    void upload_skip(const std::vector<char>& live, bool consume = false) {  // skip = !live
        for (size_t i = 0; i < live.size(); ++i) c_skip[i] = live[i] ? 0 : 1;
        upload_control(consume);
    }
    void upload_flags(const std::vector<int32_t>& f) {  // nonzero = selected
        c_skip = f;
        upload_control();
    }

    // queue the evaluation of F and the gradient at Xbuf for the problems not skipped
    void newton_eval_enqueue(double* Xbuf) {
        PolishArgs pa = polish_args(Xbuf);
        hipLaunchKernelGGL(k_newton_cone_b, dim3(n_cone_blocks), dim3(kThreads), 0, stream, pa, batch_tables());
        SpmvArgs ga = spmv_args(G2, Xbuf);
        // (-g goes to its own buffer, not to the PCG residual r)
        ga.is_head = q_ishead.d; ga.gout = q_g.d; ga.r = q_negg.d; ga.done = q_skip.d;
        const unsigned ggrid = xcd_grid(ga, G2.nblocks);
        hipLaunchKernelGGL(k_spmv<MODE_GRAD>, dim3(ggrid), dim3(kThreads), 0, stream, ga);
        // F and gradient partials land in host-mapped memory as the kernels write them; the PCG gate
        // words (device-resident: the kernels read them) are pushed, then the sequence number
        eval_seq = publish(q_pcgdone.d, d_gate_host, ((size_t)2 * H->count + 1) / 2);
    }
    unsigned long long eval_seq = 0;
    std::vector<double> act_flips;  // per problem: cones whose activity differs from the last factorisation's (last evaluation)
    // after the synchronisation: F and |grad|_inf of the problems in `which`
    void newton_eval_collect(const std::vector<char>& which, std::vector<double>& F, std::vector<double>& gn) {
        const HostSystem& h = *H;
        if ((int)act_flips.size() != h.count) act_flips.assign((size_t)h.count, 0.0);
        for (int p = 0; p < h.count; ++p) {
            if (!which[p]) continue;
            double f = 0.0, gmax = 0.0;
            double fl = 0.0;
            for (int b = h.cone_part_ptr[p]; b < h.cone_part_ptr[p + 1]; ++b) { f += h_newton[b]; fl += h_newton[n_fpart / 2 + b]; }
            act_flips[p] = fl;
            for (int bl = h.rbG2.part_ptr[p]; bl < h.rbG2.part_ptr[p + 1]; ++bl) {
                const double* o = h_dres + (size_t)bl * kPartStride;
                gmax = (o[0] != o[0]) ? o[0] : std::max(gmax, o[0]);
                f += o[2];
            }
            F[p] = f;
            gn[p] = gmax;
        }
    }
    void newton_eval_batch(double* Xbuf, const std::vector<char>& which, std::vector<double>& F, std::vector<double>& gn) {
        newton_eval_enqueue(Xbuf);
        wait_published(eval_seq);
        newton_eval_collect(which, F, gn);
    }

    // Queue PCG iterations on H delta = -g for the problems in `live`, each to its own relative
    // tolerance eta[p]; a problem's launches turn into no-ops once its gate has fired (q_skip[p] is
    // raised by the device).  No host synchronisation.
    // host-side cost of queueing the Newton PCG (SCORE_TRACE=host: printed when the handle goes)
    double enq_ms = 0.0, wait_ms = 0.0;
    long enq_launches = 0, waits = 0;
    int pcg_steps_queued = 0;
    double* pcg_rz_cur = nullptr;
    double* pcg_p_cur = nullptr;
    double* pcg_p_oth = nullptr;
    PrecArgs pcg_pa{};
    // start a PCG solve (INIT + first product); the control words -- skip flags, tolerances -- were uploaded by the caller
    void newton_pcg_begin() {
        const HostSystem& h = *H;
        PrecArgs& pa = pcg_pa;
        pa = PrecArgs{};
        pa.work = prec_work.d; pa.chains = chainsH.d; pa.levels = levelsH.d; pa.rec = prec_recH.d; pa.fac = q_fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = q_dinv.d; pa.done = q_skip.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = q_hblk_part.d; pa.uni = uni_for(hblocks());
        pa.r = r.d; pa.r_in = r.d; pa.z = z.d; pa.w = w.d; pa.xt = q_delta.d; pa.kx = q_dummy.d; pa.pw_part = q_pw.d;
        pa.gate_used = q_gate_used.d;
        pa.early_done = 1;  // launches queued beyond the gate are no-ops: keep them cheap
        {
            // the right-hand side is read where the evaluation left it (-g in q_negg) and the solution
            // starts from zero without a memset: the first STEP writes r and delta
            pcg_rz_cur = rz_part0.d; pcg_p_cur = p.d; pcg_p_oth = p2.d;
            pcg_steps_queued = 0;
            gate_epoch = (gate_epoch + 1) & 0x7ffff;
            if (gate_epoch == 0) gate_epoch = 1;
            pa.p = pcg_p_cur; pa.rz_in = nullptr; pa.rz_out = pcg_rz_cur;
            pa.r_in = q_negg.d;
            pa.gate_init = q_pcgdone.d;  // gate flags start as the host's skip flags
            launch_prec<PREC_INIT>(pa);
            pa.gate_init = nullptr;
            SpmvArgs a = spmv_args(Hm, pcg_p_cur);
            a.p = pcg_p_cur; a.pw_part = q_pw.d; a.done = q_skip.d;
            launch_h<MODE_KP>(a);
        }
        pa.done = q_pcgdone.d;
        pa.gate_flag = q_pcgdone.d; pa.gate_tol2 = q_gate_tol2.d; pa.gate_ref = q_gate_ref.d;
        pa.gate_host = d_gate_live; pa.gate_epoch = gate_epoch; pa.gate_count = h.count;
        pcg_first = true;
    }
    bool pcg_first = false;
    // one PCG iteration: STEP (delta += a p ; r -= a w ; z = M^-1 r -- or: the gate fires, nothing happens), then the product
    void newton_pcg_pair() {
        PrecArgs& pa = pcg_pa;
        double* rz_nxt = (pcg_rz_cur == rz_part0.d) ? rz_part1.d : rz_part0.d;
        const bool first = pcg_first;
        pcg_first = false;
        pa.p = pcg_p_cur; pa.rz_in = pcg_rz_cur; pa.rz_out = rz_nxt;
        pa.gate_first = first ? 1 : 0;
        pa.r_in = first ? q_negg.d : r.d;
        pa.xt_zero = first ? 1 : 0;
        launch_prec<PREC_STEP>(pa, probe_slot(1, pcg_steps_queued));
        tev = nullptr;
        SpmvArgs a = spmv_args(Hm, pcg_p_cur);
        a.p = pcg_p_cur; a.z = z.d; a.p_out = pcg_p_oth; a.rz_new = rz_nxt; a.rz_old = pcg_rz_cur; a.pw_part = q_pw.d; a.done = q_pcgdone.d;
        a.early_done = 1;
        launch_h<MODE_KPB>(a, probe_slot(0, pcg_steps_queued));
        tev = nullptr;
        std::swap(pcg_p_cur, pcg_p_oth);
        if (++pcg_steps_queued % kDirectEvery == 0) {  // (see linear_solve_core)
            SpmvArgs d = spmv_args(Hm, pcg_p_cur);
            d.p = pcg_p_cur; d.pw_part = q_pw.d; d.done = q_pcgdone.d; d.early_done = 1;
            launch_h<MODE_KP>(d);
        }
        pcg_rz_cur = rz_nxt;
    }
    // The PCG queued a few iterations AHEAD of the device instead of to a guessed length: pcg_gate's lead workgroups publish
    // "fired" and "STEPs executed" in host-mapped memory as they go (PrecArgs::gate_host); the host keeps `depth` iterations
    // in the queue beyond what has executed and stops at the first look that shows every live problem's gate fired.  At most
    // `depth` launches pairs run as no-ops (the guessed queue: a third of the launches on the headline solve), none is ever
    // missing (no resume).  Returns the iterations queued.
    int newton_pcg_follow(const std::vector<char>& live, int cap) {
        const HostSystem& h = *H;
        // (iterations kept in the queue beyond what the device has executed: 3 when this thread spins -- 2-5 measured alike,
        //  TRIED.md --, kEconomyDepth when it sleeps between looks: 25 us are most of a PCG iteration)
        const bool economy = economy_waits();
        const int depth = economy ? kEconomyDepth : 3;
        newton_pcg_begin();
        int queued = 0;
        const auto t0 = std::chrono::steady_clock::now();
        long long slept_ns = 0;
        struct SpinAccount {
            std::chrono::steady_clock::time_point t0; long long& slept;
            ~SpinAccount() {
                const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                wait_stats().spin_ns.fetch_add(std::max(0LL, ns - slept), std::memory_order_relaxed);
            }
        } spin_account{t0, slept_ns};
        unsigned spins = 0;
        for (;;) {
            bool all = true;
            int used = 1 << 30;
            for (int p = 0; p < h.count; ++p) {
                if (!live[p]) continue;
                const int32_t f = __atomic_load_n(&h_gate_live[p], __ATOMIC_ACQUIRE), u = __atomic_load_n(&h_gate_live[h.count + p], __ATOMIC_ACQUIRE);
                if (f != gate_epoch) {
                    all = false;
                    used = std::min(used, (u >> 12) == gate_epoch ? (int)(u & 0xfff) : 0);  // (the slowest live problem that still runs)
                }
            }
            if (all || queued >= cap) break;
            if (queued - used < depth) {
                newton_pcg_pair();
                ++queued;
                spins = 0;
                continue;
            }
            if (economy) {  // the queue is full: nothing to do for most of a PCG iteration
                const auto s0 = std::chrono::steady_clock::now();
                economy_sleep();
                slept_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - s0).count();
                spins += 1023;
            } else {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            if ((++spins & 1023) == 0) {
                HIP_CHECK(hipGetLastError());
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {  // (a device that stopped answering: let the caller's wait report it)
                    break;
                }
                if (spins > (1u << 14)) sched_yield();
            }
        }
        enq_launches += 2 * queued + 2;
        return queued;
    }

    // dual_scale (optional): per problem |A'y|_inf -- Newton then stops at half the tolerance the
    // driver's dual residual test will apply, eps_abs + eps_rel |A'y|, instead of its own absolute one
    bool polish_lockstep(const HostSystem& h, const score_settings& s_, const std::vector<int>& done_host,
                         int* newton_iters, int* cg_used, const std::vector<double>* dual_scale) {
        *newton_iters = 0; *cg_used = 0;
        pcg_used_total = 0;
        const double t_start = now_ms();
        release_prequeued();  // (nothing is pending between solves: every exit below releases; this is the belt to those braces)
        hassemble_queued = false;
        c_step.assign(h.count, 1.0); c_tol2.assign(h.count, 0.0); c_skip.assign(h.count, 0);
        if (!Q.available) return false;
        const int count = h.count, nbh = Hm.nblocks;
        double* X = q_X0.d;   // current point [u | nu]
        double* Xt = q_X1.d;  // trial point
        std::vector<char> part(count);
        bool any = false;
        for (int p = 0; p < count; ++p) { part[p] = done_host[p] ? 0 : 1; any = any || part[p]; }
        if (!any) return false;
        NewtonVecArgs va{};
        va.n = h.n_tot; va.is_head = q_ishead.d; va.g = q_g.d; va.part = q_gd.d;
        // start from the ADMM iterate x of every problem (head variables eliminated: kept at zero)
        HIP_CHECK(hipMemsetAsync(q_g.d, 0, q_g.n * sizeof(double), stream));
        va.u = xy.d; va.delta = xy.d; va.step = 0.0; va.out = X;
        hipLaunchKernelGGL(k_newton_trial, dim3((unsigned)((h.n_tot + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, va);
        std::vector<double> F(count, 0.0), gn(count, 0.0), Ft(count, 0.0), gt(count, 0.0), eta(count, 0.0);
        std::vector<double> step(count, 1.0), gd(count, 0.0);
        upload_skip(part);
        newton_eval_enqueue(X);
        prequeue_control();  // (the first iteration's control words and assembly: see the loop)
        wait_published(eval_seq);
        newton_eval_collect(part, F, gn);
        const double tol = std::max(1e-12, 0.3 * s_.eps_abs);
        std::vector<double> tolp(count, tol);
        if (dual_scale)
            for (int p = 0; p < count; ++p) tolp[p] = std::max(tol, 0.5 * (s_.eps_abs + s_.eps_rel * (*dual_scale)[p]));
        std::vector<char> live(count), stalled(count, 0);
        int it = 0;
        const int it_max = newton_limit > 0 ? std::min(50, newton_limit) : 50;
        for (; it < it_max; ++it) {
            any = false;
            for (int p = 0; p < count; ++p) { live[p] = part[p] && !stalled[p] && gn[p] > tolp[p]; any = any || live[p]; }
            if (!any) break;
            // inexact Newton: the linear residual only has to shrink superlinearly with |g|
            // (and never more digits than the step needs to land below the tolerance)
            for (int p = 0; p < count; ++p) {
                const double superlinear = std::max(1e-8, newton_eta_coef * std::pow(gn[p], newton_eta_pow));
                const double enough = gn[p] > 0.0 ? 0.1 * tolp[p] / gn[p] : 1.0;
                eta[p] = std::min(newton_eta_max, std::max(superlinear, enough));
            }
            // ONE upload of the control words serves the whole iteration: skip = !live for the Hessian,
            // the PCG solve and the first trial point; the PCG tolerances; unit step lengths
            for (int p = 0; p < count; ++p) { c_tol2[p] = eta[p] * eta[p]; c_step[p] = 1.0; }
            // The chain factors of the preconditioner are recomputed only when the active set of a live problem has
            // moved since they were last computed (act_flips, counted by the evaluation kernel): in the last Newton
            // iterations the blocks of B change by the step alone and the factors of the previous iteration
            // precondition as well (same PCG counts; k_factor + k_fac_round + k_deep_pack are 60-150 us a time).
            constexpr double flip_tol = 0.0;
            // (decided problem by problem: what a problem computes never depends on its batch mates)
            bool refactor = false;
            c_reref.assign((size_t)count, 0);
            for (int p = 0; p < count; ++p) {
                c_reref[p] = live[p] && (it == 0 || act_flips[p] > flip_tol);
                refactor = refactor || c_reref[p];
            }
            if (st.verbose) {
                double mx = 0.0;
                int nre = 0;
                for (int p = 0; p < count; ++p) if (live[p]) { mx = std::max(mx, act_flips[p]); nre += c_reref[p]; }
                std::fprintf(stderr, "[score] newton it %d: active-set flips since the last factorisation (max over live problems) %.0f -> %d problems refactor\n", it + 1, mx, nre);
            }
            upload_skip(live, /*consume=*/true);
            np_newton_it = it;
            newton_hessian(q_fskip.d, refactor);  // (a frozen problem's short entries keep their values, see k_hassemble)
            newton_pcg_follow(live, 400);  // (queued a few iterations ahead of the device until every live problem's gate has fired)
            int used_now = 0;
            // backtracking per problem; a problem leaves the search when its step is accepted
            std::vector<char> ls = live, accepted(count, 0);
            for (int p = 0; p < count; ++p) step[p] = 1.0;
            for (int k = 0; k < 40; ++k) {
                if (k > 0) {
                    c_step = step;
                    upload_skip(ls);  // (uploads the step lengths too)
                }
                va.u = X; va.delta = q_delta.d; va.step = 0.0; va.out = Xt;
                hipLaunchKernelGGL(k_newton_trial_b, dim3(nbh), dim3(kThreads), 0, stream, va, batch_tables());
                newton_eval_enqueue(Xt);  // overwrites nu / B / g of the problems searched; copies gd and the gate words too
                // (the usual course -- full step accepted, next iteration -- starts with a control upload and the assembly of H
                //  from the blocks this evaluation leaves: both queued now, behind the evaluation, waiting for the host's words)
                if (k == 0 && it + 1 < it_max) prequeue_control();
                wait_published(eval_seq);  // the one wait of a Newton iteration (step 1 accepted)
                if (k == 0) {
                    used_now = 0;
                    for (int p = 0; p < count; ++p)
                        if (live[p]) used_now = std::max(used_now, (int)h_gate[count + p]);
                }
                newton_eval_collect(ls, Ft, gt);
                if (k == 0) {
                    for (int p = 0; p < count; ++p) {
                        gd[p] = 0.0;
                        if (!live[p]) continue;
                        for (int b = Q.rbH.part_ptr[p]; b < Q.rbH.part_ptr[p + 1]; ++b) gd[p] += h_gd[b];
                    }
                    pcg_used_total += used_now;
                    probe_mark(used_now);
                }
                std::vector<int32_t> acc_now(count, 0);
                bool any_acc = false, any_ls = false;
                for (int p = 0; p < count; ++p) {
                    if (!ls[p]) continue;
                    const bool armijo = Ft[p] <= F[p] + 1e-4 * step[p] * gd[p];
                    const bool tiny = std::fabs(step[p] * gd[p]) <= 1e-13 * std::max(1.0, std::fabs(F[p]));
                    if ((Ft[p] == Ft[p]) && (armijo || (tiny && gt[p] < gn[p]))) {
                        F[p] = Ft[p]; gn[p] = gt[p];
                        accepted[p] = 1; acc_now[p] = 1; any_acc = true;
                        ls[p] = 0;
                    } else {
                        step[p] *= 0.5;
                        any_ls = true;
                    }
                }
                if (any_acc) {
                    if (!any_ls && k == 0) {
                        std::swap(X, Xt);  // every live problem accepted its full step: trade the buffers ...
                        // ... and keep the points of the problems that did not move
                        bool others = false;
                        std::vector<int32_t> keep(count, 0);
                        for (int p = 0; p < count; ++p)
                            if (!live[p]) { keep[p] = 1; others = true; }
                        if (others && count > 1) {
                            upload_flags(keep);
                            hipLaunchKernelGGL(k_copy_segments, dim3(64, 2 * count), dim3(kThreads), 0, stream, X, (const double*)Xt,
                                               (const int64_t*)q_seg_begin.d, (const int64_t*)q_seg_end.d, (const int32_t*)q_skip.d);
                        }
                    } else {  // X <- Xt on the segments (unknowns and cone rows) of the accepted problems
                        upload_flags(acc_now);
                        hipLaunchKernelGGL(k_copy_segments, dim3(64, 2 * count), dim3(kThreads), 0, stream, X, (const double*)Xt,
                                           (const int64_t*)q_seg_begin.d, (const int64_t*)q_seg_end.d, (const int32_t*)q_skip.d);
                    }
                }
                if (!any_ls) break;
            }
            bool any_stalled = false;
            for (int p = 0; p < count; ++p)
                if (live[p] && !accepted[p]) { stalled[p] = 1; any_stalled = true; }
            if (st.verbose) {
                for (int p = 0; p < count; ++p)
                    if (live[p]) std::fprintf(stderr, "[score] newton it %d prob %d F %.12g |g| %.3e step %.3g pcg %d (conv %d) t %.3f ms%s\n", it + 1, p, F[p], gn[p], step[p], h_gate[count + p], h_gate[p], now_ms() - t_start, stalled[p] ? " (stalled)" : "");
            }
            if (any_stalled) {  // re-establish nu / B / g of the current point of the stalled problems
                std::vector<char> sv(stalled.begin(), stalled.end());
                for (int p = 0; p < count; ++p) sv[p] = sv[p] && live[p];
                upload_skip(sv);
                std::vector<double> fx(count), gx(count);
                newton_eval_batch(X, sv, fx, gx);
            }
        }
        *newton_iters = it;
        *cg_used = pcg_used_total;
        pcg_used_total = 0;
        // hand the polished points to the ADMM state of the problems that took part
        upload_skip(part);
        va.u = X; va.delta = X; va.step = 0.0; va.out = Xt;
        hipLaunchKernelGGL(k_polish_copy_x_b, dim3(nbh), dim3(kThreads), 0, stream, va, xy.d, xtu.d, batch_tables());
        FinishArgs fa2{};
        fa2.P = polish_args(X);
        fa2.x = xy.d; fa2.xt = xtu.d; fa2.s = this->s.d; fa2.y = xy.d + h.n_tot;
        hipLaunchKernelGGL(k_polish_finish_b, dim3(n_cone_blocks), dim3(kThreads), 0, stream, fa2, batch_tables());
        if (n_cone_blocks) hipLaunchKernelGGL(k_refresh_u, dim3(n_cone_blocks), dim3(kThreads), 0, stream, cone_args(xtu.d));
        {
            SpmvArgs a = spmv_args(K, xtu.d);
            a.p = xtu.d; a.w = kx.d; a.done = q_skip.d;
            launch_spmv<MODE_KP>(K, a);
        }
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipGetLastError());
        probe_collect();
        return true;
    }

    void time_kernel(const std::string& which, int reps, double* ms) {
        PrecArgs pa{};
        pa.work = prec_work.d; pa.chains = chains.d; pa.levels = levels.d; pa.rec = prec_rec.d; pa.fac = fac.d;
        pa.node_col = node_col.d; pa.diag_cols = diag_cols.d; pa.dinv = dinv.d; pa.done = done.d;
        pa.prec_part_ptr = prec_part_ptr.d; pa.kblk_part_ptr = kblk_part_ptr.d; pa.uni = uni_for(kblocks());
        pa.r = r.d; pa.r_in = r.d; pa.z = z.d; pa.p = p.d; pa.w = w.d; pa.xt = xtu.d; pa.kx = kx.d;
        pa.pw_part = pw_part.d; pa.rz_in = rz_part0.d; pa.rz_out = rz_part1.d;
        VecArgs va{};
        va.first_row = vb_first.d; va.end_row = vb_end.d; va.blk_prob = vb_prob.d; va.done = done.d;
        va.prec_part_ptr = prec_part_ptr.d; va.kblk_part_ptr = kblk_part_ptr.d;
        va.pw_part = pw_part.d; va.p = p.d; va.w = w.d; va.kx = kx.d; va.xt = xtu.d; va.x = xy.d;
        va.alpha_relax = st.alpha; va.rz_old = rz_part0.d; va.apply_alpha = 1;
        auto once = [&]() {
            if (which == "rhs") { SpmvArgs ra = spmv_args(G1, xtu.d); ra.apply_update = 1; launch_spmv<MODE_RHS>(G1, ra); }
            else if (which.rfind("prec_init:", 0) == 0) { pa.debug_skip = std::atoi(which.c_str() + 10); launch_prec<PREC_INIT>(pa); }
            else if (which == "prec_init") launch_prec<PREC_INIT>(pa);
            else if (which.rfind("prec_step:", 0) == 0) { pa.debug_skip = std::atoi(which.c_str() + 10); launch_prec<PREC_STEP>(pa); }
            else if (which == "prec_step") launch_prec<PREC_STEP>(pa);
            // (segmented long chains: the chain kernel alone / the second level alone -- k_join_solve + k_join_apply)
            else if (which == "prec_init_chain") { join_suspend = true; launch_prec<PREC_INIT>(pa); join_suspend = false; }
            else if (which == "prec_step_chain") { join_suspend = true; launch_prec<PREC_STEP>(pa); join_suspend = false; }
            else if (which == "join_init") { if (n_join_items) join_apply<PREC_INIT>(pa, false); }
            else if (which == "join_step") { if (n_join_items) join_apply<PREC_STEP>(pa, false); }
            else if (which == "kp") launch_kp(p.d);
            else if (which == "kpb") launch_kpb(p.d, p2.d, rz_part1.d, rz_part0.d);
            else if (which == "xupdate") hipLaunchKernelGGL(k_xupdate, dim3(n_vblocks), dim3(kThreads), 0, stream, va);
            else if (which == "cone") { if (n_cone_blocks) { ConeArgs ca = cone_args(xtu.d); ca.apply_alpha = 1; hipLaunchKernelGGL(k_cone, dim3(n_cone_blocks), dim3(kThreads), 0, stream, ca); } }
            else if (which == "nop") hipLaunchKernelGGL(k_nop, dim3(K.nblocks), dim3(kThreads), 0, stream, (int*)nullptr);
            else if (which == "nop1") hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, stream, (int*)nullptr);
            else if (which == "nop_load") hipLaunchKernelGGL(k_nop_load, dim3(K.nblocks), dim3(kThreads), 0, stream, K.blk_prob.d, done.d, (int*)nullptr);
            else throw std::runtime_error("unknown kernel name");
        };
        // problems that have converged are skipped by every kernel: time them as active
        const HostSystem& h = *H;
        std::vector<int32_t> zero(h.count, 0), keep(h.count);
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipMemcpyAsync(keep.data(), done.d, keep.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipMemcpyAsync(done.d, zero.data(), zero.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(sync_stream(stream));
        for (int i = 0; i < 5; ++i) once();
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipEventRecord(ev0, stream));
        for (int i = 0; i < reps; ++i) once();
        HIP_CHECK(hipEventRecord(ev1, stream));
        HIP_CHECK(hipEventSynchronize(ev1));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, ev0, ev1));
        *ms = (double)t / std::max(1, reps);
        HIP_CHECK(hipMemcpyAsync(done.d, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(sync_stream(stream));
    }

    // roofline probe: average launch duration of the KKT SpMV (w = K p), HIP
    // events on the stream the solver launches on
    void time_kkt(int reps, double* ms, double* bytes) {
        const HostSystem& h = *H;
        std::vector<double> hp(h.n_tot);
        for (int64_t i = 0; i < h.n_tot; ++i) hp[i] = 1.0 + 1e-3 * (double)(i % 7);
        HIP_CHECK(sync_stream(stream));
        staged_h2d(p.d, hp.data(), hp.size() * sizeof(double), stream);
        HIP_CHECK(sync_stream(stream));
        std::vector<int32_t> zero(h.count, 0), keep(h.count);
        HIP_CHECK(hipMemcpyAsync(keep.data(), done.d, keep.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipMemcpyAsync(done.d, zero.data(), zero.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(sync_stream(stream));
        for (int i = 0; i < 10; ++i) launch_kp(p.d);
        HIP_CHECK(sync_stream(stream));
        HIP_CHECK(hipEventRecord(ev0, stream));
        for (int i = 0; i < reps; ++i) launch_kp(p.d);
        HIP_CHECK(hipEventRecord(ev1, stream));
        HIP_CHECK(hipEventSynchronize(ev1));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, ev0, ev1));
        *ms = (double)t / std::max(1, reps);
        double bsum = 0;
        for (double v : h.kkt_bytes) bsum += v;
        *bytes = bsum;
        HIP_CHECK(hipMemcpyAsync(done.d, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_CHECK(sync_stream(stream));
    }
};

}  // namespace

// SO(d) rounding of `n` d x d blocks (row-major, contiguous), one block per lane (score_round.hpp)
template <int D>
__global__ __launch_bounds__(256) void k_round_so(const double* __restrict__ M, double* __restrict__ R,
                                                  int32_t* __restrict__ degenerate, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double m[D * D], r[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) m[k] = M[i * (D * D) + k];
    int32_t bad;
    if (D == 2) score::round_so2(m, r, &bad); else score::round_so3(m, r, &bad);
#pragma unroll
    for (int k = 0; k < D * D; ++k) R[i * (D * D) + k] = r[k];
    degenerate[i] = bad;
}

struct score_assembled {
    score::AssembledQP qp;
};

struct score_handle {
    score::Solver<HipBackend> solver;
};

// Local refinement after SCORE on the device (score_gn.hpp): state, blocks and gathers here, the damped
// normal equations through the linear-mode handle `lin` (its K0 values and right-hand side are written
// in place by the gather kernels; the step is read from its solution vector).
struct score_refine {
    score::GnProblem P;
    score_handle* lin = nullptr;
    int device = 0;
    double setup_ms = 0;
    DevBuf<int32_t> rel_i, rel_j, rng_a, rng_b, pri_l, hc_ptr, hc_slot, gc_ptr, gc_slot, is_diag;
    DevBuf<double> rel_t, rel_R, rel_kappa, rel_tau, rng_dist, rng_prec, pri_t, pri_prec, pin;
    DevBuf<double> u, ut, hblk, gblk, rhs, cost_part, gmax_part;
    int n_mblocks = 0, n_ublocks = 0, n_hblocks = 0, n_sblocks = 0;
    std::vector<double> part_host;

    HipBackend& be() { return lin->solver.be; }
    hipStream_t stream() { return lin->solver.be.stream; }
    score::GnDev dev() const {
        score::GnDev d{};
        d.Np = P.Np; d.Nl = P.Nl; d.n = P.n; d.n_rel = P.n_rel(); d.n_rng = P.n_rng(); d.n_pri = P.n_pri();
        d.rel_i = rel_i.d; d.rel_j = rel_j.d; d.rng_a = rng_a.d; d.rng_b = rng_b.d; d.pri_l = pri_l.d;
        d.rel_t = rel_t.d; d.rel_R = rel_R.d; d.rel_kappa = rel_kappa.d; d.rel_tau = rel_tau.d;
        d.rng_dist = rng_dist.d; d.rng_prec = rng_prec.d; d.pri_t = pri_t.d; d.pri_prec = pri_prec.d; d.pin = pin.d;
        return d;
    }
    void create(const score_graph& g, const score_settings* s) {
        const double t0 = score::now_ms();
        score::PhaseTimer pt(s && s->verbose != 0);
        score::gn_build(g, P);
        pt.mark("refine: pattern + contribution lists");
        score_problem pat{};
        pat.n = (int32_t)P.n; pat.m = 0;
        pat.P_rowptr = P.hptr.data(); pat.P_col = P.hcol.data();
        pat.block_size = 3; pat.n_chains = (int32_t)P.chain_ptr.size() - 1;
        pat.chain_ptr = P.chain_ptr.data(); pat.node_first_col = P.node_first_col.data();
        if (score_linear_create(&pat, s, &lin) != 0) throw std::runtime_error(g_err);
        pt.mark("refine: linear-mode handle");
        device = lin->solver.st.device;
        tl_copy_stream = stream();
        struct ArenaScope {  // the refinement's buffers come from (and go back with) the handle's arena
            explicit ArenaScope(DevArena* a) { tl_arena = a; }
            ~ArenaScope() { tl_arena = nullptr; }
        } arena_scope(&be().arena);
        rel_i.upload(P.rel_i); rel_j.upload(P.rel_j); rng_a.upload(P.rng_a); rng_b.upload(P.rng_b); pri_l.upload(P.pri_l);
        rel_t.upload(P.rel_t); rel_R.upload(P.rel_R); rel_kappa.upload(P.rel_kappa); rel_tau.upload(P.rel_tau);
        rng_dist.upload(P.rng_dist); rng_prec.upload(P.rng_prec); pri_t.upload(P.pri_t); pri_prec.upload(P.pri_prec);
        hc_ptr.upload(P.hc_ptr); hc_slot.upload(P.hc_slot); gc_ptr.upload(P.gc_ptr); gc_slot.upload(P.gc_slot);
        std::vector<int32_t> dg(P.hcol.size(), 0);
        for (int64_t i = 0; i < P.n; ++i) dg[(size_t)P.diag_pos[(size_t)i]] = 1;
        is_diag.upload(dg);
        // state: 2-D the unknowns themselves (theta, x, y per free pose; landmarks); 3-D [R | t] of every pose, landmarks
        pin.alloc(3); u.alloc((size_t)P.state_size()); ut.alloc((size_t)P.state_size()); rhs.alloc((size_t)P.n);
        hblk.alloc((size_t)std::max<int64_t>(1, P.hblk_size())); gblk.alloc((size_t)std::max<int64_t>(1, P.gblk_size()));
        n_mblocks = (int)std::max<int64_t>(1, (P.n_meas() + kThreads - 1) / kThreads);
        n_ublocks = (int)std::max<int64_t>(1, (P.n + kThreads - 1) / kThreads);
        n_sblocks = (int)std::max<int64_t>(1, (P.Np + P.Nl + kThreads - 1) / kThreads);
        n_hblocks = (int)std::max<int64_t>(1, ((int64_t)P.hcol.size() + kThreads - 1) / kThreads);
        cost_part.alloc((size_t)n_mblocks); gmax_part.alloc((size_t)n_ublocks);
        be().linear_buffers(lin->solver.H);
        pt.mark("refine: uploads + buffers");
        setup_ms = score::now_ms() - t0;
    }
    ~score_refine() { if (lin) score_destroy(lin); }

    // ---- the backend concept of gn_levenberg_marquardt ----
    double eval_at(const double* where, bool with_blocks) {
        if (P.dim == 2)
            hipLaunchKernelGGL(score::k_gn_blocks, dim3(n_mblocks), dim3(kThreads), 0, stream(), dev(), where, hblk.d, gblk.d,
                               cost_part.d, with_blocks ? 1 : 0);
        else
            hipLaunchKernelGGL(score::k_gn_blocks3, dim3(n_mblocks), dim3(kThreads), 0, stream(), dev(), where, hblk.d, gblk.d,
                               cost_part.d, with_blocks ? 1 : 0);
        part_host.resize((size_t)n_mblocks);
        HIP_CHECK(hipMemcpyAsync(part_host.data(), cost_part.d, (size_t)n_mblocks * sizeof(double), hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(sync_stream(stream()));
        double f = 0.0;
        for (double v : part_host) f += v;
        return f;
    }
    double eval_current(bool with_blocks) { return eval_at(u.d, with_blocks); }
    double eval_trial() {
        if (P.dim == 2)
            hipLaunchKernelGGL(score::k_gn_trial, dim3(n_ublocks), dim3(kThreads), 0, stream(), (const double*)u.d,
                               (const double*)be().xtu.d, ut.d, (int64_t)P.n);
        else  // retraction: R <- R Exp(omega), t <- t + v
            hipLaunchKernelGGL(score::k_gn_trial3, dim3(n_sblocks), dim3(kThreads), 0, stream(), (const double*)u.d,
                               (const double*)be().xtu.d, ut.d, (int64_t)P.Np, (int64_t)P.Nl);
        return eval_at(ut.d, false);
    }
    void accept() { std::swap(u.d, ut.d); }
    double assemble() {  // g = J'r (rhs = -g) from the blocks of the current point; returns |g|_inf
        hipLaunchKernelGGL(score::k_gn_gather_g, dim3(n_ublocks), dim3(kThreads), 0, stream(), (const int32_t*)gc_ptr.d,
                           (const int32_t*)gc_slot.d, (const double*)gblk.d, rhs.d, gmax_part.d, (int64_t)P.n);
        part_host.resize((size_t)n_ublocks);
        HIP_CHECK(hipMemcpyAsync(part_host.data(), gmax_part.d, (size_t)n_ublocks * sizeof(double), hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(sync_stream(stream()));
        double m = 0.0;
        for (double v : part_host) m = std::max(m, v);
        return m;
    }
    bool solve(double lambda, double rel_tol, int* used) {  // (J'J + lambda I) step = -g, step left in the handle's xtu
        hipLaunchKernelGGL(score::k_gn_gather_h, dim3(n_hblocks), dim3(kThreads), 0, stream(), (const int32_t*)hc_ptr.d,
                           (const int32_t*)hc_slot.d, (const double*)hblk.d, (const int32_t*)is_diag.d, lambda, be().K0d.d,
                           (int64_t)P.hcol.size());
        return be().linear_solve_core(lin->solver.H, rhs.d, rel_tol, 4000, used);
    }
    void run(const double* poses_in, const double* lms_in, int max_iters, double tol, double* poses_out, double* lms_out,
             score::GnInfo& info) {
        std::vector<double> u0((size_t)P.state_size());
        if (P.dim == 2) {
            for (int64_t p = 1; p < P.Np; ++p)
                for (int k = 0; k < 3; ++k) u0[(size_t)(3 * (p - 1) + k)] = poses_in[3 * p + k];
            for (int64_t l = 0; l < 2 * P.Nl; ++l) u0[(size_t)(3 * (P.Np - 1) + l)] = lms_in[l];
            HIP_CHECK(hipMemcpyAsync(pin.d, poses_in, 3 * sizeof(double), hipMemcpyHostToDevice, stream()));
        } else {  // the state is the input layout: [R | t] per pose, then the landmarks
            std::copy(poses_in, poses_in + 12 * P.Np, u0.begin());
            if (P.Nl) std::copy(lms_in, lms_in + 3 * P.Nl, u0.begin() + (std::ptrdiff_t)(12 * P.Np));
        }
        staged_h2d(u.d, u0.data(), u0.size() * sizeof(double), stream());
        HIP_CHECK(sync_stream(stream()));
        score::gn_levenberg_marquardt(*this, max_iters, tol, 1e-9, info);
        staged_d2h(u0.data(), u.d, u0.size() * sizeof(double), stream());
        HIP_CHECK(sync_stream(stream()));
        if (P.dim == 2) {
            for (int k = 0; k < 3; ++k) poses_out[k] = poses_in[k];
            for (int64_t p = 1; p < P.Np; ++p)
                for (int k = 0; k < 3; ++k) poses_out[3 * p + k] = u0[(size_t)(3 * (p - 1) + k)];
            for (int64_t l = 0; l < 2 * P.Nl; ++l) lms_out[l] = u0[(size_t)(3 * (P.Np - 1) + l)];
        } else {
            std::copy(u0.begin(), u0.begin() + (std::ptrdiff_t)(12 * P.Np), poses_out);
            if (P.Nl) std::copy(u0.begin() + (std::ptrdiff_t)(12 * P.Np), u0.end(), lms_out);
        }
    }
};

extern "C" {

void score_default_settings(score_settings* s) { score::default_settings(s); }

// Setup builds (and teardown frees) a few hundred megabytes of host vectors per handle.  With glibc's
// default thresholds each of them is an mmap / munmap pair with fresh page faults; raising the
// thresholds once keeps that memory in the process heap, where the next handle finds it again.
static void tune_host_allocator_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        // (glibc refuses an mmap threshold above HEAP_MAX_SIZE / 2 = 32 MiB and then keeps its default -- a
        //  larger request here used to be a silent no-op)
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
    });
}

int score_create_batch(const score_problem* p, int32_t count, const score_settings* s, score_handle** out) {
    try {
        ActiveSolve active;
        tune_host_allocator_once();
        if (!p || !out) throw std::runtime_error("null argument");
        score_settings st;
        if (s) st = *s; else score::default_settings(&st);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
        if (st.device < 0 || st.device >= ndev) throw std::runtime_error("score_settings.device out of range");
        DeviceGuard guard(st.device);
        auto* h = new score_handle();
        try {
            h->solver.create(p, count, st);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int score_create(const score_problem* p, const score_settings* s, score_handle** out) {
    return score_create_batch(p, 1, s, out);
}
static int score_create_from_graphs_impl(const score_graph* graphs, int32_t count, const score_settings* s, score_handle** out, const HipBackend::GenSource* gen_src) {
    try {
        ActiveSolve active;
        tune_host_allocator_once();
        if (!graphs || !out) throw std::runtime_error("null argument");
        score_settings st;
        if (s) st = *s; else score::default_settings(&st);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
        if (st.device < 0 || st.device >= ndev) throw std::runtime_error("score_settings.device out of range");
        DeviceGuard guard(st.device);
        auto* h = new score_handle();
        h->solver.be.gen_src = gen_src;
        struct ClearSrc { score_handle* h; ~ClearSrc() { h->solver.be.gen_src = nullptr; } };
        try {
            ClearSrc clear{h};
            h->solver.create_from_graphs(graphs, count, st, [&](const score_graph* gs, bool keep_hf) {
                std::vector<score::AssembledQP> qps((size_t)count);
                std::vector<score::AssembledQP*> ptrs((size_t)count);
                for (int i = 0; i < count; ++i) ptrs[(size_t)i] = &qps[(size_t)i];
                score::assemble_graphs(gs, count, ptrs.data());
                std::vector<score_problem> probs((size_t)count);
                for (int i = 0; i < count; ++i) qps[(size_t)i].view(&probs[(size_t)i]);
                h->solver.create(probs.data(), count, st, keep_hf);
            });
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int score_create_from_graphs(const score_graph* graphs, int32_t count, const score_settings* s, score_handle** out) {
    return score_create_from_graphs_impl(graphs, count, s, out, nullptr);
}
int score_read_estimates(score_handle* h, int32_t qcqp_directions, double* poses, double* relaxed, double* landmarks, double* ranges,
                         int32_t* degenerate) {
    try {
        if (!h) throw std::runtime_error("null handle");
        if (!h->solver.est.valid()) throw std::runtime_error("score_read_estimates: the handle was not made by score_create_from_graphs");
        DeviceGuard guard(h->solver.st.device);
        ActiveSolve active;
        h->solver.be.read_estimates(h->solver.H, h->solver.est, (qcqp_directions || h->solver.est.dirs_always) ? 1 : 0, poses, relaxed, landmarks, ranges, degenerate);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_graphs_connected(const score_graph* graphs, int32_t count) {
    if (!graphs || count < 0) { g_err = "null argument"; return -1; }
    for (int32_t i = 0; i < count; ++i)
        if (!score::graph_connected(graphs[i])) return i + 1;
    return 0;
}
int score_dims(const score_handle* h, int64_t* n_total, int64_t* m_total, int32_t* count) {
    if (!h) { g_err = "null handle"; return -1; }
    if (n_total) *n_total = h->solver.user_n();  // (the programs as given: score_headform.hpp)
    if (m_total) *m_total = h->solver.user_m();
    if (count) *count = h->solver.H.count;
    return 0;
}
int score_solve(score_handle* h, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        ActiveSolve active;
        return h->solver.solve(x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_reset(score_handle* h) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        h->solver.reset();
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_solve_steps(score_handle* h, int32_t iters, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        ActiveSolve active;
        return h->solver.steps(iters, x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_newton_steps(score_handle* h, int32_t iters, double* x, double* y, double* s, score_info* infos) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        ActiveSolve active;
        return h->solver.newton_steps(iters, x, y, s, infos);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_linear_create(const score_problem* pattern, const score_settings* s, score_handle** out) {
    try {
        if (!pattern || !out) throw std::runtime_error("null argument");
        score::LinearPattern L;
        score::make_linear_pattern(*pattern, s, L);
        score_handle* h = nullptr;
        if (score_create_batch(&L.prob, 1, &L.st, &h) != 0) return -1;
        auto& S = h->solver;
        if ((int64_t)S.H.K0.size() != (int64_t)pattern->P_rowptr[pattern->n]) {
            score_destroy(h);
            throw std::runtime_error("score_linear_create: internal pattern differs from the given one");
        }
        S.linear_mode = true;
        S.linear_nnz = (int64_t)S.H.K0.size();
        *out = h;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_linear_solve(score_handle* h, const double* values, const double* rhs, double* x, double rel_tol,
                       int32_t max_iters, int32_t* iters_used, double* rel_residual) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        int used = 0;
        const int rc = h->solver.linear_solve(values, rhs, x, rel_tol, max_iters, &used, rel_residual);
        if (iters_used) *iters_used = used;
        return rc;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_time_kkt_apply(score_handle* h, int32_t reps, double* ms, double* bytes) {
    try {
        if (!h) throw std::runtime_error("null handle");
        DeviceGuard guard(h->solver.st.device);
        h->solver.be.time_kkt(reps, ms, bytes);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_time_iteration(score_handle* h, int32_t warmup, int32_t iters, double* us, int32_t with_events) {
    try {
        if (!h || !us) throw std::runtime_error("null argument");
        DeviceGuard guard(h->solver.st.device);
        // the driver's reset: iterates, penalties AND the PCG count an adaptive solve may have raised
        h->solver.reset();
        h->solver.be.time_iteration(warmup, iters, us, with_events);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_debug_time(score_handle* h, const char* kernel, int32_t reps, double* ms) {
    try {
        if (!h || !kernel || !ms) throw std::runtime_error("null argument");
        DeviceGuard guard(h->solver.st.device);
        h->solver.be.time_kernel(kernel, reps, ms);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int64_t score_debug_get(score_handle* h, const char* name, double* out, int64_t len) {
    if (!h || !name) return -1;
    try {
        DeviceGuard guard(h->solver.st.device);
        return h->solver.be.get_vec(name, out, len);
    } catch (const std::exception& e) { g_err = e.what(); return -2; }
}
void score_destroy(score_handle* h) {
    if (!h) return;
    try {
        ActiveSolve active;
        DeviceGuard guard(h->solver.st.device);
        score::PhaseTimer pt(h->solver.st.verbose != 0);
        delete h;
        pt.mark("destroy: total");
    } catch (...) {
        delete h;
    }
}
int score_assemble(const score_graph* g, score_assembled** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        auto* a = new score_assembled();
        try {
            score::assemble_graph(*g, a->qp);
        } catch (...) {
            delete a;
            throw;
        }
        *out = a;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_assemble_batch(const score_graph* graphs, int32_t count, score_assembled** out) {
    try {
        if (!graphs || !out || count <= 0) throw std::runtime_error("null argument");
        std::vector<score_assembled*> made((size_t)count, nullptr);
        std::vector<score::AssembledQP*> qps((size_t)count, nullptr);
        try {
            for (int i = 0; i < count; ++i) { made[(size_t)i] = new score_assembled(); qps[(size_t)i] = &made[(size_t)i]->qp; }
            score::assemble_graphs(graphs, count, qps.data());
        } catch (...) {
            for (auto* a : made) delete a;
            throw;
        }
        for (int i = 0; i < count; ++i) out[i] = made[(size_t)i];
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_assembled_view(const score_assembled* a, score_problem* view) {
    if (!a || !view) { g_err = "null argument"; return -1; }
    a->qp.view(view);
    return 0;
}
void score_assembled_free(score_assembled* a) { delete a; }
int score_refine_create(const score_graph* g, const score_settings* s, score_refine** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        score_settings st;
        if (s) st = *s; else score::default_settings(&st);
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
        if (st.device < 0 || st.device >= ndev) throw std::runtime_error("score_settings.device out of range");
        DeviceGuard guard(st.device);  // the refinement's own buffers live on the handle's device too
        auto* r = new score_refine();
        try {
            r->create(*g, s);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_refine_run(score_refine* r, const double* poses_in, const double* landmarks_in, int32_t max_iters, double tol,
                     double* poses_out, double* landmarks_out, score_refine_info* info) {
    try {
        if (!r || !poses_in || !poses_out || (r->P.Nl > 0 && (!landmarks_in || !landmarks_out))) throw std::runtime_error("null argument");
        DeviceGuard guard(r->device);
        const double t0 = score::now_ms();
        score::GnInfo gi;
        r->run(poses_in, landmarks_in, max_iters, tol, poses_out, landmarks_out, gi);
        if (info) {
            info->cost_initial = gi.cost_initial; info->cost_final = gi.cost_final; info->grad_inf = gi.grad_inf;
            info->iterations = gi.iterations; info->linear_solves = gi.linear_solves; info->pcg_iters = gi.pcg_iters;
            info->setup_ms = r->setup_ms; info->solve_ms = score::now_ms() - t0;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void score_refine_destroy(score_refine* r) {
    if (!r) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(r->device);
    delete r;
    if (prev >= 0) (void)hipSetDevice(prev);
}
}  // extern "C"

// (the generated arrays stay in device memory for as long as the batch lives: score_create_from_generated builds handles
//  from them without another transfer)
struct score_generated {
    score::GeneratedBatch B;
    DevArena arena;
    int device = -1;
    const int32_t* d_rel_base = nullptr; const int32_t* d_rel_to = nullptr; const int32_t* d_ra = nullptr; const int32_t* d_rb = nullptr;
    const double* d_rel_t = nullptr; const double* d_rel_R = nullptr; const double* d_rel_kappa = nullptr; const double* d_rel_tau = nullptr;
    const double* d_dist = nullptr; const double* d_prec = nullptr;
    // (nothing to wait for when the batch goes: score_create_from_generated returns with its handle's setup -- the only reader of
    //  these arrays -- complete; the arena's blocks go back to the cache)
};
namespace {
// The generator on the device: walks + beacons (one thread per robot / beacon), the ranges counted per (trial, group, time),
// scanned, filled; the arrays come back through pinned staging (the host lays out the handles from them:
// score_create_from_graphs takes the views like any other score_graph).
void generate_manhattan_device(const score::GenSpec& S, int count, int device, score_generated& Gd) {
    using namespace score;
    GeneratedBatch& B = Gd.B;
    gen_check_spec(S, count);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available (the SCORE solver has no CPU fallback)");
    if (device < 0 || device >= ndev) throw std::runtime_error("score_generate_manhattan: device out of range");
    DeviceGuard guard(device);
    B = GeneratedBatch();
    B.S = S; B.count = count;
    B.size_fixed();
    const size_t R = (size_t)S.n_robots, T = (size_t)S.n_poses, E = (size_t)B.edges(), c = (size_t)count, G = (size_t)gen_groups(S), nb = (size_t)S.n_beacons;
    hipStream_t st = stream_pool().take(device);
    DevArena& arena = Gd.arena;
    arena.dev = device;
    Gd.device = device;
    struct Scope {
        DevArena* keep_a; hipStream_t keep_s;
        Scope(DevArena* a, hipStream_t s) : keep_a(tl_arena), keep_s(tl_copy_stream) { tl_arena = a; tl_copy_stream = s; }
        ~Scope() { tl_arena = keep_a; tl_copy_stream = keep_s; }
    } scope(&arena, st);
    try {
        DevBuf<int32_t> px, py, ph, pz, bx, by, bz, rel_base, rel_to, cnt, off, ra, rb;
        DevBuf<double> rel_t, rel_R, rel_kappa, rel_tau, dist, prec;
        const size_t dd = (size_t)S.dim;
        px.alloc(c * R * T); py.alloc(c * R * T); ph.alloc(c * R * T); bx.alloc(c * nb); by.alloc(c * nb);
        if (dd == 3) { pz.alloc(c * R * T); bz.alloc(c * nb); }
        rel_base.alloc(c * E); rel_to.alloc(c * E); rel_t.alloc(dd * c * E); rel_R.alloc(dd * dd * c * E); rel_kappa.alloc(c * E); rel_tau.alloc(c * E);
        const size_t n_cnt = c * G * T;
        cnt.alloc(n_cnt + 1); off.alloc(n_cnt + 1);
        GenArgs a{};
        a.S = S; a.count = count;
        a.px = px.d; a.py = py.d; a.ph = ph.d; a.bx = bx.d; a.by = by.d; a.pz = pz.d; a.bz = bz.d;
        a.rel_base = rel_base.d; a.rel_to = rel_to.d; a.rel_t = rel_t.d; a.rel_R = rel_R.d; a.rel_kappa = rel_kappa.d; a.rel_tau = rel_tau.d;
        a.cnt = cnt.d; a.off = off.d;
        const size_t n_walk = std::max(c * R, c * nb);
        hipLaunchKernelGGL(k_gen_walk, dim3((unsigned)((n_walk + 63) / 64)), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_gen_odom, dim3((unsigned)((c * E + 255) / 256)), dim3(256), 0, st, a);
        HIP_CHECK(hipMemsetAsync(cnt.d + n_cnt, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_gen_ranges<false>, dim3((unsigned)((n_cnt + 255) / 256)), dim3(256), 0, st, a);
        size_t tb = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, cnt.d, off.d, (int32_t)0, n_cnt + 1, rocprim::plus<int32_t>(), st));
        DevBuf<unsigned char> scratch;
        scratch.alloc(tb + 256);
        HIP_CHECK(rocprim::exclusive_scan((void*)scratch.d, tb, cnt.d, off.d, (int32_t)0, n_cnt + 1, rocprim::plus<int32_t>(), st));
        HIP_CHECK(hipGetLastError());
        // the trials' first ranges + the total: every (G T)-th entry of the scan
        std::vector<int32_t> offs(n_cnt + 1);
        staged_d2h(offs.data(), off.d, (n_cnt + 1) * sizeof(int32_t), st);
        for (size_t t = 0; t <= c; ++t) B.rng_first[t] = offs[t * G * T];
        const size_t total = (size_t)B.rng_first[c];
        B.size_ranges((int64_t)total);
        ra.alloc(total); rb.alloc(total); dist.alloc(total); prec.alloc(total);
        a.ra = ra.d; a.rb = rb.d; a.dist = dist.d; a.prec = prec.d;
        hipLaunchKernelGGL(k_gen_ranges<true>, dim3((unsigned)((n_cnt + 255) / 256)), dim3(256), 0, st, a);
        HIP_CHECK(hipGetLastError());
        // every array into ONE pinned block (all transfers queued, one wait), then out to the batch's vectors by the thread team
        struct Part { void* dst; const void* src; size_t bytes, off; };
        std::vector<Part> parts;
        size_t tot_bytes = 0;
        auto back = [&](auto& dst, const auto& src) {
            if (dst.empty()) return;
            const size_t nbytes = dst.size() * sizeof(dst[0]);
            parts.push_back(Part{dst.data(), src.d, nbytes, tot_bytes});
            tot_bytes += (nbytes + 255) & ~(size_t)255;
        };
        back(B.px, px); back(B.py, py); back(B.ph, ph); back(B.bx, bx); back(B.by, by); back(B.pz, pz); back(B.bz, bz);
        back(B.rel_base, rel_base); back(B.rel_to, rel_to); back(B.rel_t, rel_t); back(B.rel_R, rel_R); back(B.rel_kappa, rel_kappa); back(B.rel_tau, rel_tau);
        back(B.ra, ra); back(B.rb, rb); back(B.dist, dist); back(B.prec, prec);
        // (a trial's range endpoints are trial-local already: pose r * T + t, landmark Np + b)
        size_t got = std::max<size_t>(tot_bytes, 256);
        char* pin = (char*)block_cache().take(got, device, true);
        hipError_t err = hipSuccess;
        for (const Part& pt : parts) {
            const hipError_t e1 = hipMemcpyAsync(pin + pt.off, pt.src, pt.bytes, hipMemcpyDeviceToHost, st);
            if (e1 != hipSuccess) err = e1;
        }
        const hipError_t es = sync_stream(st);
        if (err == hipSuccess && es == hipSuccess)
            parallel_ranges((int64_t)parts.size(), 1, [&](int, int64_t p0, int64_t p1) {
                for (int64_t k = p0; k < p1; ++k) std::memcpy(parts[(size_t)k].dst, pin + parts[(size_t)k].off, parts[(size_t)k].bytes);
            });
        block_cache().give(pin, got, device, true);
        HIP_CHECK(err); HIP_CHECK(es);
        Gd.d_rel_base = rel_base.d; Gd.d_rel_to = rel_to.d; Gd.d_rel_t = rel_t.d; Gd.d_rel_R = rel_R.d; Gd.d_rel_kappa = rel_kappa.d; Gd.d_rel_tau = rel_tau.d;
        Gd.d_ra = ra.d; Gd.d_rb = rb.d; Gd.d_dist = dist.d; Gd.d_prec = prec.d;
    } catch (...) {
        (void)sync_stream(st);
        stream_pool().give(device, st);
        throw;
    }
    stream_pool().give(device, st);
}
}  // namespace

extern "C" {
int score_generate_manhattan(const score_manhattan_spec* spec, int32_t count, int32_t device, score_generated** out) {
    try {
        if (!spec || !out) throw std::runtime_error("null argument");
        score::GenSpec S{spec->n_robots, spec->n_poses, spec->n_beacons, spec->side, spec->p_range, spec->sigma_t, spec->sigma_theta, spec->sigma_range, spec->seed,
                         spec->dim == 0 ? 2 : spec->dim};
        auto* g = new score_generated();
        try { generate_manhattan_device(S, count, device, *g); } catch (...) { delete g; throw; }
        *out = g;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_generated_graph(const score_generated* g, int32_t index, score_graph* view) {
    try {
        if (!g || !view) throw std::runtime_error("null argument");
        g->B.view(index, view);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int score_generated_truth(const score_generated* g, int32_t index, double* poses, double* beacons) {
    try {
        if (!g) throw std::runtime_error("null argument");
        g->B.truth(index, poses, beacons);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void score_generated_free(score_generated* g) { delete g; }
int score_create_from_generated(const score_generated* g, int32_t first, int32_t count, int32_t relaxation, const score_settings* s, score_handle** out) {
    try {
        if (!g || !out) throw std::runtime_error("null argument");
        if (first < 0 || count <= 0 || first + count > g->B.count) throw std::runtime_error("score_create_from_generated: worlds out of range");
        if (relaxation != 0 && relaxation != 1) throw std::runtime_error("score_create_from_generated: relaxation must be 0 (SOCP) or 1 (QCQP)");
        std::vector<score_graph> views((size_t)count);
        for (int i = 0; i < count; ++i) { g->B.view(first + i, &views[(size_t)i]); views[(size_t)i].relaxation = relaxation; }
        score_settings st;
        if (s) st = *s; else score::default_settings(&st);
        HipBackend::GenSource src{};
        const bool resident = g->device >= 0 && st.device == g->device;
        if (resident) {  // the measurement arrays where the generator left them (world `first` onwards: the worlds follow each other)
            const size_t E = (size_t)g->B.edges(), eo = (size_t)first * E, ro = (size_t)g->B.rng_first[(size_t)first];
            const size_t dd = (size_t)g->B.S.dim;
            src.rel_base = g->d_rel_base + eo; src.rel_to = g->d_rel_to + eo; src.rel_t = g->d_rel_t + dd * eo; src.rel_R = g->d_rel_R + dd * dd * eo;
            src.rel_kappa = g->d_rel_kappa + eo; src.rel_tau = g->d_rel_tau + eo;
            src.rng_a = g->d_ra + ro; src.rng_b = g->d_rb + ro; src.rng_dist = g->d_dist + ro; src.rng_prec = g->d_prec + ro;
        }
        return score_create_from_graphs_impl(views.data(), count, &st, out, resident ? &src : nullptr);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
}  // extern "C"

extern "C" {
int score_round_to_so(int32_t dim, int64_t n, const double* blocks, double* rotations, int32_t* degenerate, int32_t device) {
    try {
        if (dim != 2 && dim != 3) throw std::runtime_error("score_round_to_so: dim must be 2 or 3");
        if (n < 0 || (n > 0 && (!blocks || !rotations || !degenerate))) throw std::runtime_error("score_round_to_so: null argument");
        if (n == 0) return 0;
        DeviceGuard guard(device);
        // staging in pinned, device-mapped host memory from the block cache: the kernel reads and writes it
        // across the link directly (1.4 MB each way for 20 000 3-D poses), no copy-engine submissions
        const size_t in_bytes = (size_t)n * dim * dim * sizeof(double), flag_bytes = (size_t)n * sizeof(int32_t);
        size_t cap = 2 * in_bytes + flag_bytes;
        char* stage = (char*)block_cache().take(cap, device, true);
        hipStream_t st = stream_pool().take(device);
        int rc = 0;
        try {
            std::memcpy(stage, blocks, in_bytes);
            char* dstage = nullptr;
            HIP_CHECK(hipHostGetDevicePointer((void**)&dstage, stage, 0));
            const double* dM = (const double*)dstage;
            double* dR = (double*)(dstage + in_bytes);
            int32_t* dF = (int32_t*)(dstage + 2 * in_bytes);
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (dim == 2) hipLaunchKernelGGL(k_round_so<2>, dim3(grid), dim3(256), 0, st, dM, dR, dF, (int64_t)n);
            else hipLaunchKernelGGL(k_round_so<3>, dim3(grid), dim3(256), 0, st, dM, dR, dF, (int64_t)n);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(sync_stream(st));
            std::memcpy(rotations, stage + in_bytes, in_bytes);
            std::memcpy(degenerate, stage + 2 * in_bytes, flag_bytes);
        } catch (const std::exception& e) { g_err = e.what(); rc = -1; }
        // After an error the kernel may still be running on the staging block: hand block and stream back only once
        // the stream has drained; if even that fails, drop them (a leak of one block beats a kernel writing into a
        // block another handle has been given).
        if (rc == 0 || sync_stream(st) == hipSuccess) {
            stream_pool().give(device, st);
            block_cache().give(stage, cap, device, true);
        } else {
            (void)hipGetLastError();
        }
        return rc;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int64_t score_trim_caches(void) {
    const size_t freed = block_cache().trim();
    stream_pool().trim();
    malloc_trim(0);  // the host heap kept by tune_host_allocator_once goes back to the system as well
    return (int64_t)freed;
}
int32_t score_host_counters(double* out, int32_t len) {
    const HostWaitStats& w = wait_stats();
    const double v[4] = {1e-6 * (double)w.spin_ns.load(), 1e-6 * (double)w.sleep_ns.load(), (double)w.waits.load(), (double)w.sleeps.load()};
    if (out) for (int i = 0; i < len && i < 4; ++i) out[i] = v[i];
    return 4;
}
const char* score_last_error(void) { return g_err.c_str(); }
int32_t score_abi_version(void) { return SCORE_ABI_VERSION * 1000 + (int32_t)sizeof(score_problem); }
const char* score_backend(void) { return "hip-gfx950"; }
}
