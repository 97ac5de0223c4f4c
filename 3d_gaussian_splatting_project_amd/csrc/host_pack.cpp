// host_pack.cpp — see host_pack.hpp.  Pure C++, built by g++ (AVX2 forms of the inner loops, chosen at run time).
#include "host_pack.hpp"

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
static inline void cpu_relax() { _mm_pause(); }
#else
static inline void cpu_relax() {}
#endif

namespace gsx {

MapLayout map_layout(int w, int h, bool tiled, bool coarse) {
    MapLayout L;
    L.w = w;
    L.h = h;
    const size_t npix = (size_t)w * (size_t)h;
    L.strip_bytes = tiled ? (h + 7) / 8 * 128 : 0;
    L.fine_bytes = tiled ? (size_t)L.strip_bytes * (size_t)((w + 15) / 16) : npix + 4;
    L.cw = (w + 3) / 4;
    L.ch = (h + 3) / 4;
    const bool c = tiled && coarse;
    L.cstrip_bytes = c ? (L.ch + 7) / 8 * 128 : 0;
    L.coarse_off = (L.fine_bytes + 255) / 256 * 256;
    L.map_bytes = c ? L.coarse_off + (size_t)L.cstrip_bytes * (size_t)((L.cw + 15) / 16) : L.fine_bytes;
    return L;
}

// ---- worker pool ---------------------------------------------------------------------------------------------------
// Fork-join whose cost is a handful of cache-line transfers (a 1080p map is packed in ~30 us, 200 times per run: round 2's
// first pool spent 3.5-4.4 us per call forking and joining, 10 % of the hand-over):
//   fork   the caller writes the job (one line) and bumps `generation` (the one line the idle workers poll).  Nothing per
//          worker is written: thread t owns the contiguous parts [parts*t/T, parts*(t+1)/T) and every thread derives any
//          thread's bounds itself.
//   claim  a thread's next unclaimed part lives in a 64-bit word on the thread's own line, tagged with the generation:
//          (generation << 32) | next.  A stale tag means "nothing claimed yet in this run", so nobody has to reset the
//          words; owner and helpers claim with the same compare-and-swap.  A thread works through its own share first, then
//          helps the others (a thread that shares its core, reads remote memory or was descheduled would otherwise hold the
//          whole call up).
//   join   every worker decrements one counter (a line of its own, touched once per worker and run) when it has nothing
//          left to claim; the caller spins on that line only.
// (One shared `next part` counter next to the flag the idle workers poll cost 0.3 ms per call on a 2-socket EPYC: every
// fetch_add fought the pollers for the line.)  Placement: see below.
struct alignas(64) Slot {
    std::atomic<uint64_t> claim{0};  // (generation << 32) | next unclaimed part of this thread's share
};

struct Workers::Impl {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv;
    alignas(64) std::atomic<uint64_t> generation{0};  // bumped once per run(); its own line: polled by every idle worker
    alignas(64) int parts = 0;                        // the job: written before the bump, read after it
    void (*fn)(void*, int) = nullptr;
    void* arg = nullptr;
    alignas(64) std::atomic<int> remaining{0};        // workers that have not finished the current run
    alignas(64) int nthreads = 1;
    std::atomic<bool> stop{false};
    Slot* slots = nullptr;

    // claims one part of thread v's share in run g; -1 when the share is exhausted
    int claim(int v, uint32_t g, int nparts) {
        const int lo = (int)((long long)nparts * v / nthreads), hi = (int)((long long)nparts * (v + 1) / nthreads);
        std::atomic<uint64_t>& w = slots[v].claim;
        uint64_t cur = w.load(std::memory_order_relaxed);
        for (;;) {
            const int idx = (uint32_t)(cur >> 32) == g ? (int)(uint32_t)cur : lo;
            if (idx >= hi) return -1;
            if (w.compare_exchange_weak(cur, ((uint64_t)g << 32) | (uint32_t)(idx + 1), std::memory_order_relaxed)) return idx;
        }
    }
    void share(int t, uint32_t g) {
        const int nparts = parts;
        void (*const f)(void*, int) = fn;
        void* const a = arg;
        for (int k = 0; k < nthreads; ++k) {  // own share first, then the others' leftovers
            const int v = t + k < nthreads ? t + k : t + k - nthreads;
            for (int p; (p = claim(v, g, nparts)) >= 0;) f(a, p);
        }
    }
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            // the calls of one labelling run arrive ~40 us apart: spin for a while before going to sleep
            int spins = 0;
            uint64_t g;
            while ((g = generation.load(std::memory_order_acquire)) == seen && !stop.load(std::memory_order_relaxed)) {
                if (++spins < 20000) {
                    cpu_relax();
                } else {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return generation.load(std::memory_order_acquire) != seen || stop.load(); });
                }
            }
            if (stop.load()) return;
            seen = g;
            share(t, (uint32_t)g);
            remaining.fetch_sub(1, std::memory_order_release);
        }
    }
};

// ---- placement -----------------------------------------------------------------------------------------------------
// Measured on a 2-socket EPYC 9575F (8 CCDs of 8 cores per socket), 16 threads packing 1080p maps:
//   unpinned 130 GB/s of int32 bytes, all on two CCDs 110 (a CCD's fabric link carries ~55 GB/s), one per CCD across
//   BOTH sockets 23 (remote memory + the fork-join lines bouncing between sockets), two per CCD on the caller's socket
//   167; maps on the OTHER node than the pool 78.  So: the pool lives on the NUMA node that holds the first map it is
//   given (move_pages query; else the node of the thread that creates it) and deals its workers round-robin over that
//   node's L3 domains.  Each worker is bound to the CPUs of one L3 domain, not to one CPU.  GSX_HOST_AFFINITY=0 leaves
//   the threads where the scheduler puts them.
static bool parse_cpulist(const char* path, cpu_set_t* out) {
    FILE* f = std::fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!ok) return false;
    CPU_ZERO(out);
    for (char* q = buf; *q && *q != '\n';) {  // "0-63,128-191"
        char* end;
        const long a = std::strtol(q, &end, 10);
        if (end == q) return false;
        long b = a;
        if (*end == '-') {
            q = end + 1;
            b = std::strtol(q, &end, 10);
            if (end == q) return false;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, out);
        q = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return true;
}

// NUMA node that holds the page of `addr` (move_pages(2) in query mode), -1 if the kernel will not say
int numa_node_of(const void* addr) {
#if defined(SYS_move_pages)
    if (!addr) return -1;
    void* page = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(addr) & ~(uintptr_t)4095);
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) == 0 && status >= 0) return status;
#endif
    return -1;
}

// the L3 domains of one NUMA node (`want_node`; < 0: the node of the calling thread), each intersected with the
// process's affinity mask
// GSX_HOST_AFFINITY: "0" = leave the workers where the scheduler puts them; "node" = bind them to the node's CPUs as one set;
// anything else (default) = deal them round-robin over the node's L3 domains.  Round 3 measured the three under bench.py's own
// run, interleaved three times on one box (profiles/r03/affinity_ab.txt): 8.26 / 8.45 / 8.13 ms per step on average, every
// policy best once - the L3 dealing takes 3-6 steps longer to settle (the scheduler spreads each pair of workers over its CCD's
// cores only gradually) but holds the lower steady rate when the socket's other tenants are busy.  Default unchanged.
static bool affinity_per_l3() {
    const char* e = std::getenv("GSX_HOST_AFFINITY");
    return !(e && e[0] == 'n');
}

static std::vector<cpu_set_t> l3_domains(int want_node) {
    std::vector<cpu_set_t> out;
    const char* e = std::getenv("GSX_HOST_AFFINITY");
    if (e && e[0] == '0') return out;
    const int cpu = sched_getcpu();
    cpu_set_t allowed, node;
    if (cpu < 0 || sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    bool found = false;
    if (want_node >= 0) {
        char path[96];
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", want_node);
        cpu_set_t both;
        found = parse_cpulist(path, &node);
        if (found) {
            CPU_AND(&both, &node, &allowed);
            found = CPU_COUNT(&both) > 0;  // the process may not run there at all: fall back to the caller's node
        }
    }
    for (int n = 0; n < 256 && !found; ++n) {
        char path[96];
        std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n);
        if (!parse_cpulist(path, &node)) break;
        found = CPU_ISSET(cpu, &node);
    }
    if (!found) return out;
    cpu_set_t todo;
    CPU_AND(&todo, &node, &allowed);
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &todo)) continue;
        char path[128];
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
        cpu_set_t l3, dom;
        if (!parse_cpulist(path, &l3)) return std::vector<cpu_set_t>();
        CPU_AND(&dom, &l3, &todo);
        if (CPU_COUNT(&dom) == 0) CPU_SET(c, &dom);
        out.push_back(dom);
        for (int k = 0; k < CPU_SETSIZE; ++k)
            if (CPU_ISSET(k, &dom)) CPU_CLR(k, &todo);
    }
    return out;
}

Workers::Workers(int threads, int numa_node) : impl_(new Impl), nthreads_(threads < 1 ? 1 : threads) {
    Impl* const impl = impl_;  // the threads capture the Impl, not `this`: a Workers that fails to construct leaves nothing dangling
    impl->nthreads = nthreads_;
    impl->slots = new (std::nothrow) Slot[(size_t)nthreads_];
    if (!impl->slots) {  // no room for the claim words: single-threaded packing
        nthreads_ = impl->nthreads = 1;
        impl->slots = new Slot[1];
    }
    std::vector<cpu_set_t> doms = nthreads_ > 1 ? l3_domains(numa_node) : std::vector<cpu_set_t>();
    if (!affinity_per_l3() && doms.size() > 1) {
        // policy "node": one set = all of the node's CPUs the process may use; the scheduler places the workers inside it
        cpu_set_t all;
        CPU_ZERO(&all);
        for (const cpu_set_t& d : doms) CPU_OR(&all, &all, &d);
        doms.assign(1, all);
    }
    // the caller itself sits in one of the domains (it is not moved); start dealing after it
    size_t first = 0;
    const int cpu = sched_getcpu();
    for (size_t d = 0; d < doms.size(); ++d)
        if (cpu >= 0 && CPU_ISSET(cpu, &doms[d])) first = d;
    for (int i = 1; i < nthreads_; ++i) {
        try {
            impl->threads.emplace_back([impl, i] { impl->loop(i); });
        } catch (...) {
            // thread or pid limit reached: stop and join the workers that did start and pack on the calling thread alone (their
            // shares were cut for nthreads_ threads; a smaller pool would need them cut again)
            {
                std::lock_guard<std::mutex> lk(impl->m);
                impl->stop.store(true);
            }
            impl->cv.notify_all();
            for (auto& t : impl->threads) t.join();
            impl->threads.clear();
            impl->stop.store(false);
            nthreads_ = impl->nthreads = 1;
            break;
        }
        if (!doms.empty()) {
            const cpu_set_t& dom = doms[(first + (size_t)i) % doms.size()];
            (void)pthread_setaffinity_np(impl->threads.back().native_handle(), sizeof dom, &dom);
        }
    }
}

Workers::~Workers() {
    {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->stop.store(true);
    }
    impl_->cv.notify_all();
    for (auto& t : impl_->threads) t.join();
    delete[] impl_->slots;
    delete impl_;
}

void Workers::run(int parts, void (*fn)(void*, int), void* arg) {
    if (parts <= 0) return;
    if (nthreads_ == 1 || parts == 1) {
        for (int p = 0; p < parts; ++p) fn(arg, p);
        return;
    }
    Impl& s = *impl_;
    // every worker has finished the previous run (run() waited for it), so nobody reads the job while it changes
    s.parts = parts;
    s.fn = fn;
    s.arg = arg;
    s.remaining.store(nthreads_ - 1, std::memory_order_relaxed);
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(s.m);  // pairs with the sleepers' predicate check
        g = s.generation.fetch_add(1, std::memory_order_release) + 1;
    }
    s.cv.notify_all();
    s.share(0, (uint32_t)g);
    while (s.remaining.load(std::memory_order_acquire) != 0) cpu_relax();
}

// CPUs' worth of time the cgroup of this process may use per scheduler period (cgroup v2 cpu.max, v1 cfs quota), 0 = unlimited
// or unknown.
static double cgroup_cpu_quota() {
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        double period = 0.0;
        const int got = std::fscanf(f, "%31s %lf", q, &period);
        std::fclose(f);
        if (got == 2 && period > 0.0 && std::strcmp(q, "max") != 0) {
            const double v = std::atof(q);
            return v > 0.0 ? v / period : 0.0;
        }
        return 0.0;
    }
    double quota = 0.0, period = 0.0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(f, "%lf", &quota) != 1) quota = 0.0;
        std::fclose(f);
    }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(f, "%lf", &period) != 1) period = 0.0;
        std::fclose(f);
    }
    return quota > 0.0 && period > 0.0 ? quota / period : 0.0;
}

int default_host_threads() {
    if (const char* e = std::getenv("GSX_HOST_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) return v > 256 ? 256 : v;
    }
    int cpus = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
    if (cpus < 1) cpus = (int)std::thread::hardware_concurrency();
    if (cpus < 1) cpus = 1;
    // A CPU-time quota of the process's cgroup counts like fewer CPUs: the workers spin between the maps of a run, and threads
    // beyond the quota only get the whole group frozen for the rest of a scheduler period.
    const double quota = cgroup_cpu_quota();
    if (quota > 0.0 && quota < (double)cpus) cpus = quota < 1.0 ? 1 : (int)quota;
    // One process per GPU: the ranks of a node share its CPUs (and the quota).  torch.distributed.run exports the number of
    // ranks on this node.
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int ranks = std::atoi(e);
        if (ranks > 1) cpus = cpus / ranks < 1 ? 1 : cpus / ranks;
    }
    return cpus > 16 ? 16 : cpus;
}

// ---- the narrowing copy --------------------------------------------------------------------------------------------
// bin = label + ADD as an unsigned number: label -1 -> 0, anything below -1 wraps far above `bins`.
// rows16<T, ADD> narrows `rows` groups of 16 pixels (source groups `pitch` elements apart) into 16-byte groups `opitch`
// bytes apart and returns non-zero if a bin >= bins was seen.  A scalar form for every dtype, AVX2 forms for the wide ones.
template <typename T, unsigned ADD>
static inline unsigned narrow_n(const T* __restrict__ src, int cnt, unsigned bins, uint8_t* __restrict__ out) {
    unsigned bad = 0;
    for (int k = 0; k < cnt; ++k) {
        const uint64_t b = (uint64_t)(int64_t)src[k] + ADD;  // T = uint8_t: plain zero-extension
        bad |= (unsigned)(b >= bins);
        out[k] = (uint8_t)b;
    }
    return bad;
}

template <typename T, unsigned ADD>
static unsigned rows16_scalar(const T* __restrict__ src, size_t pitch, int rows, unsigned bins, uint8_t* __restrict__ out, size_t opitch) {
    unsigned bad = 0;
    for (int r = 0; r < rows; ++r) bad |= narrow_n<T, ADD>(src + (size_t)r * pitch, 16, bins, out + (size_t)r * opitch);
    return bad;
}

#if defined(__x86_64__)
// How far ahead of the narrowing loops the map is prefetched, in bytes.  A band is 8 consecutive rows = one contiguous piece of
// the map and a thread's bands follow each other, so the stream never ends.  Measured on the GPU box (16 threads, 200 1080p
// int32 maps): 0 B 8.1 ms, 512 B 8.4-8.6, 2 KB 7.8, 4 KB 7.6, 6-8 KB 7.5-7.7, 16 KB 7.9, 32 KB 8.2, 64 KB 8.5: under load a
// core must keep ~100 lines in flight, more than the hardware prefetcher asks for on its own.
// The hint is NTA: the map is read exactly once, and lines that bypass L2/L3 leave the strips being assembled (and the pinned
// records) in the cache: 7.4 -> 6.85 ms.  (A negative value selects the T0 hint, for the A/B.)
static int g_prefetch_bytes = 8192;
static bool g_prefetch_t0 = false;
static bool g_prefetch_burst = true;
void set_host_prefetch(int bytes) { g_prefetch_t0 = bytes < 0; g_prefetch_bytes = bytes < 0 ? -bytes : bytes; }
void set_host_prefetch_burst(bool on) { g_prefetch_burst = on; }
static inline void prefetch_map(const void* p) {
#if defined(__x86_64__)
    if (g_prefetch_t0) _mm_prefetch(static_cast<const char*>(p), _MM_HINT_T0);
    else _mm_prefetch(static_cast<const char*>(p), _MM_HINT_NTA);
#else
    (void)p;
#endif
}
__attribute__((target("avx2"))) static unsigned rows16_i32_avx2(const int32_t* __restrict__ src, size_t pitch, int rows,
                                                                 unsigned bins, uint8_t* __restrict__ out, size_t opitch) {
    const __m256i one = _mm256_set1_epi32(1);
    __m256i mx = _mm256_setzero_si256();
    const int ahead = g_prefetch_bytes / 4;
    for (int r = 0; r < rows; ++r) {
        const int32_t* p = src + (size_t)r * pitch;
        // ask ahead of the hardware prefetcher (the groups of one call are consecutive along a pixel row)
        prefetch_map(p + ahead);
        __m256i a = _mm256_add_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), one);
        __m256i b = _mm256_add_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 8)), one);
        mx = _mm256_max_epu32(mx, _mm256_max_epu32(a, b));
        // packus saturates, but a value it would change (> 255, or "negative" = below label -1) is flagged bad anyway
        __m256i w = _mm256_permute4x64_epi64(_mm256_packus_epi32(a, b), 0xD8);  // u16: a0..7 | b0..7
        __m128i q = _mm_packus_epi16(_mm256_castsi256_si128(w), _mm256_extracti128_si256(w, 1));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + (size_t)r * opitch), q);
    }
    __m128i m = _mm_max_epu32(_mm256_castsi256_si128(mx), _mm256_extracti128_si256(mx, 1));
    m = _mm_max_epu32(m, _mm_shuffle_epi32(m, 0x4E));
    m = _mm_max_epu32(m, _mm_shuffle_epi32(m, 0xB1));
    return (unsigned)_mm_cvtsi128_si32(m) >= bins ? 1u : 0u;
}

// AVX-512 form (Zen 4/5, Skylake-X and later): one 64-byte load, one saturating down-convert per 16 pixels
__attribute__((target("avx512f"))) static unsigned rows16_i32_avx512(const int32_t* __restrict__ src, size_t pitch, int rows,
                                                                     unsigned bins, uint8_t* __restrict__ out, size_t opitch) {
    const __m512i one = _mm512_set1_epi32(1);
    __m512i mx = _mm512_setzero_si512();
    const int ahead = g_prefetch_bytes / 4;
    for (int r = 0; r < rows; ++r) {
        const int32_t* p = src + (size_t)r * pitch;
        prefetch_map(p + ahead);
        const __m512i a = _mm512_add_epi32(_mm512_loadu_si512(p), one);
        mx = _mm512_max_epu32(mx, a);
        // unsigned saturation: a value it would change (> 255, or "negative" = below label -1) is flagged bad anyway
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + (size_t)r * opitch), _mm512_cvtusepi32_epi8(a));
    }
    return _mm512_reduce_max_epu32(mx) >= bins ? 1u : 0u;
}

__attribute__((target("avx2"))) static unsigned rows16_i64_avx2(const int64_t* __restrict__ src, size_t pitch, int rows,
                                                                 unsigned bins, uint8_t* __restrict__ out, size_t opitch) {
    const __m256i one = _mm256_set1_epi64x(1);
    const __m256i idx = _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7);  // low dwords first, high dwords second
    __m256i mx = _mm256_setzero_si256(), hi_or = _mm256_setzero_si256();
    const int ahead = g_prefetch_bytes / 8;
    for (int r = 0; r < rows; ++r) {
        const int64_t* p = src + (size_t)r * pitch;
        prefetch_map(p + ahead);
        prefetch_map(p + ahead + 8);
        __m256i lo[2];
        for (int h = 0; h < 2; ++h) {
            __m256i a = _mm256_add_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 8 * h)), one);
            __m256i b = _mm256_add_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 8 * h + 4)), one);
            a = _mm256_permutevar8x32_epi32(a, idx);  // [lo a0..3 | hi a0..3]
            b = _mm256_permutevar8x32_epi32(b, idx);
            lo[h] = _mm256_permute2x128_si256(a, b, 0x20);                          // low dwords of 8 values
            hi_or = _mm256_or_si256(hi_or, _mm256_permute2x128_si256(a, b, 0x31));  // any high dword set -> out of range
            mx = _mm256_max_epu32(mx, lo[h]);
        }
        __m256i w = _mm256_permute4x64_epi64(_mm256_packus_epi32(lo[0], lo[1]), 0xD8);
        __m128i q = _mm_packus_epi16(_mm256_castsi256_si128(w), _mm256_extracti128_si256(w, 1));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + (size_t)r * opitch), q);
    }
    __m128i m = _mm_max_epu32(_mm256_castsi256_si128(mx), _mm256_extracti128_si256(mx, 1));
    m = _mm_max_epu32(m, _mm_shuffle_epi32(m, 0x4E));
    m = _mm_max_epu32(m, _mm_shuffle_epi32(m, 0xB1));
    return ((unsigned)_mm_cvtsi128_si32(m) >= bins || !_mm256_testz_si256(hi_or, hi_or)) ? 1u : 0u;
}
// the coarse byte of the four 4x4 cells of one strip: rows r0..r3 are 64 consecutive bytes (16 pixels each)
__attribute__((target("avx2"))) static uint32_t coarse_word_avx2(const uint8_t* __restrict__ q) {
    const __m128i r0 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q));
    const __m128i r1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 16));
    const __m128i r2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 32));
    const __m128i r3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 48));
    const __m128i first = _mm_shuffle_epi8(r0, _mm_setr_epi8(0, 0, 0, 0, 4, 4, 4, 4, 8, 8, 8, 8, 12, 12, 12, 12));
    __m128i eq = _mm_and_si128(_mm_and_si128(_mm_cmpeq_epi8(r0, first), _mm_cmpeq_epi8(r1, first)),
                               _mm_and_si128(_mm_cmpeq_epi8(r2, first), _mm_cmpeq_epi8(r3, first)));
    eq = _mm_cmpeq_epi32(eq, _mm_set1_epi32(-1));                       // per cell: all 16 pixels equal the first
    const __m128i val = _mm_blendv_epi8(_mm_set1_epi8((char)255), first, eq);
    return (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(val, _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1)));
}
// ... for one cell row of a band and its first 4 * n4 strips at once: everything inlined, the constants loaded once, and the four
// words of four consecutive strips stored as the 16 bytes they are in the coarse level (a coarse strip = 16 cell columns).  lines =
// the cell row's first row in strip 0, the strips' lines spitch apart; co = the cell row's 16 bytes of coarse strip 0.
__attribute__((target("avx2"))) static void coarse_row_avx2(const uint8_t* __restrict__ lines, size_t spitch, int n4,
                                                            uint8_t* __restrict__ co, size_t cstrip_bytes) {
    const __m128i pick = _mm_setr_epi8(0, 0, 0, 0, 4, 4, 4, 4, 8, 8, 8, 8, 12, 12, 12, 12);
    const __m128i gather = _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i ones = _mm_set1_epi32(-1), mixed = _mm_set1_epi8((char)255);
    for (int g = 0; g < n4; ++g) {
        __m128i w[4];
        for (int k = 0; k < 4; ++k) {
            const uint8_t* q = lines + (size_t)(4 * g + k) * spitch;
            const __m128i r0 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q));
            const __m128i r1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 16));
            const __m128i r2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 32));
            const __m128i r3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 48));
            const __m128i first = _mm_shuffle_epi8(r0, pick);
            __m128i eq = _mm_and_si128(_mm_and_si128(_mm_cmpeq_epi8(r0, first), _mm_cmpeq_epi8(r1, first)),
                                       _mm_and_si128(_mm_cmpeq_epi8(r2, first), _mm_cmpeq_epi8(r3, first)));
            eq = _mm_cmpeq_epi32(eq, ones);
            w[k] = _mm_shuffle_epi8(_mm_blendv_epi8(mixed, first, eq), gather);  // the strip's word in the low 4 bytes
        }
        const __m128i lo = _mm_unpacklo_epi32(w[0], w[1]), hi = _mm_unpacklo_epi32(w[2], w[3]);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(co + (size_t)g * cstrip_bytes), _mm_unpacklo_epi64(lo, hi));
    }
}
static const bool g_avx2 = __builtin_cpu_supports("avx2");
static const bool g_avx512 = __builtin_cpu_supports("avx512f") && !std::getenv("GSX_HOST_NO_AVX512");
#else
static const bool g_avx2 = false;
static uint32_t coarse_word_avx2(const uint8_t*) { return 0; }
static void coarse_row_avx2(const uint8_t*, size_t, int, uint8_t*, size_t) {}
#endif

template <typename T, unsigned ADD>
static inline unsigned rows16(const T* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    return rows16_scalar<T, ADD>(src, pitch, rows, bins, out, opitch);
}
#if defined(__x86_64__)
template <>
inline unsigned rows16<int32_t, 1u>(const int32_t* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    if (g_avx512) return rows16_i32_avx512(src, pitch, rows, bins, out, opitch);
    return g_avx2 ? rows16_i32_avx2(src, pitch, rows, bins, out, opitch) : rows16_scalar<int32_t, 1u>(src, pitch, rows, bins, out, opitch);
}
// uint8 sources (8-bit class images: ADD = 1; already packed bins: ADD = 0): one 16-byte load, compare, add and store per group -
// SSE2, the x86-64 baseline.  bin = v + ADD >= bins  <=>  v >= bins - ADD (never, if that exceeds 255).  (The scalar form made the
// hand-over of 200 uint8 1080p maps SLOWER than that of the int32 maps they are a quarter of: 11.0 against 9.0 ms per run.)
template <unsigned ADD>
static inline unsigned rows16_u8_sse2(const uint8_t* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    if (bins < ADD) return rows16_scalar<uint8_t, ADD>(src, pitch, rows, bins, out, opitch);
    const unsigned lim_u = bins - ADD;
    const __m128i lim = _mm_set1_epi8((char)(lim_u > 255u ? 255u : lim_u));
    const __m128i add = _mm_set1_epi8((char)ADD);
    __m128i acc = _mm_setzero_si128();
    for (int r = 0; r < rows; ++r) {
        if ((r & 3) == 0) prefetch_map(src + (size_t)r * pitch + g_prefetch_bytes);  // one line = four groups
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + (size_t)r * pitch));
        acc = _mm_or_si128(acc, _mm_cmpeq_epi8(_mm_max_epu8(v, lim), v));  // v >= lim (unsigned)
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + (size_t)r * opitch), _mm_add_epi8(v, add));
    }
    return lim_u > 255u ? 0u : (unsigned)(_mm_movemask_epi8(acc) != 0);
}
template <>
inline unsigned rows16<uint8_t, 1u>(const uint8_t* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    return rows16_u8_sse2<1u>(src, pitch, rows, bins, out, opitch);
}
template <>
inline unsigned rows16<uint8_t, 0u>(const uint8_t* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    return rows16_u8_sse2<0u>(src, pitch, rows, bins, out, opitch);
}
template <>
inline unsigned rows16<int64_t, 1u>(const int64_t* src, size_t pitch, int rows, unsigned bins, uint8_t* out, size_t opitch) {
    return g_avx2 ? rows16_i64_avx2(src, pitch, rows, bins, out, opitch) : rows16_scalar<int64_t, 1u>(src, pitch, rows, bins, out, opitch);
}
#endif

struct PackJob {
    const void* seg;
    MapLayout L;
    unsigned bins;
    uint8_t* dst;     // full-resolution level (strips)
    uint8_t* coarse;  // coarse level (plain form: dst + L.coarse_off)
    // compact form only
    uint32_t* table = nullptr;
    uint8_t* stream = nullptr;
    // written by the workers: on lines of their own (every band reads the fields above; a counter next to them would pull
    // that line out of sixteen caches 135 times per map)
    alignas(64) std::atomic<unsigned> bad{0};
    alignas(64) std::atomic<uint32_t> blocks{0};
    char pad_[60] = {};
};

// one band of 8 pixel rows = one 128-B line of every strip (and two rows of coarse cells).  The source is read ROW BY
// ROW, start to end (a band is 8 consecutive rows = one contiguous piece of the map): plain streaming, which a core
// reads at twice the rate of eight row streams walked strip by strip in lock step.  The 16-byte pieces go to the strips'
// lines (strip_bytes apart), which stay in L1/L2 for the 8 rows of the band; the coarse bytes are made from them at the end.
template <typename T, unsigned ADD>
static void pack_band_tiled(PackJob* j, int part) {
    const MapLayout L = j->L;  // by value: the u8 stores below may alias anything reached through a pointer
    const unsigned bins = j->bins;
    uint8_t* const dst = j->dst;
    const T* seg = static_cast<const T*>(j->seg);
    const int band = part;  // 8 pixel rows = two rows of cells
    constexpr int c0 = 0, c1 = 2;
    const int r0 = band * 8, r1 = r0 + 8;                               // pixel rows of this band ...
    const int y0 = r0 < L.h ? r0 : L.h, y1 = r1 < L.h ? r1 : L.h;       // ... that exist
    const int strips = (L.w + 15) / 16, full = L.w / 16;
    unsigned bad = 0;
    {
        // A band that does not continue this thread's stream (the first of its share, a band taken over from another thread)
        // starts cold, a whole prefetch distance behind: ask for that distance at once (hand-over of 200 maps 6.87 -> 6.72 ms).
        static thread_local const void* continues_at = nullptr;
        const char* first = reinterpret_cast<const char*>(seg + (size_t)y0 * L.w);
        const char* end = reinterpret_cast<const char*>(seg + (size_t)y1 * L.w);
        if (first != continues_at && g_prefetch_burst) {
            const char* stop = first + g_prefetch_bytes < end ? first + g_prefetch_bytes : end;
            for (const char* q = first; q < stop; q += 64) prefetch_map(q);
        }
        continues_at = end;
    }
    // Compact form: nothing reads the strips behind this function (the coarse bytes and the mixed cells' blocks are all that
    // leaves), so the band's lines need not live in a map-sized scratch whose every line is a fresh miss - they go to a buffer
    // of this thread that the next band reuses: `strips` lines of 128 bytes, in the first-level cache for good.
    const bool local = j->stream != nullptr;
    static thread_local std::vector<uint8_t> lines_tl;
    if (local && lines_tl.size() < (size_t)strips * 128 + 64) lines_tl.resize((size_t)strips * 128 + 64);
    uint8_t* const lbase = local ? lines_tl.data() + ((64 - (reinterpret_cast<uintptr_t>(lines_tl.data()) & 63)) & 63) : nullptr;
    const size_t spitch = local ? (size_t)128 : (size_t)L.strip_bytes;             // from one strip's line to the next strip's
    uint8_t* const band0 = local ? lbase : dst + (size_t)band * 128;               // the band's line of strip 0
    for (int y = y0; y < y1; ++y) {
        const T* row = seg + (size_t)y * L.w;
        uint8_t* o = band0 + (size_t)(y - r0) * 16;  // row y of strip 0
        if (full) bad |= rows16<T, ADD>(row, 16, full, bins, o, spitch);
        if (full < strips) {  // ragged last strip: columns past the map hold bin 0
            uint8_t* e = o + (size_t)full * spitch;
            std::memset(e, 0, 16);
            bad |= narrow_n<T, ADD>(row + (size_t)full * 16, L.w - full * 16, bins, e);
        }
    }
    // a band that lies inside the map: its two cell rows' coarse bytes four strips at a time (coarse_row_avx2); the strips that are
    // left (and every strip of a ragged band, or without AVX2) take the general loop below
    int s_fast = 0;
    if (g_avx2 && L.cstrip_bytes && y1 == r1 && band * 2 + 2 <= L.ch && band * 8 + 8 <= L.h) {
        s_fast = (full / 4) * 4;
        for (int cyl = c0; cyl < c1 && s_fast; ++cyl)
            coarse_row_avx2(band0 + (size_t)cyl * 64, spitch, s_fast / 4, j->coarse + (size_t)(band * 2 + cyl) * 16, (size_t)L.cstrip_bytes);
    }
    for (int s = s_fast; s < strips; ++s) {
        const int x0 = s * 16;
        const int cnt = L.w - x0 >= 16 ? 16 : L.w - x0;
        uint8_t* line = band0 + (size_t)s * spitch;
        const int z0 = y1 > r0 ? y1 : r0;  // this band's rows past the map: defined bytes
        if (z0 < r1) std::memset(line + (size_t)(z0 - band * 8) * 16, 0, (size_t)(r1 - z0) * 16);
        if (L.cstrip_bytes) {
            // coarse cells of this strip: 4 per cell row; a cell holds the bin its 16 pixels share, else 255
            // (cells that stick out of the map are "mixed": the vote then reads the exact pixel)
            for (int cyl = c0; cyl < c1; ++cyl) {
                const int cy = band * 2 + cyl;
                if (cy >= L.ch) break;
                uint32_t word = 0xffffffffu;
                if (cy * 4 + 4 <= L.h) {
                    if (g_avx2 && cnt == 16) {
                        word = coarse_word_avx2(line + cyl * 64);
                    } else {
                        uint32_t r[4][4];
                        std::memcpy(r, line + cyl * 64, 64);
                        word = 0;
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t same = (r[0][c] & 0xffu) * 0x01010101u;
                            const bool uniform = x0 + 4 * c + 4 <= L.w && r[0][c] == same && r[1][c] == same && r[2][c] == same && r[3][c] == same;
                            word |= (uniform ? (r[0][c] & 0xffu) : 255u) << (8 * c);
                        }
                    }
                }
                const int cx0 = s * 4;  // first cell of the strip
                const int ncell = L.cw - cx0 >= 4 ? 4 : L.cw - cx0;
                uint8_t* co = j->coarse + (size_t)(cx0 >> 4) * L.cstrip_bytes + (cx0 & 15) + (size_t)cy * 16;
                if (ncell == 4) std::memcpy(co, &word, 4);
                else std::memcpy(co, &word, (size_t)ncell);
            }
        }
    }
    if (bad) j->bad.fetch_or(1u, std::memory_order_relaxed);
    if (j->stream) {
        // compact form: the mixed cells (coarse byte 255) of this band's two cell rows go to the stream as 16-byte blocks, cell
        // row by cell row and inside a row by cell column; uniform cells are their coarse byte.  The strips' lines were
        // written a moment ago: they are in this core's cache.  Cost ~ the number of mixed cells: a row of 16 cells is one
        // compare + movemask.  table[2 * band + c] = where cell row c's blocks start.
        constexpr int kMaxCs = 1024;  // coarse strips of a 65535-pixel row
        uint16_t masks[2][kMaxCs];
        const int ncs = (L.cw + 15) >> 4;
        uint32_t n[2] = {0, 0};
        for (int cyl = c0; cyl < c1; ++cyl) {
            const int cy = band * 2 + cyl;
            if (cy >= L.ch) break;
            if (L.cw & 15)  // the cells past the map's last cell column: defined bytes (the level goes into the pool as it is)
                std::memset(j->coarse + (size_t)(L.cw >> 4) * L.cstrip_bytes + (size_t)cy * 16 + (L.cw & 15), 0, (size_t)(16 - (L.cw & 15)));
            for (int cs = 0; cs < ncs; ++cs) {
                const uint8_t* co = j->coarse + (size_t)cs * L.cstrip_bytes + (size_t)cy * 16;
                unsigned m;
#if defined(__x86_64__)
                m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(co)), _mm_set1_epi8((char)255)));
#else
                m = 0;
                for (int k = 0; k < 16; ++k) m |= (unsigned)(co[k] == 255) << k;
#endif
                masks[cyl][cs] = (uint16_t)m;  // padding cells are 0, never 255
                n[cyl] += (unsigned)__builtin_popcount(m);
            }
        }
        const uint32_t total = n[0] + n[1];
        uint32_t first = total ? j->blocks.fetch_add(total, std::memory_order_relaxed) : 0u;
        for (int cyl = c0; cyl < c1; ++cyl) {
            j->table[2 * band + cyl] = first;
            uint8_t* out = j->stream + (size_t)first * 16;
            first += n[cyl];
            if (!n[cyl]) continue;
            const uint8_t* rows = band0 + (size_t)(cyl * 4) * 16;  // row 4*cyl of the band in strip 0
            for (int cs = 0; cs < ncs; ++cs) {
                for (unsigned m = masks[cyl][cs]; m; m &= m - 1) {
                    const int cx = cs * 16 + __builtin_ctz(m);
                    const uint8_t* q = rows + (size_t)(cx >> 2) * spitch + (cx & 3) * 4;  // rows 16 bytes apart
                    uint32_t r[4];
                    std::memcpy(&r[0], q, 4), std::memcpy(&r[1], q + 16, 4), std::memcpy(&r[2], q + 32, 4), std::memcpy(&r[3], q + 48, 4);
                    std::memcpy(out, r, 16);
                    out += 16;
                }
            }
        }
    }
}

template <typename T, unsigned ADD>
static void pack_band_rows(PackJob* j, int band) {  // row-major u8 map ("seg_tiled" = 0)
    const MapLayout L = j->L;
    const T* seg = static_cast<const T*>(j->seg);
    const int y0 = band * 8, y1 = y0 + 8 < L.h ? y0 + 8 : L.h;
    const size_t i0 = (size_t)y0 * L.w, i1 = (size_t)y1 * L.w;
    const size_t full = (i1 - i0) / 16;
    unsigned bad = full ? rows16<T, ADD>(seg + i0, 16, (int)full, j->bins, j->dst + i0, 16) : 0u;
    const size_t i = i0 + full * 16;
    if (i < i1) bad |= narrow_n<T, ADD>(seg + i, (int)(i1 - i), j->bins, j->dst + i);
    if (bad) j->bad.fetch_or(1u, std::memory_order_relaxed);
}

template <typename T, unsigned ADD>
static void pack_part(void* arg, int part) {
    PackJob* j = static_cast<PackJob*>(arg);
    if (j->L.strip_bytes) pack_band_tiled<T, ADD>(j, part);
    else pack_band_rows<T, ADD>(j, part);
}

// A map is cut into bands of 8 pixel rows, one part of the fork-join each, dealt in contiguous shares: every thread streams
// through one piece of the map.  Measured and removed again in round 2, all slower on the GPU box: bands cut into column
// segments (270 parts per 1080p map: +12 %), several bands per part (+0..12 %), and the bands that do not divide evenly
// (135 = 16 x 8 + 7) handed out as single cell rows so that no thread idles at the end (+6.5 %: a part that does not continue a
// thread's stream starts cold, 8 KB of prefetch distance behind).
int host_pack_map(Workers* pool, const void* seg, int seg_dtype, const MapLayout& L, int bins, uint8_t* dst) {
    PackJob j;
    j.seg = seg;
    j.L = L;
    j.bins = (unsigned)bins;
    j.dst = dst;
    j.coarse = dst + L.coarse_off;
    void (*fn)(void*, int) = seg_dtype == 0   ? pack_part<int32_t, 1u>
                             : seg_dtype == 1 ? pack_part<int64_t, 1u>
                             : seg_dtype == 2 ? pack_part<uint8_t, 0u>
                                              : pack_part<uint8_t, 1u>;
    const int bands = (L.h + 7) / 8;
    // the coarse level's padding (cell rows / columns past the map) is never read; zero it so that a packed map is a
    // function of the map alone (the all-gather of protocol v4 ships these bytes)
    if (L.cstrip_bytes) std::memset(dst + L.coarse_off, 0, L.map_bytes - L.coarse_off);
    if (L.coarse_off > L.fine_bytes && L.cstrip_bytes) std::memset(dst + L.fine_bytes, 0, L.coarse_off - L.fine_bytes);
    if (pool) pool->run(bands, fn, &j);
    else
        for (int b = 0; b < bands; ++b) fn(&j, b);
    if (!L.strip_bytes) std::memset(dst + (size_t)L.w * L.h, 0, 4);  // the row-major form's 4 bytes of slack
    return j.bad.load() ? 1 : 0;
}

CompactLayout compact_layout(const MapLayout& L) {
    CompactLayout C;
    C.bands = (L.h + 7) / 8;
    C.table_bytes = ((size_t)C.bands * 8 + 255) / 256 * 256;  // one entry per cell row = two per band
    C.coarse_bytes = (L.map_bytes - L.coarse_off + 255) / 256 * 256;
    C.stream_off = C.table_bytes + C.coarse_bytes;
    C.capacity = C.stream_off + (size_t)L.cw * (size_t)L.ch * 16;
    return C;
}

int host_pack_map_compact(Workers* pool, const void* seg, int seg_dtype, const MapLayout& L, int bins, uint8_t* scratch, uint8_t* rec,
                          size_t* blocks) {
    const CompactLayout C = compact_layout(L);
    PackJob j;
    j.seg = seg;
    j.L = L;
    j.bins = (unsigned)bins;
    j.dst = scratch;
    j.coarse = rec + C.table_bytes;
    j.table = reinterpret_cast<uint32_t*>(rec);
    j.stream = rec + C.stream_off;
    void (*fn)(void*, int) = seg_dtype == 0   ? pack_part<int32_t, 1u>
                             : seg_dtype == 1 ? pack_part<int64_t, 1u>
                             : seg_dtype == 2 ? pack_part<uint8_t, 0u>
                                              : pack_part<uint8_t, 1u>;
    std::memset(rec + (size_t)C.bands * 8, 0, C.table_bytes - (size_t)C.bands * 8);
    // cell rows past the map (a coarse strip has room for a multiple of 8): defined bytes, like the cell columns past the
    // map that the bands clear - the level goes into the pool as it is
    const size_t row_room = (size_t)L.cstrip_bytes / 16;
    for (int cs = 0; cs * 16 < L.cw && row_room > (size_t)L.ch; ++cs)
        std::memset(j.coarse + (size_t)cs * L.cstrip_bytes + (size_t)L.ch * 16, 0, (row_room - (size_t)L.ch) * 16);
    if (pool) pool->run(C.bands, fn, &j);
    else
        for (int b = 0; b < C.bands; ++b) fn(&j, b);
    *blocks = j.blocks.load();
    return j.bad.load() ? 1 : 0;
}

struct CopyJob {
    char* dst;
    const char* src;
    size_t bytes, chunk;
};
static void copy_part(void* arg, int part) {
    CopyJob* j = static_cast<CopyJob*>(arg);
    const size_t lo = (size_t)part * j->chunk;
    const size_t hi = lo + j->chunk < j->bytes ? lo + j->chunk : j->bytes;
    if (lo < hi) std::memcpy(j->dst + lo, j->src + lo, hi - lo);
}
void host_copy(Workers* pool, void* dst, const void* src, size_t bytes) {
    if (!pool || bytes < ((size_t)1 << 20)) {
        std::memcpy(dst, src, bytes);
        return;
    }
    CopyJob j{static_cast<char*>(dst), static_cast<const char*>(src), bytes, (size_t)256 << 10};
    pool->run((int)((bytes + j.chunk - 1) / j.chunk), copy_part, &j);
}

// ---- bins (u8) -> labels (int32) -------------------------------------------------------------------------------------
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void widen_avx2(int32_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
    const __m256i one = _mm256_set1_epi32(1);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m128i b = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(src + i));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_sub_epi32(_mm256_cvtepu8_epi32(b), one));
    }
    for (; i < n; ++i) dst[i] = (int32_t)src[i] - 1;
}
#endif
static void widen_scalar(int32_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (int32_t)src[i] - 1;
}
struct WidenJob {
    int32_t* dst;
    const uint8_t* src;
    size_t n, chunk;
};
static void widen_part(void* arg, int part) {
    WidenJob* j = static_cast<WidenJob*>(arg);
    const size_t lo = (size_t)part * j->chunk;
    const size_t hi = lo + j->chunk < j->n ? lo + j->chunk : j->n;
    if (lo >= hi) return;
#if defined(__x86_64__)
    if (g_avx2) return widen_avx2(j->dst + lo, j->src + lo, hi - lo);
#endif
    widen_scalar(j->dst + lo, j->src + lo, hi - lo);
}
void host_widen_labels(Workers* pool, int32_t* dst, const uint8_t* src, size_t n) {
    WidenJob j{dst, src, n, (size_t)64 << 10};
    const int parts = (int)((n + j.chunk - 1) / j.chunk);
    if (!pool || parts <= 1) {
        for (int p = 0; p < parts; ++p) widen_part(&j, p);
        return;
    }
    pool->run(parts, widen_part, &j);
}

struct StressJob {
    std::atomic<int>* hits;
};
static void stress_part(void* arg, int part) { static_cast<StressJob*>(arg)->hits[part].fetch_add(1, std::memory_order_relaxed); }

long long workers_stress(int threads, int runs, int max_parts) {
    if (max_parts < 1) max_parts = 1;
    Workers pool(threads);
    std::vector<std::atomic<int>> hits((size_t)max_parts);
    long long wrong = 0;
    uint64_t state = 0x9e3779b97f4a7c15ull;
    for (int r = 0; r < runs; ++r) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const int parts = 1 + (int)((state >> 33) % (uint64_t)max_parts);
        for (int p = 0; p < max_parts; ++p) hits[(size_t)p].store(0, std::memory_order_relaxed);
        StressJob j{hits.data()};
        pool.run(parts, stress_part, &j);
        for (int p = 0; p < max_parts; ++p) wrong += hits[(size_t)p].load(std::memory_order_relaxed) != (p < parts ? 1 : 0);
    }
    return wrong;
}

}  // namespace gsx
