// api.hip — the C ABI (include/arroy_hip.h): dataset staging, search-side and split-side entry points.
// The forest build lives in forest.hip.  Host code only orchestrates: every arithmetic result comes from
// a HIP kernel; there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <immintrin.h>
#include <new>
#include <thread>
#include <cctype>
#include <cstdio>
#include <mutex>
#include <vector>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "common.h"

namespace ah {

static thread_local std::string g_error;
static thread_local ah_error_detail g_detail = {AH_OK, 0, 0, 0};

static thread_local int g_guard_depth = 0;  // > 0: this thread is inside an extern "C" entry point (guarded(), common.h)

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    const int depth = g_guard_depth;
    g_guard_depth = 0;  // (the message itself is never the allocation that AH_FAIL_ALLOC_AFTER fails)
    try {
        g_error = buf;
    } catch (...) {  // not even the message fits in memory: the status code still says what happened
        g_error.clear();
    }
    g_guard_depth = depth;
    g_detail = ah_error_detail{AH_ERR_DEVICE, 0, 0, 0};
}
NoFailScope::NoFailScope() : saved(g_guard_depth) { g_guard_depth = 0; }
NoFailScope::~NoFailScope() { g_guard_depth = saved; }
int guard_enter() { return g_guard_depth++; }
void guard_leave() { g_guard_depth--; }
int guard_failed(const char *what, int kind, const char *text) noexcept {
    const int status = kind == 0 ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE;
    if (kind == 0) set_error("%s: host allocation failed", what);
    else if (kind == 1) set_error("%s: unexpected C++ exception: %s", what, text ? text : "");
    else set_error("%s: unexpected C++ exception", what);
    set_error_status(status);
    return status;
}
bool guard_active() { return g_guard_depth > 0; }
void set_error_status(int status) { g_detail.status = status; }
void set_error_detail(uint32_t item, uint64_t expected, uint64_t received) {
    g_detail.item = item;
    g_detail.expected = expected;
    g_detail.received = received;
}
const char *last_error() { return g_error.c_str(); }

// ---- tunables (common.h): environment at load time, ah_tuning_set at run time ---------------------------------------
namespace {
struct TunableSlot {
    const char *name;
    long long def;
    std::atomic<long long> value;
};
TunableSlot *tunable_table() {
    static TunableSlot table[TUN_COUNT] = {
#define AH_X(id, name, def) {name, (long long)(def), {(long long)(def)}},
        AH_TUNABLES(AH_X)
#undef AH_X
    };
    static const bool loaded = [] {
        for (TunableSlot &t : table) {
            const char *e = getenv(t.name);
            if (!e) continue;
            char *end = nullptr;
            const long long v = strtoll(e, &end, 0);
            t.value.store(end == e ? 1 : v, std::memory_order_relaxed);  // set but not a number ("AH_DEBUG=yes"): on
        }
        return true;
    }();
    (void)loaded;
    return table;
}
}  // namespace
long long tun(int id) { return tunable_table()[id].value.load(std::memory_order_relaxed); }

int Context::ensure_device(size_t bytes) {
    clean_status = nullptr;  // (whoever carves the scratch next may write anywhere in it)
    if (bytes <= d_cap) return AH_OK;
    size_t cap = std::max(bytes + bytes / 4, d_cap * 2);  // headroom: similar-sized submissions must not regrow
    cap = (cap + 4095) & ~(size_t)4095;
    if (d_scratch) AH_HIP(dev_free(d_scratch));
    d_scratch = nullptr;
    d_cap = 0;
    AH_HIP(dev_malloc(&d_scratch, cap));
    d_cap = cap;
    return AH_OK;
}
hipError_t create_copy_stream(hipStream_t *out) {
    int least = 0, greatest = 0;
    if (tun(TUN_READBACK_PRIORITY) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least)
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// ---- NUMA placement --------------------------------------------------------------------------------------------------
// A two-socket host reaches the GPU through one of its sockets.  The 9.4 GB a 10M x 100-tree build hands back travel device ->
// pinned bounce buffer -> the forest's blobs on a few copy threads; with the blobs first-touched on the far socket (whichever
// cores the page-commit threads happened to get, or a recycled blob another build committed) those copies cross the socket
// link: a build whose blobs were first touched from the far socket ran 1.273-1.285 s with 17 ms after its last launch, the same
// build with them on the device's node 1.265-1.272 s with 13 ms, whichever socket the caller ran on (scripts/exp_numa.py,
// profiles/r06_experiments.txt).  So the blobs prefer the device's node, and the threads that fill them run there.
namespace {
int read_int_file(const char *path, int fallback) {
    FILE *f = fopen(path, "r");
    if (!f) return fallback;
    int v = fallback;
    if (fscanf(f, "%d", &v) != 1) v = fallback;
    fclose(f);
    return v;
}
bool parse_cpulist(const char *path, cpu_set_t *out) {  // "0-63,128-191"
    FILE *f = fopen(path, "r");
    if (!f) return false;
    CPU_ZERO(out);
    int a = 0, b = 0;
    bool any = false;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        int c = fgetc(f);
        if (c == '-') {
            if (fscanf(f, "%d", &b) != 1) break;
            c = fgetc(f);
        }
        for (int i = a; i <= b && i < CPU_SETSIZE; i++) {
            CPU_SET(i, out);
            any = true;
        }
        if (c != ',') break;
    }
    fclose(f);
    return any;
}
std::mutex g_numa_mu;
std::vector<std::pair<int, int>> g_numa_of_device;  // (device, node)
}  // namespace
int numa_node_of_device(int device) {
    if (tun(TUN_NUMA) == 0) return -1;
    {
        std::lock_guard<std::mutex> lk(g_numa_mu);
        for (const auto &e : g_numa_of_device)
            if (e.first == device) return e.second;
    }
    int node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) == hipSuccess && bus[0]) {
        for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
        char path[160];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
        node = read_int_file(path, -1);
    } else {
        (void)hipGetLastError();
    }
    if (node >= 0) {  // (one node online: nothing to place)
        FILE *f = fopen("/sys/devices/system/node/node1/cpulist", "r");
        if (!f) node = -1;
        else fclose(f);
    }
    std::lock_guard<std::mutex> lk(g_numa_mu);
    g_numa_of_device.push_back({device, node});
    return node;
}
bool numa_bind_thread_to_node(int node) {
    if (node < 0) return false;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    cpu_set_t of_node, allowed, both;
    if (!parse_cpulist(path, &of_node)) return false;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    CPU_AND(&both, &of_node, &allowed);
    // (a container with few or none of its CPUs on that node: leave the thread where it may run — eight copy threads squeezed onto
    // two cores would cost more than the socket link)
    if (CPU_COUNT(&both) < 8) return false;
    return sched_setaffinity(0, sizeof both, &both) == 0;
}
void numa_prefer_node(void *p, size_t bytes, int node) {
    if (node < 0 || node >= 1024 || !p || !bytes) return;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    (void)syscall(SYS_mbind, p, bytes, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(8 * sizeof mask), 0u);
}

// One pinned block obtained ahead of time (ah_dataset_reserve_build's helper thread, while the records are staged): pinning the
// ~0.4 GB a 10M x 100-tree build wants for its node tables and read-back buffer takes the driver ~120 ms, which a cold first
// build used to spend between its entry and its first launch.  The next context of that device that has to grow takes it.
namespace {
struct PinnedSpare {
    std::mutex mu;
    void *p = nullptr;
    size_t cap = 0;
    int device = -1;
};
PinnedSpare &pinned_spare() {
    static PinnedSpare *s = new PinnedSpare();  // never destroyed (contexts may outlive static destruction order)
    return *s;
}
}  // namespace
void pinned_spare_fill(int device, size_t bytes) {
    PinnedSpare &sp = pinned_spare();
    const size_t cap = (bytes + bytes / 4 + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> lk(sp.mu);
        if (sp.p && sp.device == device && sp.cap >= cap) return;
    }
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();  // no pinned memory to be had now: the build asks for itself
        return;
    }
    void *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(sp.mu);
        old = sp.p;
        sp.p = p;
        sp.cap = cap;
        sp.device = device;
    }
    if (old) (void)hipHostFree(old);
}
size_t pinned_spare_trim() {
    PinnedSpare &sp = pinned_spare();
    void *p = nullptr;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(sp.mu);
        p = sp.p;
        cap = sp.cap;
        sp.p = nullptr;
        sp.cap = 0;
        sp.device = -1;
    }
    if (p) (void)hipHostFree(p);
    return p ? cap : 0;
}
int Context::ensure_pinned(size_t bytes) {
    if (bytes <= h_cap) return AH_OK;
    size_t cap = std::max(bytes + bytes / 4, h_cap * 2);
    cap = (cap + 4095) & ~(size_t)4095;
    AH_REQUIRE(!fail_alloc_tick(), AH_ERR_OUT_OF_MEMORY, "pinned host allocation of %zu bytes failed (AH_FAIL_ALLOC_AFTER)", cap);
    void *spare = nullptr;
    size_t spare_cap = 0;
    {
        int dev = -1;
        (void)hipGetDevice(&dev);
        PinnedSpare &sp = pinned_spare();
        std::lock_guard<std::mutex> lk(sp.mu);
        if (sp.p && sp.device == dev && sp.cap >= bytes) {
            spare = sp.p;
            spare_cap = sp.cap;
            sp.p = nullptr;
            sp.cap = 0;
            sp.device = -1;
        }
    }
    if (h_pinned) AH_HIP(hipHostFree(h_pinned));
    h_pinned = nullptr;
    h_cap = 0;
    if (spare) {
        h_pinned = spare;
        h_cap = spare_cap;
        return AH_OK;
    }
    AH_HIP(hipHostMalloc(&h_pinned, cap, hipHostMallocDefault));
    h_cap = cap;
    return AH_OK;
}
int Context::ensure_filter(size_t bytes) {
    if (bytes <= d_filter_cap) return AH_OK;
    size_t cap = std::max(bytes + bytes / 4, d_filter_cap * 2);
    cap = (cap + 4095) & ~(size_t)4095;
    if (d_filter) AH_HIP(dev_free(d_filter));
    d_filter = nullptr;
    d_filter_cap = 0;
    AH_HIP(dev_malloc(&d_filter, cap));
    d_filter_cap = cap;
    return AH_OK;
}
int Context::ensure_multi(size_t bytes) {
    if (d_multi) return AH_OK;
    AH_HIP(dev_malloc(&d_multi, bytes));
    const hipError_t e = hipMemsetAsync(d_multi, 0, bytes, stream);  // (stream order: before the first kernel that reads it)
    if (e != hipSuccess) {
        (void)dev_free(d_multi);
        d_multi = nullptr;
        AH_HIP(e);
    }
    return AH_OK;
}
// ---- caching device allocator (common.h) ------------------------------------------------------------------------------
namespace {
struct DevBlock {
    void *p;
    size_t bytes;
    int device;
};
// (heap-allocated and never destroyed: handles may be destroyed by finalizers that run after this library's static destructors)
std::mutex &g_dev_mu = *new std::mutex();
std::vector<DevBlock> &g_dev_idle = *new std::vector<DevBlock>();  // blocks nobody uses, oldest first
std::vector<DevBlock> &g_dev_live = *new std::vector<DevBlock>();  // blocks handed out (a few hundred at most: a linear scan is fine)
size_t g_dev_idle_bytes = 0;
size_t g_dev_pending = 0;  // records promised to dev_malloc calls that have not pushed theirs yet (under g_dev_mu)
inline size_t dev_round(size_t bytes) {  // whole 2 MiB for the big blocks (what the driver maps anyway), 4 KiB below
    const size_t g = bytes >= (1u << 20) ? (2u << 20) : 4096u;
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}
}  // namespace

// AH_FAIL_ALLOC_AFTER=n (test aid): the n-th allocation counted from the moment the tunable was set fails — device blocks
// (dev_malloc), pinned host memory, the forests' host blobs and every `operator new` of this library (below).
bool fail_alloc_tick() {
    if (tun(TUN_FAIL_ALLOC_AFTER) <= 0) return false;
    TunableSlot &t = tunable_table()[TUN_FAIL_ALLOC_AFTER];
    long long v = t.value.load(std::memory_order_relaxed);
    while (v > 0)
        if (t.value.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) return v == 1;
    return false;
}

namespace {
std::atomic<int> g_live_datasets[64];
std::atomic<int> g_live_total{0};
}  // namespace
void dataset_born(int device) {
    g_live_datasets[device & 63].fetch_add(1, std::memory_order_relaxed);
    g_live_total.fetch_add(1, std::memory_order_relaxed);
}
void dataset_gone(int device) {
    const bool last_here = g_live_datasets[device & 63].fetch_sub(1, std::memory_order_acq_rel) == 1;
    const bool last = g_live_total.fetch_sub(1, std::memory_order_acq_rel) == 1;
    if (tun(TUN_CACHE_KEEP_IDLE) != 0) return;
    if (last_here) (void)dev_cache_trim(device);
    if (last) (void)host_cache_trim();
}

size_t dev_cache_live_bytes(int device) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    size_t n = 0;
    for (const DevBlock &b : g_dev_live)
        if (device < 0 || b.device == device) n += b.bytes;
    return n;
}

size_t dev_cache_trim(int device) {
    NoFailScope no_fail;
    std::vector<DevBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        for (size_t i = 0; i < g_dev_idle.size();) {
            if (device < 0 || g_dev_idle[i].device == device) {
                drop.push_back(g_dev_idle[i]);
                g_dev_idle_bytes -= g_dev_idle[i].bytes;
                g_dev_idle.erase(g_dev_idle.begin() + (ptrdiff_t)i);
            } else {
                i++;
            }
        }
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    size_t bytes = 0;
    for (const DevBlock &b : drop) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
        bytes += b.bytes;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return bytes;
}

size_t dev_cache_idle_bytes(int device) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    size_t n = 0;
    for (const DevBlock &b : g_dev_idle)
        if (b.device == device) n += b.bytes;
    return n;
}

hipError_t dev_malloc(void **p, size_t bytes, bool optional) {
    *p = nullptr;
    if (fail_alloc_tick()) return hipErrorOutOfMemory;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const size_t want = dev_round(bytes);
    const bool caching = tun(TUN_DEVICE_CACHE_MB) > 0;
    // Room for the block's record BEFORE the block is taken (a failed push_back afterwards would lose it) — and kept for THIS
    // caller: the slots promised to calls still between here and their push_back are counted (`g_dev_pending`), so two concurrent
    // allocations cannot be promised the same one (round-5 advice); growth is geometric, not one exact-fit reallocation per block.
    try {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        const size_t need = g_dev_live.size() + g_dev_pending + 1;
        if (g_dev_live.capacity() < need) g_dev_live.reserve(std::max(need, 2 * g_dev_live.capacity()));
        g_dev_pending++;
    } catch (...) {
        return hipErrorOutOfMemory;
    }
    struct Promise {  // gives the slot back on every way out that did not use it
        bool used = false;
        ~Promise() {
            if (!used) {
                std::lock_guard<std::mutex> lk(g_dev_mu);
                g_dev_pending--;
            }
        }
    } promise;
    if (caching) {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        size_t best = g_dev_idle.size();
        for (size_t i = 0; i < g_dev_idle.size(); i++) {
            const DevBlock &b = g_dev_idle[i];
            // a block of the size asked for, or a little larger (never one that would strand more than an eighth)
            if (b.device == device && b.bytes >= want && b.bytes - want <= std::max<size_t>(want / 8, 2u << 20) &&
                (best == g_dev_idle.size() || b.bytes < g_dev_idle[best].bytes))
                best = i;
        }
        if (best != g_dev_idle.size()) {
            const DevBlock b = g_dev_idle[best];
            g_dev_idle.erase(g_dev_idle.begin() + (ptrdiff_t)best);
            g_dev_idle_bytes -= b.bytes;
            g_dev_live.push_back(b);
            g_dev_pending--;
            promise.used = true;
            *p = b.p;
            return hipSuccess;
        }
    }
    void *q = nullptr;
    e = hipMalloc(&q, want);
    if (e != hipSuccess && !optional && dev_cache_trim(device) > 0) {  // out of memory with idle blocks on the shelf: give them back, retry
        (void)hipGetLastError();
        e = hipMalloc(&q, want);
    }
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        g_dev_live.push_back(DevBlock{q, want, device});
        g_dev_pending--;
        promise.used = true;
    }
    *p = q;
    return hipSuccess;
}

// idle blocks beyond the budget, oldest first (called with g_dev_mu held; the caller frees them after unlocking)
static void dev_idle_over_budget(size_t limit, std::vector<DevBlock> *drop) {
    while (g_dev_idle_bytes > limit && !g_dev_idle.empty()) {
        drop->push_back(g_dev_idle.front());
        g_dev_idle_bytes -= g_dev_idle.front().bytes;
        g_dev_idle.erase(g_dev_idle.begin());
    }
}
static void dev_release_blocks(const std::vector<DevBlock> &drop) {
    for (const DevBlock &b : drop) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != b.device) (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
        if (cur >= 0 && cur != b.device) (void)hipSetDevice(cur);
    }
}

hipError_t dev_free_unused(void *p) {
    if (!p) return hipSuccess;
    NoFailScope no_fail;
    const size_t limit = (size_t)std::max<long long>(0, tun(TUN_DEVICE_CACHE_MB)) << 20;
    std::vector<DevBlock> drop;
    hipError_t e = hipErrorInvalidValue;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        for (size_t i = 0; i < g_dev_live.size(); i++)
            if (g_dev_live[i].p == p) {
                g_dev_idle.push_back(g_dev_live[i]);
                g_dev_idle_bytes += g_dev_live[i].bytes;
                g_dev_live.erase(g_dev_live.begin() + (ptrdiff_t)i);
                e = hipSuccess;
                break;
            }
        dev_idle_over_budget(limit, &drop);  // (ah_dataset_reserve_build parks tens of GB: within the same budget)
    }
    dev_release_blocks(drop);
    return e;
}

hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    NoFailScope no_fail;
    DevBlock blk{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        for (size_t i = 0; i < g_dev_live.size(); i++)
            if (g_dev_live[i].p == p) {
                blk = g_dev_live[i];
                g_dev_live.erase(g_dev_live.begin() + (ptrdiff_t)i);
                break;
            }
    }
    if (!blk.p) return hipFree(p);  // not ours (cannot happen: every allocation of the library comes from dev_malloc)
    const size_t limit = (size_t)std::max<long long>(0, tun(TUN_DEVICE_CACHE_MB)) << 20;
    if (limit == 0) return hipFree(p);
    // hipFree waits for the device; a cached block must not change hands while a queued kernel may still touch it either
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != blk.device) (void)hipSetDevice(blk.device);
    const hipError_t e = hipDeviceSynchronize();
    if (prev >= 0 && prev != blk.device) (void)hipSetDevice(prev);
    std::vector<DevBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        g_dev_idle.push_back(blk);
        g_dev_idle_bytes += blk.bytes;
        dev_idle_over_budget(limit, &drop);  // over the budget: the oldest go back to the driver
    }
    dev_release_blocks(drop);
    return e;
}

void Context::destroy() {
    NoFailScope no_fail;
    if (d_scratch) (void)dev_free(d_scratch);
    if (d_filter) (void)dev_free(d_filter);
    if (d_multi) (void)dev_free(d_multi);
    d_multi = nullptr;
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_t0) (void)hipEventDestroy(ev_t0);
    if (ev_t1) (void)hipEventDestroy(ev_t1);
    ev_t0 = ev_t1 = nullptr;
    for (hipEvent_t &e : ev_ring)
        if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    copy_stream = nullptr;
    d_scratch = h_pinned = d_filter = nullptr;
    d_cap = h_cap = d_filter_cap = 0;
    stream = nullptr;
}

// carve `bytes` (256-byte aligned) out of a linear scratch region
struct Carver {
    uint8_t *base;
    size_t off = 0;
    explicit Carver(void *b) : base(reinterpret_cast<uint8_t *>(b)) {}
    template <typename T>
    T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};
static inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// The host side of staging is a gather of records out of (unaligned) storage pages into pinned memory; one core
// cannot keep a PCIe Gen5 link busy (a 32 MiB chunk takes 0.6 ms on the wire), so chunks are spread over a small
// process-wide pool of persistent workers (spawning threads per chunk costs as much as the copy itself).
class WorkerPool {
   public:
    static WorkerPool &get() {
        static WorkerPool pool;
        return pool;
    }
    size_t size() const { return threads_.size() + 1; }
    // fn(part) for part in [0, parts), on the workers and the calling thread; returns when all parts are done.  If
    // another caller holds the pool (concurrent readers staging candidate lists), the work simply runs inline.
    template <typename F>
    void run(size_t parts, F &&fn) {
        if (parts <= 1 || threads_.empty() || !run_mu_.try_lock()) {
            for (size_t p = 0; p < parts; p++) fn(p);
            return;
        }
        std::function<void(size_t)> job = std::ref(fn);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            parts_ = parts;
            next_.store(0, std::memory_order_relaxed);
            pending_.store(parts, std::memory_order_relaxed);
            generation_++;
        }
        cv_.notify_all();
        work(&job, parts);
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();  // slices are ~100 us
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = nullptr;
        }
        // a worker may still be between "saw the job" and "took no slice": wait until none holds the pointer
        while (inside_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        run_mu_.unlock();
    }

   private:
    WorkerPool() {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned n = std::min(12u, std::max(1u, hw / 2));
        if (tun(TUN_STAGE_THREADS) > 0) n = (unsigned)tun(TUN_STAGE_THREADS);
        for (unsigned i = 1; i < n; i++) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void work(const std::function<void(size_t)> *job, size_t parts) {
        for (;;) {
            const size_t p = next_.fetch_add(1, std::memory_order_relaxed);
            if (p >= parts) return;
            (*job)(p);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *job;
            size_t parts;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
                parts = parts_;
                if (!job) continue;
                inside_.fetch_add(1, std::memory_order_acquire);
            }
            work(job, parts);
            inside_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_;
    const std::function<void(size_t)> *job_ = nullptr;
    size_t parts_ = 0;
    std::atomic<size_t> next_{0}, pending_{0};
    std::atomic<int> inside_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// Row copy into the pinned ring with non-temporal stores: the ring is written once and read by the DMA engine, so the
// destination lines need not be read for ownership nor kept in the host caches (a plain memcpy of 3 KB rows moves 3 bytes
// per byte staged; this moves 2).  `dst` is 32-byte aligned (ring rows start on 128-byte lines), `src` is arbitrary —
// LMDB hands out vectors at odd offsets (src/parallel.rs:296-311).
__attribute__((target("avx2"))) static void copy_row_stream_avx2(uint8_t *dst, const uint8_t *src, size_t bytes) {
    size_t i = 0;
    for (; i + 128 <= bytes; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    for (; i + 32 <= bytes; i += 32)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i)));
    if (i < bytes) memcpy(dst + i, src + i, bytes - i);
}
static inline void copy_row(uint8_t *dst, const uint8_t *src, size_t bytes) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && !tun(TUN_STAGE_MEMCPY) && bytes >= 512 && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0) copy_row_stream_avx2(dst, src, bytes);
    else memcpy(dst, src, bytes);
}
static inline void copy_fence() { _mm_sfence(); }  // non-temporal stores are ordered before the DMA is enqueued

// rows / ids worth a thread: below ~2 MiB in total the copy runs inline (waking workers costs more than it saves)
template <typename F>
static void parallel_rows(size_t n, size_t bytes_per_item, F &&fn) {
    WorkerPool &pool = WorkerPool::get();
    const size_t total = n * bytes_per_item;
    const size_t parts = total < (2u << 20) ? 1 : std::min<size_t>(pool.size(), total >> 19);
    if (parts <= 1) {
        fn((size_t)0, n);
        return;
    }
    const size_t per = (n + parts - 1) / parts;
    pool.run(parts, [&](size_t p) {
        const size_t lo = p * per, hi = std::min(n, lo + per);
        if (lo < hi) fn(lo, hi);
    });
}

bool guard_active();
}  // namespace ah

// The library's own allocation functions (hidden visibility: they replace `operator new` for the objects linked into
// libarroy_hip.so only, never for the embedding process).  malloc / free underneath, like libstdc++'s, so memory may cross
// between the two; the one addition is AH_FAIL_ALLOC_AFTER (common.h): inside an entry point the n-th allocation throws
// std::bad_alloc, which is how tests/test_gpu_faults.py proves that every entry point turns it into a status code.
static inline void *ah_alloc_or_null(size_t n) {
    if (ah::guard_active() && ah::fail_alloc_tick()) return nullptr;
    return malloc(n ? n : 1);
}
void *operator new(size_t n) {
    void *p = ah_alloc_or_null(n);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new[](size_t n) {
    void *p = ah_alloc_or_null(n);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(size_t n, const std::nothrow_t &) noexcept { return ah_alloc_or_null(n); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return ah_alloc_or_null(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { free(p); }

using namespace ah;

ah::DataView ah_dataset::view() const {
    DataView v;
    v.metric = metric;
    v.dims = dims;
    v.pitch = pitch;
    v.words = words;
    v.n = n;
    v.rows_f32 = d_rows_f32;
    v.rows_bq = d_rows_bq;
    v.headers = d_headers;
    v.ids = d_ids;
    v.lut = d_lut;
    v.lut_len = lut_len;
    v.identity_ids = identity_ids ? 1 : 0;
    return v;
}

// Contexts (stream + events + pinned / device scratch) of destroyed datasets are kept for the next dataset on the same
// device: page-locking a staging ring costs ~90 ms per 100 MB (measured), more than staging 1 GB of records.
namespace {
std::mutex g_ctx_mu;
std::vector<std::pair<int, ah::Context *>> g_ctx_cache;  // (device, context); deliberately never destroyed at exit
constexpr size_t kCtxCacheMax = 4;
}  // namespace

ah::Context *ah_dataset::acquire() {
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!pool.empty()) {
            Context *c = pool.back();
            pool.pop_back();
            return c;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t i = g_ctx_cache.size(); i-- > 0;)
            if (g_ctx_cache[i].first == device) {
                Context *c = g_ctx_cache[i].second;
                g_ctx_cache.erase(g_ctx_cache.begin() + (long)i);
                return c;
            }
    }
    Context *c = new (std::nothrow) Context();
    if (!c) return nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        c->destroy();
        delete c;
        return nullptr;
    }
    return c;
}
void ah_dataset::release(ah::Context *c) {
    NoFailScope no_fail;
    std::lock_guard<std::mutex> lk(mu);
    pool.push_back(c);
}

#define AH_LEASE(ds, name)                                                             \
    AH_HIP(hipSetDevice((ds)->device));                                                \
    ContextLease name##_lease(ds);                                                     \
    AH_REQUIRE(name##_lease.c != nullptr, AH_ERR_DEVICE, "cannot create a HIP stream"); \
    Context *name = name##_lease.c

extern "C" {

size_t ah_header_size(int metric) { return metric_valid(metric) ? 4 * header_floats(metric) : 0; }
size_t ah_vector_size(int metric, uint32_t dimensions) {
    if (!metric_valid(metric)) return 0;
    return metric_is_bq(metric) ? (size_t)bq_words(dimensions) * 8 : (size_t)dimensions * 4;
}
int ah_abi_version(void) { return AH_ABI_VERSION; }
const char *ah_last_error(void) { return ah::last_error(); }
int ah_last_error_detail(ah_error_detail *out) {
    AH_GUARDED("ah_last_error_detail")
    if (!out) return AH_ERR_INVALID_ARGUMENT;
    *out = g_detail;
    return AH_OK;
    AH_GUARDED_END
}

int ah_device_count(int *out_count) {
    AH_GUARDED("ah_device_count")
    AH_REQUIRE(out_count, AH_ERR_INVALID_ARGUMENT, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out_count = n;
    return AH_OK;
    AH_GUARDED_END
}

int ah_device_name(int device, char *buf, size_t buf_len) {
    AH_GUARDED("ah_device_name")
    AH_REQUIRE(buf && buf_len, AH_ERR_INVALID_ARGUMENT, "buf is NULL");
    hipDeviceProp_t p;
    AH_HIP(hipGetDeviceProperties(&p, device));
    snprintf(buf, buf_len, "%s (%s, %d CUs, %.1f GiB)", p.name, p.gcnArchName, p.multiProcessorCount,
             (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return AH_OK;
    AH_GUARDED_END
}

static TunableSlot *find_tunable(const char *name) {
    TunableSlot *t = tunable_table();
    for (int i = 0; i < TUN_COUNT; i++)
        if (name && strcmp(name, t[i].name) == 0) return t + i;
    return nullptr;
}
int ah_tuning_set(const char *name, int64_t value) {
    AH_GUARDED("ah_tuning_set")
    TunableSlot *t = find_tunable(name);
    AH_REQUIRE(t, AH_ERR_INVALID_ARGUMENT, "unknown tunable %s", name ? name : "(null)");
    t->value.store(value, std::memory_order_relaxed);
    return AH_OK;
    AH_GUARDED_END
}
int ah_tuning_get(const char *name, int64_t *out_value, int64_t *out_default) {
    AH_GUARDED("ah_tuning_get")
    TunableSlot *t = find_tunable(name);
    AH_REQUIRE(t, AH_ERR_INVALID_ARGUMENT, "unknown tunable %s", name ? name : "(null)");
    if (out_value) *out_value = t->value.load(std::memory_order_relaxed);
    if (out_default) *out_default = t->def;
    return AH_OK;
    AH_GUARDED_END
}
int ah_tuning_reset(void) {
    AH_GUARDED("ah_tuning_reset")
    TunableSlot *t = tunable_table();
    for (int i = 0; i < TUN_COUNT; i++) t[i].value.store(t[i].def, std::memory_order_relaxed);
    return AH_OK;
    AH_GUARDED_END
}

// ---------------------------------------------------------------------------------------------
// dataset
// ---------------------------------------------------------------------------------------------
int ah_dataset_create(int metric, uint32_t dimensions, uint64_t capacity, int device, ah_dataset **out) {
    AH_GUARDED("ah_dataset_create")
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(metric_valid(metric), AH_ERR_INVALID_ARGUMENT, "unknown metric %d", metric);
    AH_REQUIRE(dimensions > 0, AH_ERR_INVALID_DIMENSION, "dimensions must be > 0");
    AH_REQUIRE(capacity < 0xFFFFFFFFull, AH_ERR_INVALID_ARGUMENT, "capacity exceeds the u32 item-id space");
    int n_dev = 0;
    AH_HIP(hipGetDeviceCount(&n_dev));
    AH_REQUIRE(device >= 0 && device < n_dev, AH_ERR_DEVICE, "device %d not present (%d visible)", device, n_dev);
    AH_HIP(hipSetDevice(device));
    ah_dataset *ds = new (std::nothrow) ah_dataset();
    AH_REQUIRE(ds, AH_ERR_OUT_OF_MEMORY, "host allocation failed");
    ds->metric = metric;
    ds->dims = dimensions;
    ds->device = device;
    ds->capacity = capacity;
    const uint64_t cap = capacity ? capacity : 1;
    int st = AH_OK;
    do {
        hipError_t e;
        if (metric_is_bq(metric)) {
            ds->words = bq_words(dimensions);
            ds->pitch = (ds->words + 1u) & ~1u;  // 16-byte rows
            e = dev_malloc((void **)&ds->d_rows_bq, cap * ds->pitch * 8);
        } else {
            ds->pitch = (dimensions + 31u) & ~31u;  // 128-byte rows
            e = dev_malloc((void **)&ds->d_rows_f32, cap * (uint64_t)ds->pitch * 4);
        }
        if (e == hipSuccess) e = dev_malloc((void **)&ds->d_headers, cap * header_floats(metric) * 4);
        if (e == hipSuccess) e = dev_malloc((void **)&ds->d_ids, cap * 4);
        if (e != hipSuccess) {
            set_error("hipMalloc of %llu items x %u dims failed: %s", (unsigned long long)cap, dimensions,
                      hipGetErrorString(e));
            st = e == hipErrorOutOfMemory ? AH_ERR_OUT_OF_MEMORY : AH_ERR_DEVICE;
        }
    } while (0);
    if (st != AH_OK) {
        ah_dataset_destroy(ds);
        return st;
    }
    dataset_born(device);
    ds->counted = true;
    *out = ds;
    return AH_OK;
    AH_GUARDED_END
}

static int upload_flush(ah_dataset *ds);

int ah_dataset_destroy(ah_dataset *ds) {
    AH_GUARDED("ah_dataset_destroy")
    if (!ds) return AH_OK;
    NoFailScope no_fail;
    ds->join_reserve();
    (void)upload_flush(ds);
    (void)hipSetDevice(ds->device);
    (void)hipDeviceSynchronize();
    if (ds->up_ctx) {
        ds->pool.push_back(ds->up_ctx);
        ds->up_ctx = nullptr;
    }
    for (Context *c : ds->pool) {
        bool kept = false;
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            if (g_ctx_cache.size() < kCtxCacheMax && c->h_cap <= (512u << 20) && c->d_cap <= (1u << 30)) {
                g_ctx_cache.push_back({ds->device, c});
                kept = true;
            }
        }
        if (!kept) {
            c->destroy();
            delete c;
        }
    }
    ds->pool.clear();
    if (ds->d_rows_h16) (void)dev_free(ds->d_rows_h16);
    if (ds->d_rows_i8) (void)dev_free(ds->d_rows_i8);
    if (ds->d_rows_i8_lo) (void)dev_free(ds->d_rows_i8_lo);
    if (ds->d_scale8_rows) (void)dev_free(ds->d_scale8_rows);
    if (ds->d_dim_scale) (void)dev_free(ds->d_dim_scale);
    if (ds->d_screen_stats) (void)dev_free(ds->d_screen_stats);
    if (ds->d_rows_f32) (void)dev_free(ds->d_rows_f32);
    if (ds->d_rows_bq) (void)dev_free(ds->d_rows_bq);
    if (ds->d_headers) (void)dev_free(ds->d_headers);
    if (ds->d_ids) (void)dev_free(ds->d_ids);
    if (ds->d_lut) (void)dev_free(ds->d_lut);
    const bool counted = ds->counted;
    const int device = ds->device;
    delete ds;
    if (counted) dataset_gone(device);  // the last dataset of the device / the process: the caches go back (common.h)
    return AH_OK;
    AH_GUARDED_END
}

static int check_append(ah_dataset *ds, const uint32_t *item_ids, size_t n) {
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    AH_REQUIRE(!ds->finalized, AH_ERR_INVALID_ARGUMENT, "dataset already finalized");
    AH_REQUIRE(item_ids || n == 0, AH_ERR_INVALID_ARGUMENT, "item_ids is NULL");
    AH_REQUIRE(ds->n + n <= ds->capacity, AH_ERR_INVALID_ARGUMENT, "upload of %zu items exceeds capacity %llu", n,
               (unsigned long long)ds->capacity);
    for (size_t i = 0; i < n; i++) {
        const bool first = ds->n == 0 && i == 0;
        const uint32_t prev = i == 0 ? ds->last_id : item_ids[i - 1];
        AH_REQUIRE(first || item_ids[i] > prev, AH_ERR_INVALID_ARGUMENT,
                   "item ids must be strictly ascending (id %u after %u)", item_ids[i], prev);
    }
    return AH_OK;
}

static void note_ids(ah_dataset *ds, const uint32_t *item_ids, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (item_ids[i] != (uint32_t)(ds->n + i)) ds->identity_ids = false;
        ds->h_ids.push_back(item_ids[i]);
    }
    if (n) ds->last_id = item_ids[n - 1];
    ds->n += n;
}

// ---- staging ring ---------------------------------------------------------------------------------------------------
// Uploads are asynchronous until ah_dataset_finalize: the dataset keeps one Context (stream + pinned ring of kRing
// buffers) for all its upload calls; a call returns once its records are COPIED OUT of the caller's pages (the contract:
// no host pointer is used after return) while the last DMA transfers may still be in flight.  The host-side gather of a
// chunk is spread over the worker pool (one core cannot keep a PCIe Gen5 link busy), chunk c+1 is gathered while chunk c
// and c-1 travel.
static constexpr int kRing = 3;
static constexpr size_t kStageBytes = 32u << 20;

static int upload_flush(ah_dataset *ds) {
    if (!ds->up_ctx) return AH_OK;
    Context *c = ds->up_ctx;
    const hipError_t e = hipStreamSynchronize(c->stream);
    ds->up_ctx = nullptr;
    for (bool &u : ds->up_used) u = false;
    ds->up_buf = 0;
    ds->release(c);
    if (e != hipSuccess) {
        set_error("staging copy failed: %s", hipGetErrorString(e));
        set_error_status(AH_ERR_DEVICE);
        return AH_ERR_DEVICE;
    }
    return AH_OK;
}

// the dataset's upload context with a pinned ring of kRing x buf_bytes (grown only when idle)
static int upload_context(ah_dataset *ds, size_t buf_bytes, Context **out) {
    AH_HIP(hipSetDevice(ds->device));
    if (!ds->up_ctx) {
        ds->up_ctx = ds->acquire();
        AH_REQUIRE(ds->up_ctx, AH_ERR_DEVICE, "cannot create a HIP stream");
    }
    Context *c = ds->up_ctx;
    for (hipEvent_t &e : c->ev_ring)
        if (!e) AH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (c->h_cap < (size_t)kRing * buf_bytes) {
        AH_HIP(hipStreamSynchronize(c->stream));  // the old ring may still be a DMA source
        for (bool &u : ds->up_used) u = false;
        AH_TRY(c->ensure_pinned((size_t)kRing * buf_bytes));
    }
    *out = c;
    return AH_OK;
}

// LMDB pages -> pinned staging (header and vector split apart, rows re-pitched to 128-byte lines) -> hipMemcpyAsync.
int ah_dataset_upload_records(ah_dataset *ds, const uint32_t *item_ids, const uint8_t *const *record_ptrs,
                              size_t record_len, size_t n) {
    AH_GUARDED("ah_dataset_upload_records")
    AH_TRY(check_append(ds, item_ids, n));
    if (n == 0) return AH_OK;
    AH_REQUIRE(item_ids && record_ptrs, AH_ERR_INVALID_ARGUMENT, "NULL input");
    const size_t hs = ah_header_size(ds->metric), vs = ah_vector_size(ds->metric, ds->dims);
    // src/node.rs:252-258: [LEAF_TAG=0][header][vector]; a length mismatch is what UnalignedVector::from_bytes
    // / the dimension check reports as InvalidVecDimension
    if (record_len != 1 + hs + vs) {
        set_error("record length %zu does not match 1 + %zu + %zu for %u dimensions", record_len, hs, vs, ds->dims);
        set_error_status(AH_ERR_INVALID_DIMENSION);
        set_error_detail(0, 1 + hs + vs, record_len);
        return AH_ERR_INVALID_DIMENSION;
    }
    const size_t rb = ds->row_bytes();
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, kStageBytes / (rb + hs + 4)));
    const size_t buf_bytes = pad256(chunk * rb) + pad256(chunk * hs) + pad256(chunk * 4);
    Context *ctx = nullptr;
    AH_TRY(upload_context(ds, buf_bytes, &ctx));
    const size_t ring_stride = ctx->h_cap / kRing & ~(size_t)255;
    size_t done = 0;
    while (done < n) {
        const size_t c = std::min(chunk, n - done);
        const int b = ds->up_buf;
        uint8_t *base = reinterpret_cast<uint8_t *>(ctx->h_pinned) + (size_t)b * ring_stride;
        if (ds->up_used[b]) AH_HIP(hipEventSynchronize(ctx->ev_ring[b]));
        uint8_t *h_rows = base, *h_hdr = base + pad256(chunk * rb), *h_ids = h_hdr + pad256(chunk * hs);
        for (size_t i = 0; i < c; i++) {
            const uint8_t *rec = record_ptrs[done + i];
            AH_REQUIRE(rec && rec[0] == 0, AH_ERR_INVALID_ARGUMENT, "record %zu is not a leaf (tag %d)", done + i,
                       rec ? rec[0] : -1);
        }
        parallel_rows(c, rb + hs, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                const uint8_t *rec = record_ptrs[done + i];
                memcpy(h_hdr + i * hs, rec + 1, hs);
                copy_row(h_rows + i * rb, rec + 1 + hs, vs);
                if (rb > vs) memset(h_rows + i * rb + vs, 0, rb - vs);
            }
            copy_fence();
        });
        memcpy(h_ids, item_ids + done, c * 4);
        const uint64_t row0 = ds->n + done;
        uint8_t *d_rows = ds->d_rows_f32 ? reinterpret_cast<uint8_t *>(ds->d_rows_f32)
                                         : reinterpret_cast<uint8_t *>(ds->d_rows_bq);
        AH_HIP(hipMemcpyAsync(d_rows + row0 * rb, h_rows, c * rb, hipMemcpyHostToDevice, ctx->stream));
        AH_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t *>(ds->d_headers) + row0 * hs, h_hdr, c * hs,
                              hipMemcpyHostToDevice, ctx->stream));
        AH_HIP(hipMemcpyAsync(ds->d_ids + row0, h_ids, c * 4, hipMemcpyHostToDevice, ctx->stream));
        AH_HIP(hipEventRecord(ctx->ev_ring[b], ctx->stream));
        ds->up_used[b] = true;
        ds->up_buf = (b + 1) % kRing;
        done += c;
    }
    note_ids(ds, item_ids, n);
    // The stored DotProduct headers are taken as they are: items written by `Writer::add_item` carry {0, 0} until
    // `DotProduct::preprocess` ran over the database (src/distance/dot_product.rs:119-165), so the dataset still needs
    // ah_preprocess_dot unless the caller states that the database was preprocessed (ah_dataset_set_preprocessed).
    return AH_OK;
    AH_GUARDED_END
}

// Writer::add_item for a batch: stage f32 rows, then codec + new_header on device.
int ah_dataset_upload_vectors(ah_dataset *ds, const uint32_t *item_ids, const float *vectors, size_t n) {
    AH_GUARDED("ah_dataset_upload_vectors")
    AH_TRY(check_append(ds, item_ids, n));
    if (n == 0) return AH_OK;
    AH_REQUIRE(item_ids && vectors, AH_ERR_INVALID_ARGUMENT, "NULL input");
    const bool bq = metric_is_bq(ds->metric);
    const uint32_t fpitch = bq ? ((ds->dims + 3u) & ~3u) : ds->pitch;  // staging pitch in floats
    const size_t frb = (size_t)fpitch * 4;
    DataView dv = ds->view();
    // Rows that need no re-pitching can travel straight from the caller's memory when it is (or can be) page-locked:
    // AH_STAGE_REGISTER=1 registers the caller's buffer for the duration of the call (measurement aid).
    const bool try_register = tun(TUN_STAGE_REGISTER) != 0;
    if (try_register && !bq && fpitch == ds->dims && n * frb >= (64u << 20)) {
        Context *ctx = nullptr;
        AH_TRY(upload_context(ds, pad256(std::min<size_t>(n, kStageBytes / 4) * 4), &ctx));
        if (hipHostRegister(const_cast<float *>(vectors), n * frb, hipHostRegisterDefault) == hipSuccess) {
            const uint64_t row0 = ds->n;
            hipError_t e = hipMemcpyAsync(ds->d_rows_f32 + row0 * ds->pitch, vectors, n * frb, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ds->d_ids + row0, item_ids, n * 4, hipMemcpyHostToDevice, ctx->stream);
            int st = e == hipSuccess ? launch_headers_from_vectors(dv, row0, n, ctx->stream) : AH_ERR_DEVICE;
            const hipError_t es = hipStreamSynchronize(ctx->stream);  // the caller's pages are the DMA source
            (void)hipHostUnregister(const_cast<float *>(vectors));
            AH_REQUIRE(e == hipSuccess && es == hipSuccess && st == AH_OK, AH_ERR_DEVICE, "registered staging copy failed");
            note_ids(ds, item_ids, n);
            return AH_OK;
        }
        (void)hipGetLastError();  // not registrable: fall through to the bounce path
    }
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(n, kStageBytes / (frb + 4)));
    const size_t buf_bytes = pad256(chunk * frb) + pad256(chunk * 4);
    const bool timing = tun(TUN_TIMING) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    const auto t_begin = now();
    double t_gather = 0, t_wait = 0;
    Context *ctx = nullptr;
    AH_TRY(upload_context(ds, buf_bytes, &ctx));
    const auto t_ctx = now();
    const size_t ring_stride = ctx->h_cap / kRing & ~(size_t)255;
    if (bq) {
        if (ctx->d_cap < (size_t)kRing * pad256(chunk * frb)) AH_HIP(hipStreamSynchronize(ctx->stream));
        AH_TRY(ctx->ensure_device((size_t)kRing * pad256(chunk * frb)));
    }
    size_t done = 0;
    while (done < n) {
        const size_t c = std::min(chunk, n - done);
        const int b = ds->up_buf;
        uint8_t *base = reinterpret_cast<uint8_t *>(ctx->h_pinned) + (size_t)b * ring_stride;
        const auto t0 = now();
        if (ds->up_used[b]) AH_HIP(hipEventSynchronize(ctx->ev_ring[b]));
        const auto t1 = now();
        float *h_rows = reinterpret_cast<float *>(base);
        uint8_t *h_ids = base + pad256(chunk * frb);
        parallel_rows(c, frb, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                copy_row(reinterpret_cast<uint8_t *>(h_rows + i * fpitch),
                         reinterpret_cast<const uint8_t *>(vectors + (done + i) * (size_t)ds->dims), (size_t)ds->dims * 4);
                for (uint32_t e = ds->dims; e < fpitch; e++) h_rows[i * fpitch + e] = 0.0f;
            }
            copy_fence();
        });
        t_wait += secs(t0, t1);
        t_gather += secs(t1, now());
        memcpy(h_ids, item_ids + done, c * 4);
        const uint64_t row0 = ds->n + done;
        if (!bq) {
            AH_HIP(hipMemcpyAsync(ds->d_rows_f32 + row0 * ds->pitch, h_rows, c * frb, hipMemcpyHostToDevice,
                                  ctx->stream));
        } else {
            float *d_tmp = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(ctx->d_scratch) +
                                                     (size_t)b * pad256(chunk * frb));
            AH_HIP(hipMemcpyAsync(d_tmp, h_rows, c * frb, hipMemcpyHostToDevice, ctx->stream));
            AH_TRY(launch_quantize_rows(d_tmp, fpitch, ds->dims, ds->d_rows_bq + row0 * ds->pitch, ds->pitch, ds->words,
                                        c, ctx->stream));
        }
        AH_HIP(hipMemcpyAsync(ds->d_ids + row0, h_ids, c * 4, hipMemcpyHostToDevice, ctx->stream));
        AH_TRY(launch_headers_from_vectors(dv, row0, c, ctx->stream));
        AH_HIP(hipEventRecord(ctx->ev_ring[b], ctx->stream));
        ds->up_used[b] = true;
        ds->up_buf = (b + 1) % kRing;
        done += c;
    }
    const auto t_loop = now();
    note_ids(ds, item_ids, n);
    if (timing)
        fprintf(stderr, "[ah] upload_vectors %zu x %u: context + pinned ring %.4f s, gather %.4f s, waiting for the ring %.4f s, "
                        "launch/other %.4f s, note_ids %.4f s\n",
                n, ds->dims, secs(t_begin, t_ctx), t_gather, t_wait, secs(t_ctx, t_loop) - t_gather - t_wait, secs(t_loop, now()));
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_upload_flush(ah_dataset *ds) {
    AH_GUARDED("ah_dataset_upload_flush")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    return upload_flush(ds);
    AH_GUARDED_END
}

int ah_dataset_set_preprocessed(ah_dataset *ds, int preprocessed) {
    AH_GUARDED("ah_dataset_set_preprocessed")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    AH_REQUIRE(ds->metric == AH_DOT_PRODUCT, AH_ERR_INVALID_ARGUMENT, "only DotProduct datasets have a preprocess step");
    ds->dot_preprocessed = preprocessed != 0;
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_fill_synthetic(ah_dataset *ds, uint64_t seed, int distribution, uint64_t n_items) {
    AH_GUARDED("ah_dataset_fill_synthetic")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    AH_REQUIRE(!ds->finalized && ds->n == 0, AH_ERR_INVALID_ARGUMENT, "synthetic fill needs an empty dataset");
    AH_REQUIRE(n_items <= ds->capacity, AH_ERR_INVALID_ARGUMENT, "n_items exceeds capacity");
    AH_REQUIRE(distribution >= AH_SYNTH_UNIFORM_01 && distribution <= AH_SYNTH_LAST, AH_ERR_INVALID_ARGUMENT,
               "unknown distribution %d", distribution);
    AH_LEASE(ds, ctx);
    DataView dv = ds->view();
    if (!metric_is_bq(ds->metric)) {
        AH_TRY(launch_synth_fill(ds->d_rows_f32, ds->pitch, ds->dims, 0, n_items, seed, distribution, ctx->stream));
    } else {
        const uint32_t fpitch = (ds->dims + 3u) & ~3u;
        const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n_items, (256ull << 20) / ((uint64_t)fpitch * 4)));
        AH_TRY(ctx->ensure_device(chunk * fpitch * 4));
        for (uint64_t done = 0; done < n_items; done += chunk) {
            const uint64_t c = std::min(chunk, n_items - done);
            float *d_tmp = reinterpret_cast<float *>(ctx->d_scratch);
            AH_TRY(launch_synth_fill(d_tmp, fpitch, ds->dims, done, c, seed, distribution, ctx->stream));
            AH_TRY(launch_quantize_rows(d_tmp, fpitch, ds->dims, ds->d_rows_bq + done * ds->pitch, ds->pitch, ds->words, c,
                                        ctx->stream));
        }
    }
    AH_TRY(launch_headers_from_vectors(dv, 0, n_items, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    ds->n = n_items;
    ds->identity_ids = true;
    ds->last_id = n_items ? (uint32_t)(n_items - 1) : 0;
    ds->h_ids.clear();  // identity: no host mirror needed
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_finalize(ah_dataset *ds) {
    AH_GUARDED("ah_dataset_finalize")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    if (ds->finalized) return AH_OK;
    AH_TRY(upload_flush(ds));  // every staged record has landed
    AH_LEASE(ds, ctx);
    if (ds->identity_ids) {
        // ids 0..n-1: no table needed; the id array is still materialised for uniform kernels
        if (ds->h_ids.empty() && ds->n) {
            std::vector<uint32_t> ids(ds->n);
            for (uint64_t i = 0; i < ds->n; i++) ids[i] = (uint32_t)i;
            AH_HIP(hipMemcpy(ds->d_ids, ids.data(), ds->n * 4, hipMemcpyHostToDevice));
        }
    } else {
        const uint64_t span = (uint64_t)ds->last_id + 1;
        if (span <= 8 * ds->n + (1u << 20)) {  // dense table; otherwise kernels binary-search the id array
            AH_HIP(dev_malloc((void **)&ds->d_lut, span * 4));
            ds->lut_len = (uint32_t)span;
            AH_TRY(launch_build_lut(ds->d_ids, ds->n, ds->d_lut, ds->lut_len, ctx->stream));
            AH_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    ds->finalized = true;
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_len(const ah_dataset *ds, uint64_t *out_n_items) {
    AH_GUARDED("ah_dataset_len")
    AH_REQUIRE(ds && out_n_items, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    *out_n_items = ds->n;
    return AH_OK;
    AH_GUARDED_END
}

// One staged dataset -> a replica on another GPU of the node, device to device over xGMI (hipMemcpyPeerAsync): the
// multi-GPU build shards TREES over replicas of the read-only dataset, so a node stages the LMDB
// records once over PCIe and fans the HBM image out instead of staging N times.  The replica is in the same state as
// the source (finalized or not, DotProduct preprocessed or not); the binary16 shadow is rebuilt on the replica on
// demand (a 10 ms kernel) rather than copied.
// ah_dataset_replicate's copy engine.  Peer copies (xGMI) when the destination can address the source — the calling
// sequence of the reference's one-process build, one replica per GPU (src/writer.rs:556-591 shares ONE read-only view) — and a
// copy through pinned host memory when it cannot (no peer access between the two devices, IOMMU / container restrictions, or a
// peer copy that fails): slower, never wrong.  Either way the head and the tail of every replicated array are read back from
// both devices and compared before the replica is handed out.
struct ReplicaCopier {
    int src_dev, dst_dev;
    hipStream_t s;          // a stream of the destination device
    bool peer = false;      // peer copies are in use
    bool bounced = false;   // at least one array went through the host
    void *h_bounce = nullptr;
    static constexpr size_t kBounce = 64u << 20;
    std::string why;

    ReplicaCopier(int sd, int dd, hipStream_t st) : src_dev(sd), dst_dev(dd), s(st) {
        if (tun(TUN_REPLICATE_HOST_BOUNCE) != 0) {
            why = "AH_REPLICATE_HOST_BOUNCE";
            return;
        }
        if (sd == dd) {
            peer = true;  // (a copy inside one device)
            return;
        }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dd, sd) != hipSuccess || !can) {
            (void)hipGetLastError();
            why = "hipDeviceCanAccessPeer says no";
            return;
        }
        (void)hipSetDevice(dd);
        const hipError_t e = hipDeviceEnablePeerAccess(sd, 0);
        (void)hipGetLastError();
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e);
            return;
        }
        peer = true;
    }
    ~ReplicaCopier() {
        if (h_bounce) (void)hipHostFree(h_bounce);
    }
    hipError_t through_host(void *dst, const void *src, size_t bytes) {
        bounced = true;
        hipError_t e = hipSuccess;
        if (!h_bounce && (e = hipHostMalloc(&h_bounce, kBounce, hipHostMallocDefault)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        for (size_t off = 0; off < bytes && e == hipSuccess; off += kBounce) {
            const size_t len = std::min(kBounce, bytes - off);
            (void)hipSetDevice(src_dev);
            e = hipMemcpy(h_bounce, (const uint8_t *)src + off, len, hipMemcpyDeviceToHost);
            (void)hipSetDevice(dst_dev);
            if (e == hipSuccess) e = hipMemcpy((uint8_t *)dst + off, h_bounce, len, hipMemcpyHostToDevice);
        }
        return e;
    }
    // head and tail (up to 4 KiB each) of the two arrays, read back from both devices
    bool same_ends(const void *dst, const void *src, size_t bytes) {
        const size_t len = std::min<size_t>(bytes, 4096);
        uint8_t a[4096], b[4096];
        for (size_t off : {(size_t)0, bytes - len}) {
            (void)hipSetDevice(src_dev);
            if (hipMemcpy(a, (const uint8_t *)src + off, len, hipMemcpyDeviceToHost) != hipSuccess) return false;
            (void)hipSetDevice(dst_dev);
            if (hipMemcpy(b, (const uint8_t *)dst + off, len, hipMemcpyDeviceToHost) != hipSuccess) return false;
            if (memcmp(a, b, len) != 0) return false;
        }
        return true;
    }
    hipError_t copy(void *dst, const void *src, size_t bytes) {
        if (bytes == 0) return hipSuccess;
        hipError_t e = hipSuccess;
        if (peer) {
            e = hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess && same_ends(dst, src, bytes)) return hipSuccess;
            // a peer copy that fails, or that "succeeds" and delivers other bytes: do not trust the link again in this call
            (void)hipGetLastError();
            why = e != hipSuccess ? std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e) : "the peer copy delivered other bytes";
            peer = false;
        }
        e = through_host(dst, src, bytes);
        if (e == hipSuccess && !same_ends(dst, src, bytes)) e = hipErrorUnknown;
        return e;
    }
};

int ah_dataset_replicate(ah_dataset *src, int device, ah_dataset **out) {
    AH_GUARDED("ah_dataset_replicate")
    AH_REQUIRE(out, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    AH_REQUIRE(src, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    AH_TRY(upload_flush(src));
    // the calling thread may hold another device current (a host thread per GPU): put it back on every path out
    struct DeviceRestore {
        int prev = -1;
        DeviceRestore() {
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        }
        ~DeviceRestore() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } restore_device;
    ah_dataset *dst = nullptr;
    AH_TRY(ah_dataset_create(src->metric, src->dims, std::max<uint64_t>(src->capacity, 1), device, &dst));
    int st = AH_OK;
    do {
        ContextLease lease(dst);
        if (!lease.c) {
            set_error("cannot create a HIP stream");
            st = AH_ERR_DEVICE;
            break;
        }
        (void)hipSetDevice(device);
        ReplicaCopier cp(src->device, device, lease.c->stream);
        const size_t rb = src->row_bytes(), hs = ah_header_size(src->metric);
        const void *rows_src = src->d_rows_f32 ? (const void *)src->d_rows_f32 : (const void *)src->d_rows_bq;
        void *rows_dst = dst->d_rows_f32 ? (void *)dst->d_rows_f32 : (void *)dst->d_rows_bq;
        hipError_t e = hipSuccess;
        if (src->n) {
            e = cp.copy(rows_dst, rows_src, src->n * rb);
            if (e == hipSuccess) e = cp.copy(dst->d_headers, src->d_headers, src->n * hs);
            if (e == hipSuccess) e = cp.copy(dst->d_ids, src->d_ids, src->n * 4);
        }
        if (e == hipSuccess && src->d_lut) {
            (void)hipSetDevice(device);
            e = dev_malloc((void **)&dst->d_lut, (size_t)src->lut_len * 4);
            if (e == hipSuccess) e = cp.copy(dst->d_lut, src->d_lut, (size_t)src->lut_len * 4);
            dst->lut_len = src->lut_len;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("replicating the dataset from device %d to device %d failed%s%s: %s (peer access: %s)", src->device, device,
                      cp.bounced ? " even through pinned host memory" : "", cp.why.empty() ? "" : (" [" + cp.why + "]").c_str(),
                      e == hipErrorUnknown ? "the replica differs from its source" : hipGetErrorString(e), cp.peer ? "yes" : "no");
            set_error_status(AH_ERR_DEVICE);
            st = AH_ERR_DEVICE;
            break;
        }
        if (cp.bounced && tun(TUN_TIMING) != 0)
            fprintf(stderr, "[arroy-hip] replicate %d -> %d went through pinned host memory (%s)\n", src->device, device, cp.why.c_str());
        dst->replicated_through_host = cp.bounced;
        dst->n = src->n;
        dst->identity_ids = src->identity_ids;
        dst->last_id = src->last_id;
        dst->h_ids = src->h_ids;
        dst->finalized = src->finalized;
        dst->dot_preprocessed = src->dot_preprocessed;
    } while (0);
    if (st != AH_OK) {
        ah_dataset_destroy(dst);
        return st;
    }
    *out = dst;
    return AH_OK;
    AH_GUARDED_END
}

// host-side id -> row (the host mirror is only used to validate single ids; lists are resolved on device)
static int host_row_of_id(const ah_dataset *ds, uint32_t id, uint32_t *row) {
    if (ds->identity_ids) {
        if (id >= ds->n) {
            set_error("item %u does not exist", id);
            set_error_status(AH_ERR_MISSING_ITEM);
            set_error_detail(id, 0, 0);
            return AH_ERR_MISSING_ITEM;
        }
        *row = id;
        return AH_OK;
    }
    auto it = std::lower_bound(ds->h_ids.begin(), ds->h_ids.end(), id);
    if (it == ds->h_ids.end() || *it != id) {
        set_error("item %u does not exist", id);
        set_error_status(AH_ERR_MISSING_ITEM);
        set_error_detail(id, 0, 0);
        return AH_ERR_MISSING_ITEM;
    }
    *row = (uint32_t)(it - ds->h_ids.begin());
    return AH_OK;
}

#define AH_NEED_FINALIZED(ds)                                                                       \
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");                                     \
    AH_REQUIRE((ds)->finalized, AH_ERR_NOT_FINALIZED, "dataset not finalized (call ah_dataset_finalize)")

int ah_dataset_item_vector(ah_dataset *ds, uint32_t item_id, float *out_vector) {
    AH_GUARDED("ah_dataset_item_vector")
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(out_vector, AH_ERR_INVALID_ARGUMENT, "out_vector is NULL");
    uint32_t row;
    AH_TRY(host_row_of_id(ds, item_id, &row));
    AH_LEASE(ds, ctx);
    AH_TRY(ctx->ensure_device(pad256((size_t)ds->dims * 4)));
    AH_TRY(ctx->ensure_pinned((size_t)ds->dims * 4));
    AH_TRY(launch_decode_item(ds->view(), row, reinterpret_cast<float *>(ctx->d_scratch), ctx->stream));
    AH_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, (size_t)ds->dims * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out_vector, ctx->h_pinned, (size_t)ds->dims * 4);
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_read_headers(ah_dataset *ds, uint64_t first_row, uint64_t n, void *out_headers) {
    AH_GUARDED("ah_dataset_read_headers")
    AH_REQUIRE(ds && out_headers, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE(first_row + n <= ds->n, AH_ERR_INVALID_ARGUMENT, "row range out of bounds");
    AH_TRY(upload_flush(ds));
    AH_HIP(hipSetDevice(ds->device));
    const size_t hs = ah_header_size(ds->metric);
    AH_HIP(hipMemcpy(out_headers, reinterpret_cast<uint8_t *>(ds->d_headers) + first_row * hs, n * hs,
                     hipMemcpyDeviceToHost));
    return AH_OK;
    AH_GUARDED_END
}

int ah_preprocess_dot(ah_dataset *ds, float *out_max_norm) {
    AH_GUARDED("ah_preprocess_dot")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    AH_REQUIRE(ds->metric == AH_DOT_PRODUCT, AH_ERR_INVALID_ARGUMENT, "preprocess is only defined for DotProduct");
    AH_TRY(upload_flush(ds));
    AH_LEASE(ds, ctx);
    AH_TRY(ctx->ensure_device(256));
    AH_TRY(launch_preprocess_dot(ds->view(), reinterpret_cast<float *>(ctx->d_scratch), ctx->stream));
    float m = 0.0f;
    AH_HIP(hipMemcpyAsync(&m, ctx->d_scratch, 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    ds->dot_preprocessed = true;
    if (out_max_norm) *out_max_norm = m;
    return AH_OK;
    AH_GUARDED_END
}

// ---------------------------------------------------------------------------------------------
// search side
// ---------------------------------------------------------------------------------------------
struct QueryBufs {
    void *d_qvec;
    float *d_qhdr;
    float *d_qf32;
    uint32_t *d_err;
};

// Device scratch layout for one query call.  Returns the carver positioned after the query block.
static int stage_query(ah_dataset *ds, Context *ctx, const float *query, const uint32_t *query_item, size_t extra_dev,
                       size_t extra_pinned, QueryBufs *qb, Carver *dev_out, Carver *pin_out, bool zero_copy = false) {
    const size_t qbytes = pad256(ds->row_bytes()) + pad256(8) + pad256((size_t)ds->dims * 4) + pad256(4);
    AH_TRY(ctx->ensure_device(qbytes + extra_dev));
    AH_TRY(ctx->ensure_pinned(pad256((size_t)ds->dims * 4) + extra_pinned));
    Carver dev(ctx->d_scratch), pin(ctx->h_pinned);
    qb->d_qvec = dev.take<uint8_t>(ds->row_bytes());
    qb->d_qhdr = dev.take<float>(2);
    qb->d_qf32 = dev.take<float>(ds->dims);
    qb->d_err = dev.take<uint32_t>(1);
    float *h_q = pin.take<float>(ds->dims);
    AH_HIP(hipMemsetAsync(qb->d_err, 0, 4, ctx->stream));
    DataView dv = ds->view();
    if (query) {
        memcpy(h_q, query, (size_t)ds->dims * 4);
        // (zero_copy: a latency-bound caller — the kernel reads the pinned staging buffer over the link, no copy is queued)
        if (!zero_copy) AH_HIP(hipMemcpyAsync(qb->d_qf32, h_q, (size_t)ds->dims * 4, hipMemcpyHostToDevice, ctx->stream));
        AH_TRY(launch_prepare_query(dv, zero_copy ? h_q : qb->d_qf32, qb->d_qvec, qb->d_qhdr, ctx->stream));
    } else {
        uint32_t row;
        AH_TRY(host_row_of_id(ds, *query_item, &row));
        AH_TRY(launch_load_item_as_query(dv, row, qb->d_qvec, qb->d_qhdr, ctx->stream));
    }
    *dev_out = dev;
    *pin_out = pin;
    return AH_OK;
}

static int check_err_flags(uint32_t flags, bool need_sorted) {
    AH_REQUIRE((flags & 1u) == 0, AH_ERR_MISSING_ITEM, "a listed item id does not exist in the dataset");
    AH_REQUIRE(!need_sorted || (flags & 2u) == 0, AH_ERR_INVALID_ARGUMENT,
               "candidate ids must be ascending and unique (src/reader.rs:378-379)");
    return AH_OK;
}

static int distances_impl(ah_dataset *ds, const float *query, const uint32_t *query_item, const uint32_t *item_ids,
                          size_t n, float *out) {
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(out || n == 0, AH_ERR_INVALID_ARGUMENT, "out is NULL");
    AH_REQUIRE(item_ids || n <= ds->n, AH_ERR_INVALID_ARGUMENT, "n exceeds the number of items");
    if (n == 0) return AH_OK;
    AH_LEASE(ds, ctx);
    QueryBufs qb;
    Carver dev(nullptr), pin(nullptr);
    const size_t extra_dev = pad256(n * 4) * 2;
    const size_t extra_pin = pad256(n * 4) * 2 + 256;
    AH_TRY(stage_query(ds, ctx, query, query_item, extra_dev, extra_pin, &qb, &dev, &pin));
    uint32_t *d_ids = nullptr;
    if (item_ids) {
        d_ids = dev.take<uint32_t>(n);
        uint32_t *h_ids = pin.take<uint32_t>(n);
        memcpy(h_ids, item_ids, n * 4);
        AH_HIP(hipMemcpyAsync(d_ids, h_ids, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    float *d_out = dev.take<float>(n);
    float *h_out = pin.take<float>(n);
    uint32_t *h_err = pin.take<uint32_t>(1);
    AH_TRY(launch_distances(ds->view(), qb.d_qvec, qb.d_qhdr, d_ids, n, d_out, qb.d_err, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_out, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_err, qb.d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    AH_TRY(check_err_flags(*h_err, false));
    memcpy(out, h_out, n * 4);
    return AH_OK;
}

int ah_distances_by_vector(ah_dataset *ds, const float *query, const uint32_t *item_ids, size_t n, float *out) {
    AH_GUARDED("ah_distances_by_vector")
    AH_REQUIRE(query, AH_ERR_INVALID_ARGUMENT, "query is NULL");
    return distances_impl(ds, query, nullptr, item_ids, n, out);
    AH_GUARDED_END
}
int ah_distances_by_item(ah_dataset *ds, uint32_t query_item, const uint32_t *item_ids, size_t n, float *out) {
    AH_GUARDED("ah_distances_by_item")
    return distances_impl(ds, nullptr, &query_item, item_ids, n, out);
    AH_GUARDED_END
}

static int rerank_impl(ah_dataset *ds, const float *query, const uint32_t *query_item, const uint32_t *sorted_ids,
                       size_t n, size_t k, uint32_t *out_ids, float *out_distances, size_t *out_n) {
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(out_n, AH_ERR_INVALID_ARGUMENT, "out_n is NULL");
    *out_n = 0;
    if (!sorted_ids) n = ds->n;  // "all items"
    const size_t kk = std::min(k, n);  // src/reader.rs:394
    if (kk == 0) return AH_OK;
    AH_REQUIRE(out_ids && out_distances, AH_ERR_INVALID_ARGUMENT, "output buffers are NULL");
    AH_LEASE(ds, ctx);
    QueryBufs qb;
    Carver dev(nullptr), pin(nullptr);
    const size_t extra_dev = pad256(n * 4) * 2 + pad256(kk * 4) * 2 + pad256(topk_scratch_bytes(n, kk));
    const size_t extra_pin = pad256(n * 4) + pad256(kk * 4) * 2 + 256;
    // One short list (arroy's own call: search_k = 10 000-odd candidates of one query, src/reader.rs:381-399) is latency, not
    // bandwidth: the kernels read the query and the ids from the pinned staging buffers themselves, one block selects and writes
    // the results and the status word into pinned memory (k_topk_small) — four queue entries instead of twelve.
    const bool small = topk_small_fits(n, kk) && tun(TUN_RERANK_SMALL) != 0;
    AH_TRY(stage_query(ds, ctx, query, query_item, extra_dev, extra_pin, &qb, &dev, &pin, small && !metric_is_bq(ds->metric)));
    uint32_t *d_ids = nullptr;
    if (sorted_ids) {
        d_ids = dev.take<uint32_t>(n);
        uint32_t *h_ids = pin.take<uint32_t>(n);
        memcpy(h_ids, sorted_ids, n * 4);
        if (small) d_ids = h_ids;
        else AH_HIP(hipMemcpyAsync(d_ids, h_ids, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    float *d_dist = dev.take<float>(n);
    uint32_t *d_oi = dev.take<uint32_t>(kk);
    float *d_od = dev.take<float>(kk);
    void *d_tk = dev.take<uint8_t>(topk_scratch_bytes(n, kk));
    uint32_t *h_oi = pin.take<uint32_t>(kk);
    float *h_od = pin.take<float>(kk);
    uint32_t *h_err = pin.take<uint32_t>(1);
    DataView dv = ds->view();
    AH_TRY(launch_distances(dv, qb.d_qvec, qb.d_qhdr, d_ids, n, d_dist, qb.d_err, ctx->stream));
    if (small) {
        *h_err = 0xFFFFFFFFu;  // (overwritten by the kernel: a launch that never ran reads as "take the general path")
        AH_TRY(launch_topk_small(dv, d_dist, d_ids, n, kk, h_oi, h_od, qb.d_err, h_err, ctx->stream));
        AH_HIP(hipStreamSynchronize(ctx->stream));
        if (*h_err != 0xFFFFFFFFu && (*h_err & (4u | 8u)) == 0) {
            AH_TRY(check_err_flags(*h_err, true));
            memcpy(out_ids, h_oi, kk * 4);
            memcpy(out_distances, h_od, kk * 4);
            *out_n = kk;
            return AH_OK;
        }
        // a non-finite distance, or too many equal keys around the k-th: the general selection on the distances already there
    }
    AH_TRY(launch_topk(dv, d_dist, d_ids, n, kk, d_tk, d_oi, d_od, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_oi, d_oi, kk * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_od, d_od, kk * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_err, qb.d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    AH_TRY(check_err_flags(*h_err, true));
    memcpy(out_ids, h_oi, kk * 4);
    memcpy(out_distances, h_od, kk * 4);
    *out_n = kk;
    return AH_OK;
}

int ah_rerank_by_vector(ah_dataset *ds, const float *query, const uint32_t *sorted_ids, size_t n, size_t k,
                        uint32_t *out_ids, float *out_distances, size_t *out_n) {
    AH_GUARDED("ah_rerank_by_vector")
    AH_REQUIRE(query, AH_ERR_INVALID_ARGUMENT, "query is NULL");
    return rerank_impl(ds, query, nullptr, sorted_ids, n, k, out_ids, out_distances, out_n);
    AH_GUARDED_END
}
int ah_rerank_by_item(ah_dataset *ds, uint32_t query_item, const uint32_t *sorted_ids, size_t n, size_t k,
                      uint32_t *out_ids, float *out_distances, size_t *out_n) {
    AH_GUARDED("ah_rerank_by_item")
    return rerank_impl(ds, nullptr, &query_item, sorted_ids, n, k, out_ids, out_distances, out_n);
    AH_GUARDED_END
}

// Many queries in one submission.  Fast path (k <= 2048): five launches for the whole batch (batch.hip).
// Fallback (huge k): the single-query kernels queued back to back on one stream.
struct HostSeg {
    uint64_t off;
    uint32_t n, k;
};
struct HostTile {
    uint32_t query, first;
};

// AH_RERANK_TIMING=1: the wall time of one sub-batch of ah_rerank_batch by phase (include/arroy_hip.h: ah_rerank_stats).  The
// phases are disjoint stretches of the calling thread's time; the destructor runs after the last synchronisation of whichever
// return path was taken and adds them to the dataset's totals.
struct RerankProbe {
    using clk = std::chrono::steady_clock;
    ah_dataset *ds;
    Context *ctx;
    bool on;
    clk::time_point t_in, t_mark;
    double prep = 0, ids = 0, enqueue = 0, wait = 0;
    bool span = false;
    uint64_t nq, total;
    RerankProbe(ah_dataset *d, Context *c, uint64_t q, uint64_t t) : ds(d), ctx(c), on(tun(TUN_RERANK_TIMING) != 0), nq(q), total(t) {
        if (on) t_in = t_mark = clk::now();
    }
    // time since the last mark goes to `*bucket`
    void lap(double RerankProbe::*bucket) {
        if (!on) return;
        const auto now = clk::now();
        this->*bucket += std::chrono::duration<double>(now - t_mark).count();
        t_mark = now;
    }
    void first_enqueue(hipStream_t s) {  // the device span starts here
        if (!on) return;
        if (!ctx->ev_t0 && (hipEventCreate(&ctx->ev_t0) != hipSuccess || hipEventCreate(&ctx->ev_t1) != hipSuccess)) {
            (void)hipGetLastError();
            return;
        }
        span = hipEventRecord(ctx->ev_t0, s) == hipSuccess;
    }
    ~RerankProbe() {
        if (!on) return;
        lap(&RerankProbe::enqueue);  // (output copies after the last wait)
        float ms = 0.0f;
        if (span && hipEventRecord(ctx->ev_t1, ctx->stream) == hipSuccess && hipEventSynchronize(ctx->ev_t1) == hipSuccess &&
            hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1) != hipSuccess)
            ms = 0.0f;
        (void)hipGetLastError();
        const double wall = std::chrono::duration<double>(clk::now() - t_in).count();
        std::lock_guard<std::mutex> lk(ds->mu);
        ah_rerank_stats &r = ds->rr_stats;
        r.calls += 1;
        r.queries += nq;
        r.candidates += total;
        r.seconds_wall += wall;
        r.seconds_prep += prep;
        r.seconds_ids += ids;
        r.seconds_enqueue += enqueue;
        r.seconds_sync_wait += wait;
        r.seconds_device_span += ms * 1e-3;
    }
};
#define AH_RR_SYNC(stream_)                       \
    do {                                          \
        probe.lap(&RerankProbe::enqueue);         \
        AH_HIP(hipStreamSynchronize(stream_));    \
        probe.lap(&RerankProbe::wait);            \
    } while (0)

static int rerank_batch_chunk(ah_dataset *ds, Context *ctx, const float *queries, size_t nq, const uint32_t *ids,
                              const uint64_t *offsets, size_t k, uint32_t *out_ids, float *out_distances,
                              uint32_t *out_counts) {
    const uint64_t base = offsets[0];
    const uint64_t total = offsets[nq] - base;
    RerankProbe probe(ds, ctx, nq, total);
    uint32_t max_n = 0, max_rounds = 0;
    std::vector<HostSeg> segs(nq);
    std::vector<HostTile> tiles;
    const uint32_t tc = batch_tile_candidates();
    for (size_t q = 0; q < nq; q++) {
        const uint64_t nq_items = offsets[q + 1] - offsets[q];
        AH_REQUIRE(nq_items < 0xFFFFFFFFull, AH_ERR_INVALID_ARGUMENT, "candidate list too long");
        segs[q] = HostSeg{offsets[q] - base, (uint32_t)nq_items, (uint32_t)std::min<uint64_t>(k, nq_items)};
        out_counts[q] = segs[q].k;
        max_n = std::max(max_n, segs[q].n);
        max_rounds = std::max(max_rounds, batch_rounds(segs[q].n, segs[q].k));
        for (uint32_t f = 0; f < segs[q].n; f += tc) tiles.push_back(HostTile{(uint32_t)q, f});
    }
    const size_t qstride = pad256(ds->row_bytes());
    const size_t kstride = batch_key_stride(std::max<uint32_t>(max_n, 1));
    const size_t inv_bytes = batch_invert_wanted(ds->view(), total) ? batch_invert_counter_bytes(ds->n, total) : 0;
    // Certified top-k screen (search.hip: k_search_select_screened): Cosine / DotProduct lists long enough for the screen to
    // pay (the candidates on the binary16 copy of the rows — made by the first call that wants it — and only the few whose
    // proven distance interval reaches the top k in f32).  Submissions the row-major re-rank takes are left to it.
    bool screened = tun(TUN_RERANK_SCREEN) != 0 && (ds->metric == AH_COSINE || ds->metric == AH_DOT_PRODUCT) && ds->dims >= 32 &&
                    inv_bytes == 0 && k <= 256 && total >= 8 * k * nq && !tiles.empty();
    if (screened) screened = ensure_screen(ds, ctx->stream, false);
    // ... and, round 6, on the int8 copy before that: half the bytes again per candidate, a wider band of survivors
    const bool screened8 = screened && tun(TUN_RERANK_SCREEN8) != 0 && !ds->rerank8_off.load(std::memory_order_relaxed) &&
                           ensure_screen8_search(ds, ctx->stream);
    const size_t screen_bytes = (screened ? pad256(nq * (size_t)ds->hpitch * 2) + pad256(nq * 16) + pad256(total * 4) : 0) +
                                (screened8 ? pad256(nq * (size_t)ds->pitch8 * 2) + pad256(nq * 16) + pad256(total * 4) : 0);
    const size_t dev_bytes = pad256(nq * (size_t)ds->dims * 4) + nq * qstride + pad256(nq * 8) + pad256(nq * sizeof(HostSeg)) +
                             pad256(tiles.size() * sizeof(HostTile)) + pad256(total * 4) * 2 + 2 * pad256(nq * kstride * 8) +
                             pad256(nq * k * 4) * 2 + pad256(inv_bytes) + screen_bytes + 4096;
    const size_t pin_bytes = pad256(nq * (size_t)ds->dims * 4) + pad256(nq * sizeof(HostSeg)) +
                             pad256(tiles.size() * sizeof(HostTile)) + pad256(total * 4) + pad256(nq * k * 4) * 2 + 4096;
    AH_TRY(ctx->ensure_device(dev_bytes));
    AH_TRY(ctx->ensure_pinned(pin_bytes));
    Carver dev(ctx->d_scratch), pin(ctx->h_pinned);
    float *d_qf32 = dev.take<float>(nq * (size_t)ds->dims);
    uint8_t *d_qvecs = dev.take<uint8_t>(nq * qstride);
    float *d_qhdrs = dev.take<float>(nq * 2);
    HostSeg *d_segs = dev.take<HostSeg>(nq);
    HostTile *d_tiles = dev.take<HostTile>(tiles.size());
    uint32_t *d_ids = dev.take<uint32_t>(total);
    float *d_dist = dev.take<float>(total);
    uint64_t *d_ka = dev.take<uint64_t>(nq * kstride);
    uint64_t *d_kb = dev.take<uint64_t>(nq * kstride);
    uint32_t *d_oi = dev.take<uint32_t>(nq * k);
    float *d_od = dev.take<float>(nq * k);
    uint32_t *d_err = dev.take<uint32_t>(16);  // [error bits][counters of the screened selection]
    uint32_t *d_inv = inv_bytes ? dev.take<uint32_t>(inv_bytes / 4) : nullptr;
    uint16_t *d_q16 = screened ? dev.take<uint16_t>(nq * (size_t)ds->hpitch) : nullptr;
    float4 *d_qstats = screened ? dev.take<float4>(nq) : nullptr;
    float *d_aux = screened ? dev.take<float>(total) : nullptr;
    int8_t *d_q8 = screened8 ? dev.take<int8_t>(nq * (size_t)ds->pitch8 * 2) : nullptr;
    float4 *d_q8stats = screened8 ? dev.take<float4>(nq) : nullptr;
    float *d_aux8 = screened8 ? dev.take<float>(total) : nullptr;
    float *h_q = pin.take<float>(nq * (size_t)ds->dims);
    HostSeg *h_segs = pin.take<HostSeg>(nq);
    HostTile *h_tiles = pin.take<HostTile>(tiles.size());
    uint32_t *h_ids = pin.take<uint32_t>(total);
    uint32_t *h_oi = pin.take<uint32_t>(nq * k);
    float *h_od = pin.take<float>(nq * k);
    uint32_t *h_err = pin.take<uint32_t>(16);  // [error bits][the selection's counters: words 9, 10 of search.hip's SearchStatSlot]
    // More survivors than the selection holds (bit 3) on the int8 stage is not a reason to leave the screen: the same lists once
    // more on the binary16 rows.  A dataset where that keeps happening (candidates closer together than the int8 error) stops
    // trying: `rerank8_off` after 8 such submissions out of the last <= 64.
    bool retried8 = false;
    auto note_screened = [&](const uint32_t *words) {  // ah_rerank_stats: how the screen went (always kept: three additions)
        std::lock_guard<std::mutex> lk(ds->mu);
        ds->rr_stats.queries_screened += words[9];
        ds->rr_stats.survivors += words[10];
        ds->rr_stats.chunks_int8 += screened8 && !retried8 ? 1 : 0;
        ds->rr_stats.chunks_int8_retried += retried8 ? 1 : 0;
    };
    auto retry_on_binary16 = [&](uint32_t err_bits) -> int {
        if (!screened8 || (err_bits & ~1u) != 8u) return AH_OK;
        retried8 = true;
        const uint32_t fails = ds->rerank8_fails.fetch_add(1, std::memory_order_relaxed) + 1;
        if (fails >= 8) ds->rerank8_off.store(true, std::memory_order_relaxed);
        AH_HIP(hipMemsetAsync(d_err, 0, 64, ctx->stream));
        AH_TRY(launch_rerank_screened(ds, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, 0u, (uint32_t)tiles.size(), tc, d_ids,
                                      d_dist, d_aux, d_q16, d_qstats, (uint32_t)k, d_oi, d_od, d_err, ctx->stream, true, true));
        AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
        AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
        AH_HIP(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, ctx->stream));
        AH_HIP(hipStreamSynchronize(ctx->stream));
        return AH_OK;
    };
    memcpy(h_q, queries, nq * (size_t)ds->dims * 4);
    memcpy(h_segs, segs.data(), nq * sizeof(HostSeg));
    if (!tiles.empty()) memcpy(h_tiles, tiles.data(), tiles.size() * sizeof(HostTile));
    hipStream_t s = ctx->stream;
    probe.lap(&RerankProbe::prep);
    probe.first_enqueue(s);
    AH_HIP(hipMemcpyAsync(d_qf32, h_q, nq * (size_t)ds->dims * 4, hipMemcpyHostToDevice, s));
    AH_HIP(hipMemcpyAsync(d_segs, h_segs, nq * sizeof(HostSeg), hipMemcpyHostToDevice, s));
    if (!tiles.empty()) AH_HIP(hipMemcpyAsync(d_tiles, h_tiles, tiles.size() * sizeof(HostTile), hipMemcpyHostToDevice, s));
    AH_HIP(hipMemsetAsync(d_err, 0, 64, s));
    // candidate ids: tens of MB for a big submission.  Slices of 2M ids are copied into the pinned buffer by several
    // cores (like the staging gather) and sent right away, so the DMA of one slice overlaps the host copy of the next.
    const uint64_t n_groups = (uint64_t)std::max<long long>(1, tun(TUN_RERANK_GROUPS));
    const bool pipelined = screened && n_groups > 1 && total >= (512u << 10);
    if (pipelined && !ctx->copy_stream && create_copy_stream(&ctx->copy_stream) != hipSuccess) {
        (void)hipGetLastError();
        ctx->copy_stream = nullptr;
    }
    if (pipelined && ctx->copy_stream) {
        // The screened path takes the lists group by group: while the screen kernel of group g runs on the context's stream,
        // the host copies the ids of group g + 1 into the pinned buffer and the copy stream sends them (review item 6: the
        // upload of a submission used to run to its end before the first kernel started).
        AH_TRY(launch_prepare_queries_only(ds->view(), d_qf32, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, s));
        const uint64_t group_ids = (total + n_groups - 1) / n_groups;
        size_t qa = 0;
        uint32_t ta = 0;
        bool first = true;
        while (qa < nq) {
            size_t qb = qa + 1;
            while (qb < nq && segs[qb].off + segs[qb].n - segs[qa].off <= group_ids) qb++;
            const uint64_t lo = segs[qa].off, len = segs[qb - 1].off + segs[qb - 1].n - lo;
            uint32_t tb = ta;
            while (tb < tiles.size() && tiles[tb].query < qb) tb++;
            if (len) {
                probe.lap(&RerankProbe::enqueue);
                parallel_rows(len, 4, [&](size_t a, size_t b) { memcpy(h_ids + lo + a, ids + base + lo + a, (b - a) * 4); });
                probe.lap(&RerankProbe::ids);
                AH_HIP(hipMemcpyAsync(d_ids + lo, h_ids + lo, len * 4, hipMemcpyHostToDevice, ctx->copy_stream));
                AH_HIP(hipEventRecord(ctx->ev0, ctx->copy_stream));
                AH_HIP(hipStreamWaitEvent(s, ctx->ev0, 0));
            }
            AH_TRY(launch_rerank_screened(ds, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, ta, tb - ta, tc, d_ids, d_dist,
                                          d_aux, d_q16, d_qstats, (uint32_t)k, d_oi, d_od, d_err, s, first, qb == nq, d_q8, d_q8stats,
                                          d_aux8));
            first = false;
            qa = qb;
            ta = tb;
        }
        AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
        AH_RR_SYNC(s);
        AH_TRY(retry_on_binary16(*h_err));
        if ((*h_err & ~1u) == 0) {
            note_screened(h_err);
            AH_TRY(check_err_flags(*h_err, true));
            memcpy(out_ids, h_oi, nq * k * 4);
            memcpy(out_distances, h_od, nq * k * 4);
            return AH_OK;
        }
        // a non-finite value, or more survivors than the selection holds: the exact path, from the prepared queries
        AH_HIP(hipMemsetAsync(d_err, 0, 64, s));
        AH_TRY(launch_rerank_batch_prepared(ds->view(), (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, (uint32_t)tiles.size(),
                                            d_ids, d_dist, d_ka, d_kb, kstride, max_n, (uint32_t)k, max_rounds, d_oi, d_od, d_err, s,
                                            total, d_inv));
        AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
        AH_RR_SYNC(s);
        AH_TRY(check_err_flags(*h_err, true));
        memcpy(out_ids, h_oi, nq * k * 4);
        memcpy(out_distances, h_od, nq * k * 4);
        return AH_OK;
    }
    for (uint64_t lo = 0; lo < total; lo += (2u << 20)) {
        const uint64_t len = std::min<uint64_t>(2u << 20, total - lo);
        probe.lap(&RerankProbe::enqueue);
        parallel_rows(len, 4, [&](size_t a, size_t b) { memcpy(h_ids + lo + a, ids + base + lo + a, (b - a) * 4); });
        probe.lap(&RerankProbe::ids);
        AH_HIP(hipMemcpyAsync(d_ids + lo, h_ids + lo, len * 4, hipMemcpyHostToDevice, s));
    }
    if (screened) {
        AH_TRY(launch_prepare_queries_only(ds->view(), d_qf32, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, s));
        AH_TRY(launch_rerank_screened(ds, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, 0u, (uint32_t)tiles.size(), tc, d_ids,
                                      d_dist, d_aux, d_q16, d_qstats, (uint32_t)k, d_oi, d_od, d_err, s, true, true, d_q8, d_q8stats, d_aux8));
        AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, s));
        AH_HIP(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
        AH_RR_SYNC(s);
        AH_TRY(retry_on_binary16(*h_err));
        if ((*h_err & ~1u) == 0) {
            note_screened(h_err);
            AH_TRY(check_err_flags(*h_err, true));
            memcpy(out_ids, h_oi, nq * k * 4);
            memcpy(out_distances, h_od, nq * k * 4);
            return AH_OK;
        }
        // a non-finite value, or more survivors than the selection holds: the exact path, from the prepared queries
        AH_HIP(hipMemsetAsync(d_err, 0, 64, s));
        AH_TRY(launch_rerank_batch_prepared(ds->view(), (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles, (uint32_t)tiles.size(),
                                            d_ids, d_dist, d_ka, d_kb, kstride, max_n, (uint32_t)k, max_rounds, d_oi, d_od, d_err, s,
                                            total, d_inv));
    } else {
        AH_TRY(launch_rerank_batch(ds->view(), d_qf32, (uint32_t)nq, d_qvecs, qstride, d_qhdrs, d_segs, d_tiles,
                                   (uint32_t)tiles.size(), d_ids, d_dist, d_ka, d_kb, kstride, max_n, (uint32_t)k, max_rounds,
                                   d_oi, d_od, d_err, s, total, d_inv));
    }
    AH_HIP(hipMemcpyAsync(h_oi, d_oi, nq * k * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipMemcpyAsync(h_od, d_od, nq * k * 4, hipMemcpyDeviceToHost, s));
    AH_HIP(hipMemcpyAsync(h_err, d_err, 64, hipMemcpyDeviceToHost, s));
    AH_RR_SYNC(s);
    AH_TRY(check_err_flags(*h_err, true));
    memcpy(out_ids, h_oi, nq * k * 4);
    memcpy(out_distances, h_od, nq * k * 4);
    return AH_OK;
}

int ah_rerank_batch(ah_dataset *ds, const float *queries, size_t n_queries, const uint32_t *ids,
                    const uint64_t *offsets, size_t k, uint32_t *out_ids, float *out_distances, uint32_t *out_counts) {
    AH_GUARDED("ah_rerank_batch")
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(queries && ids && offsets && out_ids && out_distances && out_counts, AH_ERR_INVALID_ARGUMENT,
               "NULL argument");
    if (n_queries == 0) return AH_OK;
    AH_REQUIRE(k > 0, AH_ERR_INVALID_ARGUMENT, "k must be > 0");
    for (size_t q = 0; q < n_queries; q++)
        AH_REQUIRE(offsets[q + 1] >= offsets[q], AH_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    if (!batch_supported((uint32_t)std::min<size_t>(k, 0xFFFFFFFFu))) {
        // rare: count in the thousands.  One query at a time through the single-query path.
        for (size_t q = 0; q < n_queries; q++) {
            for (size_t t = 0; t < k; t++) {
                out_ids[q * k + t] = 0xFFFFFFFFu;
                uint32_t nan_bits = 0xFFFFFFFFu;
                memcpy(&out_distances[q * k + t], &nan_bits, 4);
            }
            size_t got = 0;
            const size_t nq_items = offsets[q + 1] - offsets[q];
            out_counts[q] = 0;
            if (nq_items == 0) continue;
            AH_TRY(ah_rerank_by_vector(ds, queries + q * (size_t)ds->dims, ids + offsets[q], nq_items, k, out_ids + q * k,
                                       out_distances + q * k, &got));
            out_counts[q] = (uint32_t)got;
        }
        return AH_OK;
    }
    AH_LEASE(ds, ctx);
    // bound the scratch: sub-batches of queries whose candidate lists total <= 64M ids and <= 1024 queries
    size_t q0 = 0;
    while (q0 < n_queries) {
        size_t q1 = q0 + 1;
        while (q1 < n_queries && q1 - q0 < 1024 && offsets[q1 + 1] - offsets[q0] <= (64ull << 20)) q1++;
        AH_TRY(rerank_batch_chunk(ds, ctx, queries + q0 * (size_t)ds->dims, q1 - q0, ids, offsets + q0, k, out_ids + q0 * k,
                                  out_distances + q0 * k, out_counts + q0));
        q0 = q1;
    }
    return AH_OK;
    AH_GUARDED_END
}

int ah_dataset_rerank_stats(ah_dataset *ds, ah_rerank_stats *out, int reset) {
    AH_GUARDED("ah_dataset_rerank_stats")
    AH_REQUIRE(ds, AH_ERR_INVALID_ARGUMENT, "dataset is NULL");
    std::lock_guard<std::mutex> lk(ds->mu);
    if (out) *out = ds->rr_stats;
    if (reset) ds->rr_stats = ah_rerank_stats{};
    return AH_OK;
    AH_GUARDED_END
}

// ---------------------------------------------------------------------------------------------
// build side: single-node entry points (the incremental paths of arroy call these per node)
// ---------------------------------------------------------------------------------------------
int ah_split_sides(ah_dataset *ds, const void *normal_vector, const void *normal_header, const uint32_t *sorted_ids,
                   size_t n, uint8_t *side_bits, uint64_t *out_n_left, float *out_margins) {
    AH_GUARDED("ah_split_sides")
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(normal_vector && normal_header && side_bits && out_n_left, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!sorted_ids) n = ds->n;
    *out_n_left = 0;
    if (n == 0) return AH_OK;
    AH_LEASE(ds, ctx);
    const size_t vs = ah_vector_size(ds->metric, ds->dims), hs = ah_header_size(ds->metric);
    const size_t nbytes = (n + 7) / 8;
    const size_t nwords = (n + 31) / 32;
    AH_TRY(ctx->ensure_device(pad256(ds->row_bytes()) + 256 + pad256(n * 4) * 2 + pad256(nwords * 4) + 1024));
    AH_TRY(ctx->ensure_pinned(pad256(ds->row_bytes()) + 256 + pad256(n * 4) * 2 + pad256(nwords * 4) + 1024));
    Carver dev(ctx->d_scratch), pin(ctx->h_pinned);
    uint8_t *d_nv = dev.take<uint8_t>(ds->row_bytes());
    float *d_nh = dev.take<float>(2);
    unsigned long long *d_left = dev.take<unsigned long long>(1);
    uint32_t *d_err = dev.take<uint32_t>(1);
    uint32_t *d_ids = sorted_ids ? dev.take<uint32_t>(n) : nullptr;
    uint8_t *d_bits = dev.take<uint8_t>(nwords * 4);
    float *d_marg = out_margins ? dev.take<float>(n) : nullptr;
    uint8_t *h_nv = pin.take<uint8_t>(ds->row_bytes());
    float *h_nh = pin.take<float>(2);
    uint32_t *h_ids = sorted_ids ? pin.take<uint32_t>(n) : nullptr;
    uint8_t *h_bits = pin.take<uint8_t>(nwords * 4);
    float *h_marg = out_margins ? pin.take<float>(n) : nullptr;
    unsigned long long *h_left = pin.take<unsigned long long>(1);
    uint32_t *h_err = pin.take<uint32_t>(1);
    memset(h_nv, 0, ds->row_bytes());
    memcpy(h_nv, normal_vector, vs);
    h_nh[0] = h_nh[1] = 0.0f;
    memcpy(h_nh, normal_header, hs);
    AH_HIP(hipMemcpyAsync(d_nv, h_nv, ds->row_bytes(), hipMemcpyHostToDevice, ctx->stream));
    AH_HIP(hipMemcpyAsync(d_nh, h_nh, 8, hipMemcpyHostToDevice, ctx->stream));
    if (sorted_ids) {
        memcpy(h_ids, sorted_ids, n * 4);
        AH_HIP(hipMemcpyAsync(d_ids, h_ids, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    AH_HIP(hipMemsetAsync(d_left, 0, 8, ctx->stream));
    AH_HIP(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    AH_HIP(hipMemsetAsync(d_bits, 0, nwords * 4, ctx->stream));
    AH_TRY(launch_split_sides(ds->view(), d_nv, d_nh, d_ids, n, d_bits, d_left, d_marg, d_err, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_bits, d_bits, nwords * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_margins) AH_HIP(hipMemcpyAsync(h_marg, d_marg, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_left, d_left, 8, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    AH_TRY(check_err_flags(*h_err, false));
    memcpy(side_bits, h_bits, nbytes);
    if (out_margins) memcpy(out_margins, h_marg, n * 4);
    *out_n_left = *h_left;
    return AH_OK;
    AH_GUARDED_END
}

// `D::margin(&normal, query_leaf)` for many stored normals at once: the dataset's ROWS are split-plane normals
// (header = the normal's header), the broadcast operand is the query leaf.  src/reader.rs:366-369.
int ah_margins(ah_dataset *normals, const void *leaf_vector, const void *leaf_header, const uint32_t *item_ids,
               size_t n, float *out_margins) {
    AH_GUARDED("ah_margins")
    ah_dataset *ds = normals;
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(leaf_vector && leaf_header && (out_margins || n == 0), AH_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!item_ids) n = ds->n;
    if (n == 0) return AH_OK;
    AH_LEASE(ds, ctx);
    const size_t vs = ah_vector_size(ds->metric, ds->dims), hs = ah_header_size(ds->metric);
    AH_TRY(ctx->ensure_device(pad256(ds->row_bytes()) + 1024 + pad256(n * 4) * 2));
    AH_TRY(ctx->ensure_pinned(pad256(ds->row_bytes()) + 1024 + pad256(n * 4) * 2));
    Carver dev(ctx->d_scratch), pin(ctx->h_pinned);
    uint8_t *d_v = dev.take<uint8_t>(ds->row_bytes());
    float *d_h = dev.take<float>(2);
    uint32_t *d_err = dev.take<uint32_t>(1);
    uint32_t *d_ids = item_ids ? dev.take<uint32_t>(n) : nullptr;
    float *d_m = dev.take<float>(n);
    uint8_t *h_v = pin.take<uint8_t>(ds->row_bytes());
    float *h_h = pin.take<float>(2);
    uint32_t *h_ids = item_ids ? pin.take<uint32_t>(n) : nullptr;
    float *h_m = pin.take<float>(n);
    uint32_t *h_err = pin.take<uint32_t>(1);
    memset(h_v, 0, ds->row_bytes());
    memcpy(h_v, leaf_vector, vs);
    h_h[0] = h_h[1] = 0.0f;
    memcpy(h_h, leaf_header, hs);
    AH_HIP(hipMemcpyAsync(d_v, h_v, ds->row_bytes(), hipMemcpyHostToDevice, ctx->stream));
    AH_HIP(hipMemcpyAsync(d_h, h_h, 8, hipMemcpyHostToDevice, ctx->stream));
    if (item_ids) {
        memcpy(h_ids, item_ids, n * 4);
        AH_HIP(hipMemcpyAsync(d_ids, h_ids, n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    AH_HIP(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    AH_TRY(launch_split_sides(ds->view(), d_v, d_h, d_ids, n, nullptr, nullptr, d_m, d_err, ctx->stream, 1));
    AH_HIP(hipMemcpyAsync(h_m, d_m, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    AH_TRY(check_err_flags(*h_err, false));
    memcpy(out_margins, h_m, n * 4);
    return AH_OK;
    AH_GUARDED_END
}

int ah_create_split(ah_dataset *ds, const uint32_t sample_ids[AH_SPLIT_SAMPLES], void *out_normal_vector,
                    void *out_normal_header) {
    AH_GUARDED("ah_create_split")
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(sample_ids && out_normal_vector && out_normal_header, AH_ERR_INVALID_ARGUMENT, "NULL argument");
    AH_REQUIRE(ds->metric != AH_DOT_PRODUCT || ds->dot_preprocessed, AH_ERR_NEED_PREPROCESS,
               "DotProduct needs ah_preprocess_dot before splits");
    uint32_t rows[AH_SPLIT_SAMPLES];
    for (int i = 0; i < AH_SPLIT_SAMPLES; i++) AH_TRY(host_row_of_id(ds, sample_ids[i], &rows[i]));
    AH_LEASE(ds, ctx);
    const size_t vs = ah_vector_size(ds->metric, ds->dims), hs = ah_header_size(ds->metric);
    AH_TRY(ctx->ensure_device(pad256(ds->row_bytes()) + 1024));
    AH_TRY(ctx->ensure_pinned(pad256(ds->row_bytes()) + 1024));
    Carver dev(ctx->d_scratch), pin(ctx->h_pinned);
    uint32_t *d_rows = dev.take<uint32_t>(AH_SPLIT_SAMPLES);
    uint8_t *d_nv = dev.take<uint8_t>(ds->row_bytes());
    float *d_nh = dev.take<float>(2);
    uint32_t *h_rows = pin.take<uint32_t>(AH_SPLIT_SAMPLES);
    uint8_t *h_nv = pin.take<uint8_t>(ds->row_bytes());
    float *h_nh = pin.take<float>(2);
    memcpy(h_rows, rows, sizeof rows);
    AH_HIP(hipMemcpyAsync(d_rows, h_rows, sizeof rows, hipMemcpyHostToDevice, ctx->stream));
    AH_TRY(launch_create_split(ds->view(), d_rows, d_nv, d_nh, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_nv, d_nv, ds->row_bytes(), hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipMemcpyAsync(h_nh, d_nh, 8, hipMemcpyDeviceToHost, ctx->stream));
    AH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out_normal_vector, h_nv, vs);
    memcpy(out_normal_header, h_nh, hs);
    return AH_OK;
    AH_GUARDED_END
}

// ---------------------------------------------------------------------------------------------
// measurement helpers
// ---------------------------------------------------------------------------------------------
int ah_bench_scan(ah_dataset *ds, uint32_t query_item, uint64_t n, uint32_t iterations, float *out,
                  double *out_ms_total) {
    AH_GUARDED("ah_bench_scan")
    AH_NEED_FINALIZED(ds);
    AH_REQUIRE(out_ms_total && iterations > 0 && n > 0 && n <= ds->n, AH_ERR_INVALID_ARGUMENT, "bad arguments");
    AH_LEASE(ds, ctx);
    QueryBufs qb;
    Carver dev(nullptr), pin(nullptr);
    AH_TRY(stage_query(ds, ctx, nullptr, &query_item, pad256(n * 4), 256, &qb, &dev, &pin));
    float *d_out = dev.take<float>(n);
    DataView dv = ds->view();
    AH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (uint32_t it = 0; it < iterations; it++)
        AH_TRY(launch_distances(dv, qb.d_qvec, qb.d_qhdr, nullptr, n, d_out, qb.d_err, ctx->stream));
    AH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    AH_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.0f;
    AH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *out_ms_total = ms;
    if (out) AH_HIP(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
    return AH_OK;
    AH_GUARDED_END
}

int ah_bench_memcpy(int device, uint64_t bytes, uint32_t iterations, double *out_ms_total) {
    AH_GUARDED("ah_bench_memcpy")
    AH_REQUIRE(out_ms_total && iterations > 0 && bytes > 0, AH_ERR_INVALID_ARGUMENT, "bad arguments");
    AH_HIP(hipSetDevice(device));
    DevMem a, b;
    Context c;  // stream + two events, destroyed on every path
    struct Guard {
        Context &c;
        ~Guard() { c.destroy(); }
    } guard{c};
    AH_HIP(dev_malloc(&a.p, bytes));
    AH_HIP(dev_malloc(&b.p, bytes));
    AH_HIP(hipStreamCreate(&c.stream));
    AH_HIP(hipEventCreate(&c.ev0));
    AH_HIP(hipEventCreate(&c.ev1));
    AH_HIP(hipMemsetAsync(a.p, 1, bytes, c.stream));
    AH_HIP(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, c.stream));
    AH_HIP(hipEventRecord(c.ev0, c.stream));
    for (uint32_t i = 0; i < iterations; i++) AH_HIP(hipMemcpyAsync(b.p, a.p, bytes, hipMemcpyDeviceToDevice, c.stream));
    AH_HIP(hipEventRecord(c.ev1, c.stream));
    AH_HIP(hipEventSynchronize(c.ev1));
    float ms = 0.0f;
    AH_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *out_ms_total = ms;
    return AH_OK;
    AH_GUARDED_END
}

int ah_bench_read(int device, uint64_t bytes, uint32_t iterations, double *out_ms_total) {
    AH_GUARDED("ah_bench_read")
    AH_REQUIRE(out_ms_total && iterations > 0 && bytes >= 4096, AH_ERR_INVALID_ARGUMENT, "bad arguments");
    AH_HIP(hipSetDevice(device));
    DevMem a, sink;
    Context c;
    struct Guard {
        Context &c;
        ~Guard() { c.destroy(); }
    } guard{c};
    AH_HIP(dev_malloc(&a.p, bytes));
    AH_HIP(dev_malloc(&sink.p, 8));
    AH_HIP(hipStreamCreate(&c.stream));
    AH_HIP(hipEventCreate(&c.ev0));
    AH_HIP(hipEventCreate(&c.ev1));
    AH_HIP(hipMemsetAsync(a.p, 1, bytes, c.stream));
    AH_HIP(hipMemsetAsync(sink.p, 0, 8, c.stream));
    AH_TRY(launch_bench_read(a.p, bytes, sink.as<unsigned long long>(), c.stream));
    AH_HIP(hipEventRecord(c.ev0, c.stream));
    for (uint32_t i = 0; i < iterations; i++) AH_TRY(launch_bench_read(a.p, bytes, sink.as<unsigned long long>(), c.stream));
    AH_HIP(hipEventRecord(c.ev1, c.stream));
    AH_HIP(hipEventSynchronize(c.ev1));
    float ms = 0.0f;
    AH_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *out_ms_total = ms;
    return AH_OK;
    AH_GUARDED_END
}

}  // extern "C"
