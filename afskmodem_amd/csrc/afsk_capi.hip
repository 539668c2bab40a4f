// afsk_capi.hip -- extern "C" boundary of libafsk_amd.so (see include/afsk_amd.h).
// Host-side argument checks, error strings and kernel launches; no torch types.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/afsk_amd.h"
#include "afsk_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(AFSK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(AFSK_E_NO_DEVICE, "no HIP device visible: libafsk_amd has no CPU fallback");
    }
    return AFSK_OK;
}


// ---- NUMA: keep the host side of the PCIe traffic on the socket the GPU hangs off ---------------------------
// The GPU boxes are two-socket hosts (2 x 64 cores, 4 GPUs per socket).  A staging buffer pinned on the other
// socket is read by the GPU across the inter-socket link (43 instead of 57 GB/s measured), and pool threads on
// the other socket copy page-cache pages into it at half the rate (read 100 ms instead of 55 ms of CPU per 393 MB).
// So: the pinned staging buffers are allocated while the calling thread is (temporarily) confined to the CPUs of
// the device's NUMA node, and the I/O pool's threads stay on those CPUs.  AFSK_NUMA_BIND=0 turns both off;
// anything that cannot be found out (no such attribute, no sysfs) means "no binding", never an error.
bool device_node_cpus(cpu_set_t* out) {
    static const bool enabled = [] { const char* e = std::getenv("AFSK_NUMA_BIND"); return !e || std::atoi(e) != 0; }();
    if (!enabled) return false;
    int dev = 0, node = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    // one answer per device and process (two sysfs reads otherwise, on every ingest call); a process that
    // drives GPUs on both sockets gets each device's own node
    constexpr int kMaxCachedDevices = 64;
    static std::mutex cache_mu;
    static bool known[kMaxCachedDevices], known_ok[kMaxCachedDevices];
    static cpu_set_t known_cpus[kMaxCachedDevices];
    std::lock_guard<std::mutex> cache_lock(cache_mu);
    const bool cacheable = dev >= 0 && dev < kMaxCachedDevices;
    if (cacheable && known[dev]) { if (known_ok[dev]) *out = known_cpus[dev]; return known_ok[dev]; }
    if (cacheable) { known[dev] = true; known_ok[dev] = false; }
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, dev) != hipSuccess || node < 0) {
        (void)hipGetLastError();
        node = -1;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
        for (char* c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
        if (FILE* f = std::fopen(path.c_str(), "r")) {
            if (std::fscanf(f, "%d", &node) != 1) node = -1;
            std::fclose(f);
        }
    }
    if (node < 0) return false;
    char path[96];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { std::fclose(f); return false; }
    int a = 0, b = 0;
    for (;;) {                                             // "0-63,128-191"
        if (std::fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = std::fgetc(f);
        if (c == '-') { if (std::fscanf(f, "%d", &b) != 1) break; c = std::fgetc(f); }
        for (int k = a; k <= b && k < CPU_SETSIZE; k++)
            if (CPU_ISSET(k, &allowed)) CPU_SET(k, &want);
        if (c != ',') break;
    }
    std::fclose(f);
    if (CPU_COUNT(&want) == 0) return false;
    *out = want;
    if (cacheable) { known_cpus[dev] = want; known_ok[dev] = true; }
    return true;
}

// confines the calling thread to the device's NUMA node for its lifetime (allocations made meanwhile are
// first-touched there), then restores the thread's affinity
class NodeBinder {
public:
    NodeBinder() {
        cpu_set_t want;
        if (!device_node_cpus(&want)) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof old_, &old_) != 0) return;
        bound_ = pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
    }
    ~NodeBinder() {
        if (bound_) (void)pthread_setaffinity_np(pthread_self(), sizeof old_, &old_);
    }
private:
    cpu_set_t old_;
    bool bound_ = false;
};

// Device scratch of the host-buffer entry, kept between calls: hipMalloc of a few hundred MB
// costs tens of ms (75 ms for 393 MB measured, tools/h2d_probe), far more than the transfer.
// One cached allocation per process, grown on demand; a concurrent caller that finds it busy
// falls back to a private hipMalloc/hipFree.  afsk_host_scratch_release() frees it.
constexpr size_t kStageBytes = (size_t)32 << 20;   // pinned staging window of the gather entry

struct ScratchCache {
    std::mutex mu;
    char* ptr = nullptr;
    size_t cap = 0;
    int device = -1;
    char* stage[2] = {nullptr, nullptr};            // pinned host windows (afsk_demod_streams_host)
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    // afsk_wav_ingest's staging ring: chunks 0 and 1 ARE stage[0] / stage[1]; 2 and 3 are allocated the first
    // time a ring of more than 64 MiB is asked for; one event per ring slot
    char* ring_extra[2] = {nullptr, nullptr};
    hipEvent_t ring_sent[16] = {};
};
constexpr size_t kRingChunks = 4;
constexpr int kMaxRingSlots = 16;
ScratchCache g_scratch;

class ScratchLease {
public:
    ~ScratchLease() {
        if (private_ptr_) (void)hipFree(private_ptr_);
        if (locked_) g_scratch.mu.unlock();
    }
    // block = wait for the shared cache instead of falling back to a private allocation
    hipError_t acquire(size_t bytes, char** out, bool block = false) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (block) g_scratch.mu.lock();
        if (block || g_scratch.mu.try_lock()) {
            locked_ = true;
            if (g_scratch.cap < bytes || g_scratch.device != dev) {
                if (g_scratch.ptr) (void)hipFree(g_scratch.ptr);
                g_scratch.ptr = nullptr;
                g_scratch.cap = 0;
                if (g_scratch.device != dev) {
                    // events belong to the device they were created on: a cache that moves to
                    // another device drops them (staging() recreates them there); the pinned
                    // windows are hipHostMallocPortable and stay
                    for (int k = 0; k < 2; k++)
                        if (g_scratch.stage_free[k]) {
                            (void)hipEventDestroy(g_scratch.stage_free[k]);
                            g_scratch.stage_free[k] = nullptr;
                        }
                    for (hipEvent_t& ev : g_scratch.ring_sent)
                        if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
                }
                const size_t want = (bytes + (bytes >> 3) + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
                e = hipMalloc((void**)&g_scratch.ptr, want);
                if (e != hipSuccess) { g_scratch.ptr = nullptr; return e; }
                g_scratch.cap = want;
                g_scratch.device = dev;
            }
            *out = g_scratch.ptr;
            return hipSuccess;
        }
        e = hipMalloc((void**)&private_ptr_, bytes);
        if (e != hipSuccess) { private_ptr_ = nullptr; return e; }
        *out = private_ptr_;
        return hipSuccess;
    }
    // the two pinned staging windows (only with a blocking lease, which owns the cache)
    hipError_t staging(char** s0, char** s1, hipEvent_t* e0, hipEvent_t* e1) {
        for (int k = 0; k < 2; k++) {
            if (!g_scratch.stage[k]) {
                NodeBinder on_the_devices_socket;
                hipError_t e = hipHostMalloc((void**)&g_scratch.stage[k], kStageBytes, hipHostMallocPortable);
                if (e != hipSuccess) { g_scratch.stage[k] = nullptr; return e; }
            }
            if (!g_scratch.stage_free[k]) {
                hipError_t e = hipEventCreateWithFlags(&g_scratch.stage_free[k], hipEventDisableTiming);
                if (e != hipSuccess) { g_scratch.stage_free[k] = nullptr; return e; }
            }
        }
        *s0 = g_scratch.stage[0]; *s1 = g_scratch.stage[1];
        *e0 = g_scratch.stage_free[0]; *e1 = g_scratch.stage_free[1];
        return hipSuccess;
    }
    // afsk_wav_ingest's ring of `slots` staging buffers of `window` bytes each (window divides 32 MiB,
    // slots * window <= 128 MiB) and one event per slot; only with a blocking lease
    hipError_t ring(size_t window, int slots, char** out_stage, hipEvent_t* out_sent) {
        char* s0; char* s1; hipEvent_t e0, e1;
        hipError_t e = staging(&s0, &s1, &e0, &e1);
        if (e != hipSuccess) return e;
        char* chunk[kRingChunks] = {s0, s1, nullptr, nullptr};
        const size_t need = (size_t)slots * window;
        for (size_t c = 2; c < kRingChunks && c * kStageBytes < need; c++) {
            if (!g_scratch.ring_extra[c - 2]) {
                NodeBinder on_the_devices_socket;
                e = hipHostMalloc((void**)&g_scratch.ring_extra[c - 2], kStageBytes, hipHostMallocPortable);
                if (e != hipSuccess) { g_scratch.ring_extra[c - 2] = nullptr; return e; }
            }
        }
        chunk[2] = g_scratch.ring_extra[0]; chunk[3] = g_scratch.ring_extra[1];
        for (int k = 0; k < slots; k++) {
            const size_t o = (size_t)k * window;
            out_stage[k] = chunk[o / kStageBytes] + o % kStageBytes;
            if (!g_scratch.ring_sent[k]) {
                e = hipEventCreateWithFlags(&g_scratch.ring_sent[k], hipEventDisableTiming);
                if (e != hipSuccess) { g_scratch.ring_sent[k] = nullptr; return e; }
            }
            out_sent[k] = g_scratch.ring_sent[k];
        }
        return hipSuccess;
    }
private:
    bool locked_ = false;
    char* private_ptr_ = nullptr;
};

// The host-buffer entries run on a private NON-BLOCKING stream (one per calling thread and
// device, created on first use): on the NULL stream every call would synchronise implicitly with
// all other streams of the process -- the caller's own streams and other threads' calls.
struct ThreadStream {
    hipStream_t s = nullptr;
    int device = -1;
    ~ThreadStream() {
        if (s) (void)hipStreamDestroy(s);
    }
    hipError_t get(hipStream_t* out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (s && device != dev) { (void)hipStreamDestroy(s); s = nullptr; }
        if (!s) {
            e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            if (e != hipSuccess) { s = nullptr; return e; }
            device = dev;
        }
        *out = s;
        return hipSuccess;
    }
};
thread_local ThreadStream g_thread_stream;
thread_local ThreadStream g_thread_stream2;      // afsk_wav_ingest alternates its H2D copies between the two

struct CopyJob { char* dst; const char* src; size_t bytes; };

// memcpy a list of pieces, split over the I/O pool's threads when it is worth it (one core moves ~30 GB/s into
// pinned memory from a hot source, a few GB/s from cold pages; the PCIe link takes ~56 GB/s: tools/h2d_probe).
// (r4: the persistent pool instead of std::threads spawned per staging window -- ~100 thread creations per call of
// the gather entry at 4096 streams.)
unsigned usable_cpus();
void pool_run(size_t n, unsigned width, const std::function<void(size_t)>& body);
void parallel_copy(const std::vector<CopyJob>& jobs, size_t total) {
    static const unsigned max_threads = [] {
        const char* e = std::getenv("AFSK_COPY_THREADS");
        return e ? (unsigned)std::max(1, std::atoi(e)) : std::min(16u, std::max(1u, usable_cpus()));
    }();
    unsigned nt = total >= ((size_t)4 << 20) ? max_threads : 1u;
    if (nt <= 1) {
        for (const CopyJob& j : jobs) std::memcpy(j.dst, j.src, j.bytes);
        return;
    }
    // split the byte range into nt contiguous shares
    const size_t share = (total + nt - 1) / nt;
    std::vector<std::vector<CopyJob>> parts;
    size_t ji = 0, joff = 0;                       // cursor: job index, byte offset inside it
    while (ji < jobs.size()) {
        std::vector<CopyJob> mine;
        size_t left = share;
        while (left > 0 && ji < jobs.size()) {
            const size_t take = std::min(left, jobs[ji].bytes - joff);
            if (take > 0) mine.push_back({jobs[ji].dst + joff, jobs[ji].src + joff, take});
            left -= take; joff += take;
            if (joff == jobs[ji].bytes) { ji++; joff = 0; }
        }
        parts.push_back(std::move(mine));
    }
    pool_run(parts.size(), (unsigned)parts.size(), [&](size_t k) {
        for (const CopyJob& j : parts[k]) std::memcpy(j.dst, j.src, j.bytes);
    });
}

// A small persistent pool for the file-ingest entries: spawning 15 threads per staging window cost
// more than filling the window.  run(n, width, body) executes body(i) for i in [0, n) on up to
// `width` threads (the caller is one of them) and returns when all are done; one job at a time.
class IoPool {
public:
    // from the next job on the pool's threads stay on these CPUs (the device's NUMA node).  Called by the
    // entries that talk to the device anyway -- never by the host-only ones (afsk_wav_probe, afsk_file_sizes
    // make no HIP call, also not to find out where the device lives).
    // A later call with another mask (the same process ingesting for a GPU on the other socket) moves the
    // threads: every worker re-applies the affinity when the mask's version differs from the one it applied.
    void confine_to(const cpu_set_t& cpus) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!confined_ || !CPU_EQUAL(&cpus_, &cpus)) { cpus_ = cpus; confined_ = true; cpus_ver_++; }
    }
    ~IoPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    template <class F>
    void run(size_t n, unsigned width, F&& body) {
        if (n == 0) return;
        if (width <= 1 || n == 1) {
            for (size_t i = 0; i < n; i++) body(i);
            return;
        }
        std::lock_guard<std::mutex> job_lock(job_mu_);        // one job at a time
        grow(std::min<size_t>(width - 1, n - 1));
        std::function<void(size_t)> fn = std::ref(body);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); busy_ = th_.size(); gen_++;
        }
        cv_.notify_all();
        drain(fn, n);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return busy_ == 0; });
        fn_ = nullptr;
    }
    // body(i) for i in [0, n) on POOL threads only (at least `width` of them exist afterwards, thread creation
    // permitting) while the caller runs main_fn(have_workers) -- a coordinator that consumes what the workers
    // produce; returns when both are done.  When no pool thread could be started main_fn is told so
    // (have_workers == false) and has to run the tasks itself: run_split never does.  One job at a time:
    // job_mu_ is held for the whole call, so a concurrent run() / run_split() of another thread waits.
    template <class F, class M>
    void run_split(size_t n, unsigned width, F&& body, M&& main_fn) {
        std::lock_guard<std::mutex> job_lock(job_mu_);
        grow(std::max<size_t>(1, std::min<size_t>(width, n)));
        std::function<void(size_t)> fn = std::ref(body);
        const bool have_workers = !th_.empty() && n > 0;
        if (have_workers) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                fn_ = &fn; n_ = n; next_.store(0); busy_ = th_.size(); gen_++;
            }
            cv_.notify_all();
        }
        main_fn(have_workers);
        if (have_workers) {
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return busy_ == 0; });
            fn_ = nullptr;
        }
    }
private:
    void drain(const std::function<void(size_t)>& fn, size_t n) {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            fn(i);
        }
    }
    void grow(size_t want) {
        uint64_t gen_now;
        {
            std::lock_guard<std::mutex> lk(mu_);
            gen_now = gen_;                                     // a new worker starts with the NEXT job
        }
        while (th_.size() < want) {
            try {
                th_.emplace_back([this, gen_now] { worker(gen_now); });
            } catch (...) {
                break;                                          // fewer helpers: the caller drains the rest
            }
        }
    }
    void worker(uint64_t seen) {
        uint64_t applied_ver = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
            if (quit_) return;
            seen = gen_;
            if (confined_ && applied_ver != cpus_ver_) {
                (void)pthread_setaffinity_np(pthread_self(), sizeof cpus_, &cpus_);
                applied_ver = cpus_ver_;
            }
            const std::function<void(size_t)>* fn = fn_;
            const size_t n = n_;
            lk.unlock();
            if (fn) drain(*fn, n);
            lk.lock();
            if (--busy_ == 0) done_cv_.notify_all();
        }
    }
    cpu_set_t cpus_;
    uint64_t cpus_ver_ = 0;
    bool confined_ = false;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> th_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, busy_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool quit_ = false;
};
// Never destroyed (no join of parked threads at process exit) and FORK-SAFE: after fork() the child
// has the parent's pool object but none of its threads -- run() would wait for workers that do not
// exist -- so a pthread_atfork child handler abandons the inherited pool (leaked on purpose: its
// std::thread objects are joinable and its mutexes may be held by threads that are gone) and the
// child's first call builds a fresh one.
std::atomic<IoPool*> g_io_pool{nullptr};
std::mutex g_io_pool_mu;
void io_pool_after_fork_in_child() {
    g_io_pool.store(nullptr, std::memory_order_relaxed);
    new (&g_io_pool_mu) std::mutex();            // the parent may have held it at fork time
}
IoPool& io_pool() {
    IoPool* p = g_io_pool.load(std::memory_order_acquire);
    if (p) return *p;
    std::lock_guard<std::mutex> lk(g_io_pool_mu);
    p = g_io_pool.load(std::memory_order_relaxed);
    if (!p) {
        static const int registered = pthread_atfork(nullptr, nullptr, io_pool_after_fork_in_child);
        (void)registered;
        p = new IoPool();
        g_io_pool.store(p, std::memory_order_release);
    }
    return *p;
}
template <class F>
void parallel_for(size_t n, unsigned max_threads, F&& body) {
    io_pool().run(n, max_threads, body);
}
void pool_run(size_t n, unsigned width, const std::function<void(size_t)>& body) { io_pool().run(n, width, body); }

// CPUs this process may really use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us) where one
// is set -- hardware_concurrency() reports every core of the host (256 on the GPU boxes, whose containers get
// the time of 16), and a pool sized for cores the scheduler will not grant is throttled for whole 100 ms
// periods (the 60 - 180 ms outliers of the r3 ingest) -- else the affinity mask / core count.
unsigned usable_cpus() {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 100000;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
        std::fclose(g);
        if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(h, "%lld", &period) != 1) period = 100000;
            std::fclose(h);
        }
    }
    if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    // one process per GPU: the rank processes of a node (LOCAL_WORLD_SIZE, set by torchrun and by bench.py's
    // launcher) share these CPUs -- eight ranks with a quota's worth of threads each are throttled together
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int lw = std::atoi(e);
        if (lw > 1) hw = std::max(1u, hw / (unsigned)lw);
    }
    return hw;
}

unsigned io_threads() {
    static const unsigned v = [] {
        const char* e = std::getenv("AFSK_IO_THREADS");
        return e ? (unsigned)std::max(1, std::atoi(e)) : std::min(32u, usable_cpus());
    }();
    return v;
}

bool pread_all(int fd, void* dst, size_t bytes, int64_t off) {
    char* p = (char*)dst;
    while (bytes > 0) {
        const ssize_t r = pread(fd, p, bytes, (off_t)off);
        if (r <= 0) return false;
        p += r; off += r; bytes -= (size_t)r;
    }
    return true;
}

uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t le16(const unsigned char* p) { return p[0] | (p[1] << 8); }

// The chunk walk of Python's wave.Wave_read.initfp + chunk.Chunk (what ref:214 runs), without
// interpreting the audio format: returns AFSK_WAV_* and the byte range readframes(getnframes()) covers.
int wav_probe_fd(int fd, int64_t* data_off, int64_t* data_bytes);
int wav_probe_one(const char* path, int64_t* data_off, int64_t* data_bytes) {
    *data_off = 0; *data_bytes = 0;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return AFSK_WAV_IO;
    const int rc = wav_probe_fd(fd, data_off, data_bytes);
    close(fd);
    return rc;
}
// the walk itself, on an open file (the caller closes it)
int wav_probe_fd(int fd, int64_t* data_off, int64_t* data_bytes) {
    *data_off = 0; *data_bytes = 0;
    // ONE read covers the RIFF header and the chunk headers of nearly every file (44-byte header,
    // perhaps a LIST chunk); chunk headers beyond it are read one by one.  The file size comes from
    // a short read or, for longer files, from fstat.
    constexpr int64_t kHead = 512;
    unsigned char head[kHead];
    int64_t have = 0;
    for (;;) {
        const ssize_t r = pread(fd, head + have, (size_t)(kHead - have), (off_t)have);
        if (r < 0) return AFSK_WAV_IO;
        if (r == 0) break;
        have += r;
        if (have == kHead) break;
    }
    int64_t fsize = have;
    if (have == kHead) {
        struct stat st;
        if (fstat(fd, &st) != 0) return AFSK_WAV_IO;
        fsize = (int64_t)st.st_size;
    }
    auto fetch = [&](int64_t off, int n, unsigned char* dst) {      // n bytes at off, from the buffer when possible
        if (off + n <= have) { std::memcpy(dst, head + off, (size_t)n); return true; }
        return pread_all(fd, dst, (size_t)n, off);
    };
    unsigned char h[16];
    int rc = AFSK_WAV_NO_DATA;
    if (fsize < 12 || std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0)
        return AFSK_WAV_NOT_RIFF;
    // chunks inside the RIFF form end where the RIFF size says (Chunk.read clips to it), or at EOF
    const int64_t form_end = std::min<int64_t>(fsize, 8 + (int64_t)le32(head + 4));
    int64_t pos = 12;
    int64_t framesize = 0;
    while (pos + 8 <= form_end) {
        if (!fetch(pos, 8, h)) { rc = AFSK_WAV_IO; break; }
        const int64_t csize = (int64_t)le32(h + 4);
        const int64_t body = pos + 8;
        if (std::memcmp(h, "fmt ", 4) == 0) {
            const int64_t avail = std::min<int64_t>(csize, form_end - body);
            if (avail < 16 || !fetch(body, 16, h)) { rc = AFSK_WAV_FORMAT; break; }
            const uint32_t tag = le16(h), channels = le16(h + 2), bits = le16(h + 14);
            const uint32_t width = (bits + 7) / 8;
            if (tag != 1 || channels == 0 || width == 0) { rc = AFSK_WAV_FORMAT; break; }
            framesize = (int64_t)channels * width;
        } else if (std::memcmp(h, "data", 4) == 0) {
            if (framesize == 0) { rc = AFSK_WAV_NO_DATA; break; }       // 'data' before 'fmt '
            const int64_t want = (csize / framesize) * framesize;      // getnframes() whole frames
            const int64_t avail = std::max<int64_t>(0, form_end - body);
            *data_off = body;
            *data_bytes = std::min(want, avail);
            rc = AFSK_WAV_OK;
            break;
        }
        pos = body + csize + (csize & 1);                               // chunks are padded to even sizes
    }
    return rc;
}


// ---- rate-grouped dispatch of a mixed-baud batch (afsk_demod_batch_grouped) -----------------------------
// The host can see bit_frames: the streams are bucketed by value and the batch is decoded by ONE launch of the
// per-stream kernel (every geometry a Receiver can have is compiled into it) that walks the streams BUCKET BY
// BUCKET through an index list, so that the waves resident on a CU at any time run the same geometry's code
// (wave w decodes stream index[w]; every per-stream array stays indexed by the stream number).
// Measured alternatives (profiles/archive/r4_exp2_grouped_modes.txt): one launch of the uniform kernel per bucket --
// on side streams forked / joined with events, or one after the other -- loses 2 ... 70 % to the cross-queue
// hand-over (~27 us per step) and to kernels of different queues hardly overlapping on gfx950
// (hipExtAnyOrderLaunch is not honoured there); the rate-sorted single launch gains 3 ... 12 % over stream
// order when four or more rates are mixed and loses 2 % with three (config #3), hence kSortFromGroups.
struct GroupPlan {
    struct Group { int32_t bf; int32_t first; int32_t count; };
    static constexpr size_t kSortFromGroups = 4;  // fewer distinct rates: stream order (no index list)
    int device = -1;
    int32_t n = 0;
    std::vector<Group> groups;                   // valid rates, largest bucket first; then bf 0: the refused streams
    std::vector<int32_t> h_upload;               // [n] index list (bucket after bucket) + [n] bit_frames by stream number
    int32_t* d_index = nullptr;                  // [n] the permutation; followed, in the same storage, by
    int32_t* d_bf = nullptr;                     // [n] bit_frames by stream number (the kernel's per-stream switch reads it)
    bool own_index = false;
    bool by_length = false;                      // r6: inside every window and bucket the longest streams come first

    ~GroupPlan() {
        if (own_index && d_index) (void)hipFree(d_index);
    }

    static bool valid_bf(int32_t bf) { return bf >= 4 && (bf & 3) == 0 && 2 * bf < AFSK_SYNC_WINDOW; }
    bool sorted() const { return by_length || groups.size() >= sort_from(); }
    static size_t sort_from() {                  // AFSK_GROUP_SORT_FROM overrides (A/B runs)
        const char* e = std::getenv("AFSK_GROUP_SORT_FROM");
        return (e && *e) ? (size_t)std::atol(e) : kSortFromGroups;
    }

    // Ragged batches (r6).  One wavefront decodes one stream whatever its length, and the four wavefronts of a
    // workgroup share its LDS allocation: a workgroup holds its quarter of a CU until its LONGEST stream ends, so four
    // neighbours of 0.25 / 0.25 / 0.25 / 4 s keep three wave slots idle for most of the group's life, and the launch ends
    // when the last long stream does.  With the stream lengths visible to the host (h_len) the walk therefore takes,
    // inside every window and rate bucket, the longest streams first (stable: equal lengths keep stream order) -- the
    // four streams of a workgroup are then neighbours in length, and what a window dispatches last is short.  Only when
    // the lengths differ enough to matter: the shortest stream below 3/4 of the longest (a batch of equal lengths keeps
    // its stream order and, with one rate, its plain uniform launch).
    static bool lengths_ragged(const int32_t* h_len, int32_t n_streams) {
        if (!h_len || n_streams < 8) return false;
        int32_t lo = h_len[0], hi = h_len[0];
        for (int32_t s = 1; s < n_streams; s++) { lo = std::min(lo, h_len[s]); hi = std::max(hi, h_len[s]); }
        return hi > 0 && (int64_t)lo * 4 < (int64_t)hi * 3;
    }

    // host part: buckets and permutation (stable inside a bucket: ascending stream number)
    void bucket(const int32_t* h_bf, int32_t n_streams, const int32_t* h_len = nullptr) {
        n = n_streams;
        by_length = lengths_ragged(h_len, n_streams) && !std::getenv("AFSK_GROUP_NO_LENGTH_SORT");
        std::vector<int32_t> count(AFSK_SYNC_WINDOW / 2 + 1, 0);    // slot 0 = every invalid value
        auto slot = [](int32_t bf) { return valid_bf(bf) ? bf : 0; };
        for (int32_t s = 0; s < n; s++) count[(size_t)slot(h_bf[s])]++;
        std::vector<int32_t> order;
        for (int32_t bf = 4; bf < (int32_t)count.size(); bf += 4)
            if (count[(size_t)bf]) order.push_back(bf);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return count[(size_t)a] > count[(size_t)b]; });
        if (count[0]) order.push_back(0);
        std::vector<int32_t> cursor(count.size(), 0);
        int32_t first = 0;
        for (int32_t bf : order) {
            groups.push_back({bf, first, count[(size_t)bf]});
            cursor[(size_t)bf] = first;
            first += count[(size_t)bf];
        }
        h_upload.resize(2 * (size_t)n);
        // (ragged batches walk windows twice as long: 8192 streams measured best -- time per byte against 1 s streams
        // 1.06 / 1.03 / 1.00 / 1.01 / 1.02 for windows of 2048 / 4096 / 8192 / 16384 / the whole batch,
        // profiles/r6_ragged_windows.txt -- the longest streams of a window live several generations of short ones)
        const int32_t window = by_length && !std::getenv("AFSK_GROUP_WINDOW") ? 2 * sort_window() : sort_window();
        // longest first inside [b, e) of the index list (one bucket of one window)
        auto by_len = [&](size_t b, size_t e) {
            if (by_length && e - b > 1)
                std::stable_sort(h_upload.begin() + (ptrdiff_t)b, h_upload.begin() + (ptrdiff_t)e,
                                 [&](int32_t x, int32_t y) { return h_len[x] > h_len[y]; });
        };
        if (window <= 0 || window >= n) {
            for (int32_t s = 0; s < n; s++) h_upload[(size_t)cursor[(size_t)slot(h_bf[s])]++] = s;
            for (const Group& g : groups) by_len((size_t)g.first, (size_t)g.first + (size_t)g.count);
        } else {
            // bucket order INSIDE windows of `window` consecutive streams (see sort_window)
            std::vector<int32_t> wc(count.size());
            size_t at = 0;
            for (int32_t w0 = 0; w0 < n; w0 += window) {
                const int32_t w1 = std::min(n, w0 + window);
                std::fill(wc.begin(), wc.end(), 0);
                for (int32_t s = w0; s < w1; s++) wc[(size_t)slot(h_bf[s])]++;
                size_t run = at;
                for (int32_t bf : order) { cursor[(size_t)bf] = (int32_t)run; run += (size_t)wc[(size_t)bf]; }
                for (int32_t s = w0; s < w1; s++) h_upload[(size_t)cursor[(size_t)slot(h_bf[s])]++] = s;
                run = at;
                for (int32_t bf : order) { by_len(run, run + (size_t)wc[(size_t)bf]); run += (size_t)wc[(size_t)bf]; }
                at = run;
            }
        }
        if (n > 0) std::memcpy(h_upload.data() + n, h_bf, (size_t)n * 4);
    }

    // The walk visits the buckets one after the other INSIDE windows of this many consecutive streams (r5; 0 = over
    // the whole batch, r4's form).  The 2048 streams in flight of a walk over the whole batch are spread over as many
    // times the address range as there are rates cycling over the streams, and that alone costs 3.5 - 5 % (one rate
    // read in that order: profiles/r5_exp37_address_order.txt).  Windows of 4096 streams -- two generations of
    // resident wavefronts -- keep the streams in flight within 0.4 GB and the neighbouring wavefronts on one rate's
    // code: -2.8 % for four cycling rates, -2.2 % for eighteen, unchanged at 4096 streams (1024: -3.7 / -3.1 % but
    // +2 % at 4096 streams; profiles/r5_exp38_windowed_walk.txt).  AFSK_GROUP_WINDOW overrides (A/B runs).
    static int32_t sort_window() {
        const char* e = std::getenv("AFSK_GROUP_WINDOW");
        if (e && *e) return (int32_t)std::atol(e);
        return 4096;
    }

    // device part: index list + bit_frames either in storage the caller provides (2 n int32, copied
    // asynchronously on `copy_stream`, which every launch of this plan must follow) or in an allocation of its
    // own (copied synchronously)
    hipError_t materialise(int32_t* index_storage, hipStream_t copy_stream) {
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess || n == 0) return e;
        if (index_storage) {
            d_index = index_storage;
            d_bf = d_index + n;
            return hipMemcpyAsync(d_index, h_upload.data(), (size_t)n * 8, hipMemcpyHostToDevice, copy_stream);
        }
        e = hipMalloc((void**)&d_index, (size_t)n * 8);
        if (e != hipSuccess) { d_index = nullptr; return e; }
        d_bf = d_index + n;
        own_index = true;
        return hipMemcpy(d_index, h_upload.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    }

    // ONE launch: the uniform kernel when the whole batch is one valid rate, else the per-stream kernel
    // (in bucket order from kSortFromGroups rates on); a stream with an invalid bit_frames gets status 3 from
    // the kernel itself.  Nothing but a kernel launch: asynchronous, capture-safe, no state shared between calls.
    hipError_t launch(afsk::DemodArgs a, hipStream_t stream) const {
        a.n_streams = n;
        if (groups.size() == 1 && groups[0].bf > 0) {
            a.bit_frames = nullptr;
            a.uniform_bit_frames = groups[0].bf;
            a.stream_index = by_length ? d_index : nullptr;      // one rate, ragged lengths: the uniform kernel walks the list
            return afsk::launch_demod_uniform(a, stream);
        }
        a.bit_frames = d_bf;
        a.stream_index = sorted() ? d_index : nullptr;
        return afsk::launch_demod(a, stream);
    }
};

#define AFSK_HIP(call, what)                             \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) { rc = hip_fail(e_, what); goto done; } \
    } while (0)

// The host-buffer entries allocate (std::vector, std::thread): nothing may be thrown across the
// C boundary, so the bodies live in *_impl and the exported functions catch everything.
template <class F>
int no_throw(F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(AFSK_E_HOST, "out of host memory");
    } catch (const std::exception& e) {
        return fail(AFSK_E_HOST, std::string("host-side failure: ") + e.what());
    } catch (...) {
        return fail(AFSK_E_HOST, "host-side failure");
    }
}

}  // namespace

extern "C" {

int afsk_version(void) { return AFSK_ABI_VERSION; }

int afsk_last_error(char* buf, int cap) {
    if (buf && cap > 0) {
        std::strncpy(buf, g_last_error.c_str(), (size_t)cap - 1);
        buf[cap - 1] = '\0';
    }
    return (int)g_last_error.size();
}

int afsk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int afsk_sync(void* hip_stream) {
    if (int rc = require_device()) return rc;
    hipError_t e = hipStreamSynchronize((hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "hipStreamSynchronize");
}

int afsk_demod_batch(const int16_t* samples, const int64_t* stream_offset,
                     const int32_t* stream_len, const int32_t* bit_frames,
                     int32_t amp_end_threshold, int32_t n_streams, uint8_t* out_bytes,
                     int32_t out_stride, int32_t* out_nbytes, int32_t* out_nbits,
                     int32_t* out_clock_idx, int32_t* out_term_frame, int32_t* out_status,
                     void* hip_stream) {
    return afsk_demod_batch_ex(samples, stream_offset, stream_len, bit_frames, amp_end_threshold,
                               n_streams, out_bytes, out_stride, out_nbytes, out_nbits,
                               out_clock_idx, out_term_frame, out_status, nullptr, nullptr, 0,
                               hip_stream);
}

int afsk_demod_batch_ex(const int16_t* samples, const int64_t* stream_offset,
                        const int32_t* stream_len, const int32_t* bit_frames,
                        int32_t amp_end_threshold, int32_t n_streams, uint8_t* out_bytes,
                        int32_t out_stride, int32_t* out_nbytes, int32_t* out_nbits,
                        int32_t* out_clock_idx, int32_t* out_term_frame, int32_t* out_status,
                        int32_t* out_corrected, int32_t* out_margins, int32_t margin_stride,
                        void* hip_stream) {
    if (n_streams < 0 || out_stride < 0 || margin_stride < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !bit_frames || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::DemodArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.bit_frames = bit_frames; a.amp_end = amp_end_threshold; a.n_streams = n_streams;
    a.out_bytes = out_bytes; a.out_stride = out_stride; a.out_nbytes = out_nbytes;
    a.out_nbits = out_nbits; a.out_clock_idx = out_clock_idx; a.out_term_frame = out_term_frame;
    a.out_status = out_status;
    a.out_corrected = out_corrected;
    a.out_margins = margin_stride > 0 ? out_margins : nullptr;
    a.margin_stride = margin_stride;
    hipError_t e = afsk::launch_demod(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch demod_kernel");
}

int afsk_demod_batch_uniform(const int16_t* samples, const int64_t* stream_offset,
                             const int32_t* stream_len, int32_t bit_frames,
                             int32_t amp_end_threshold, int32_t n_streams, uint8_t* out_bytes,
                             int32_t out_stride, int32_t* out_nbytes, int32_t* out_nbits,
                             int32_t* out_clock_idx, int32_t* out_term_frame, int32_t* out_status,
                             int32_t* out_corrected, int32_t* out_margins, int32_t margin_stride,
                             void* hip_stream) {
    if (n_streams < 0 || out_stride < 0 || margin_stride < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (bit_frames < 4 || (bit_frames & 3) || 2 * bit_frames >= AFSK_SYNC_WINDOW)
        return fail(AFSK_E_INVALID_BAUD, "bit_frames must be a multiple of 4 with 2*bf < 4096");
    if (n_streams == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::DemodArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.bit_frames = nullptr; a.uniform_bit_frames = bit_frames;
    a.amp_end = amp_end_threshold; a.n_streams = n_streams;
    a.out_bytes = out_bytes; a.out_stride = out_stride; a.out_nbytes = out_nbytes;
    a.out_nbits = out_nbits; a.out_clock_idx = out_clock_idx; a.out_term_frame = out_term_frame;
    a.out_status = out_status;
    a.out_corrected = out_corrected;
    a.out_margins = margin_stride > 0 ? out_margins : nullptr;
    a.margin_stride = margin_stride;
    hipError_t e = afsk::launch_demod_uniform(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch demod_uniform_kernel");
}

struct afsk_group_plan { GroupPlan p; };

int afsk_group_plan_create(const int32_t* bit_frames_host, int32_t n_streams, afsk_group_plan** out_plan) {
    return afsk_group_plan_create_ragged(bit_frames_host, nullptr, n_streams, out_plan);
}

int afsk_group_plan_create_ragged(const int32_t* bit_frames_host, const int32_t* stream_len_host, int32_t n_streams,
                                  afsk_group_plan** out_plan) {
    if (!out_plan) return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    *out_plan = nullptr;
    if (n_streams < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams > 0 && !bit_frames_host) return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    return no_throw([&] {
        std::unique_ptr<afsk_group_plan> pl(new afsk_group_plan());
        pl->p.bucket(bit_frames_host, n_streams, stream_len_host);
        hipError_t e = pl->p.materialise(nullptr, nullptr);
        if (e != hipSuccess) return hip_fail(e, "afsk_group_plan_create (index list)");
        *out_plan = pl.release();
        return AFSK_OK;
    });
}

int afsk_group_plan_info(const afsk_group_plan* plan, int32_t* out_n_streams, int32_t* out_n_groups,
                         int32_t* out_group_bit_frames, int32_t* out_group_count, int32_t cap) {
    if (!plan) return fail(AFSK_E_INVALID_ARG, "null plan");
    if (out_n_streams) *out_n_streams = plan->p.n;
    if (out_n_groups) *out_n_groups = (int32_t)plan->p.groups.size();
    for (int32_t k = 0; k < cap && k < (int32_t)plan->p.groups.size(); k++) {
        if (out_group_bit_frames) out_group_bit_frames[k] = plan->p.groups[(size_t)k].bf;
        if (out_group_count) out_group_count[k] = plan->p.groups[(size_t)k].count;
    }
    return AFSK_OK;
}

int afsk_group_plan_destroy(afsk_group_plan* plan) {
    delete plan;      // the index list; the caller has synchronised its launches
    return AFSK_OK;
}

int afsk_demod_batch_grouped(const afsk_group_plan* plan, const int16_t* samples,
                             const int64_t* stream_offset, const int32_t* stream_len,
                             int32_t amp_end_threshold, uint8_t* out_bytes, int32_t out_stride,
                             int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                             int32_t* out_term_frame, int32_t* out_status, int32_t* out_corrected,
                             int32_t* out_margins, int32_t margin_stride, void* hip_stream) {
    if (!plan) return fail(AFSK_E_INVALID_ARG, "null plan");
    if (out_stride < 0 || margin_stride < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (plan->p.n == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != plan->p.device)
        return fail(AFSK_E_INVALID_ARG, "the plan was created on another device than the current one");
    afsk::DemodArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.amp_end = amp_end_threshold;
    a.out_bytes = out_bytes; a.out_stride = out_stride; a.out_nbytes = out_nbytes;
    a.out_nbits = out_nbits; a.out_clock_idx = out_clock_idx; a.out_term_frame = out_term_frame;
    a.out_status = out_status;
    a.out_corrected = out_corrected;
    a.out_margins = margin_stride > 0 ? out_margins : nullptr;
    a.margin_stride = margin_stride;
    hipError_t e = plan->p.launch(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch demod_kernel (grouped)");
}

// The host entries see the bit_frames array: one value for all streams -> the uniform kernel;
// several -> the grouped dispatch (index list + bit_frames live in `index_storage`, 2 n int32 of the
// caller's device scratch; `keep` owns the side streams until the caller has synchronised).
static int demod_device_auto(const int32_t* h_bit_frames, const int32_t* h_stream_len, const int16_t* samples, const int64_t* stream_offset,
                             const int32_t* stream_len, const int32_t* d_bit_frames, int32_t amp_end_threshold,
                             int32_t n_streams, uint8_t* out_bytes, int32_t out_stride, int32_t* out_nbytes,
                             int32_t* out_nbits, int32_t* out_clock_idx, int32_t* out_term_frame,
                             int32_t* out_status, hipStream_t stream, int32_t* index_storage,
                             std::unique_ptr<GroupPlan>& keep) {
    (void)d_bit_frames;
    bool same = true;
    for (int32_t s = 1; s < n_streams && same; s++) same = h_bit_frames[s] == h_bit_frames[0];
    // (one rate but ragged lengths: the plan, whose walk takes the longest streams first -- GroupPlan::bucket)
    if (same && !GroupPlan::lengths_ragged(h_stream_len, n_streams))
        return afsk_demod_batch_uniform(samples, stream_offset, stream_len, h_bit_frames[0], amp_end_threshold,
                                        n_streams, out_bytes, out_stride, out_nbytes, out_nbits, out_clock_idx,
                                        out_term_frame, out_status, nullptr, nullptr, 0, stream);
    if (int rc = require_device()) return rc;
    keep.reset(new GroupPlan());
    keep->bucket(h_bit_frames, n_streams, h_stream_len);
    hipError_t e = keep->materialise(index_storage, stream);
    if (e != hipSuccess) return hip_fail(e, "grouped dispatch (index list)");
    afsk::DemodArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.amp_end = amp_end_threshold;
    a.out_bytes = out_bytes; a.out_stride = out_stride; a.out_nbytes = out_nbytes;
    a.out_nbits = out_nbits; a.out_clock_idx = out_clock_idx; a.out_term_frame = out_term_frame;
    a.out_status = out_status;
    e = keep->launch(a, stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch demod_kernel (grouped)");
}

static int demod_batch_host_impl(const int16_t* samples, int64_t total_samples,
                                 const int64_t* stream_offset, const int32_t* stream_len,
                                 const int32_t* bit_frames, int32_t amp_end_threshold,
                                 int32_t n_streams, uint8_t* out_bytes, int32_t out_stride,
                                 int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                                 int32_t* out_term_frame, int32_t* out_status) {
    if (n_streams < 0 || out_stride < 0 || total_samples < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    if (!stream_offset || !stream_len || !bit_frames || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0) ||
        (!samples && total_samples > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    for (int32_t s = 0; s < n_streams; s++) {
        const int bf = bit_frames[s];
        if (bf < 4 || (bf & 3) || 2 * bf >= AFSK_SYNC_WINDOW)
            return fail(AFSK_E_INVALID_BAUD, "bit_frames must be a multiple of 4 with 2*bf < 4096");
        if (stream_len[s] < 0 || stream_len[s] > AFSK_MAX_STREAM_LEN || stream_offset[s] < 0 ||
            stream_offset[s] + stream_len[s] > total_samples)
            return fail(AFSK_E_INVALID_ARG, "stream outside the sample buffer (or longer than AFSK_MAX_STREAM_LEN)");
    }
    if (int rc0 = require_device()) return rc0;

    int rc = AFSK_OK;
    const size_t n = (size_t)n_streams;
    const size_t sample_bytes = (size_t)(total_samples > 0 ? total_samples : 1) * 2;
    const size_t bytes_out = n * (size_t)out_stride;
    // device layout: samples | meta = offsets, len, bf | out = 5 x int32 [n], bytes [n, stride]
    const size_t o_meta = (sample_bytes + 255) & ~(size_t)255;
    const size_t meta_bytes = n * 16;
    const size_t o_out = o_meta + meta_bytes + n * 8;      // + the index list and bit_frames of a grouped dispatch
    const size_t out_bytes_total = n * 20 + bytes_out;
    const size_t total = o_out + out_bytes_total;
    hipStream_t stream = nullptr;
    {
        hipError_t e = g_thread_stream.get(&stream);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreateWithFlags (host-entry stream)");
    }
    ScratchLease lease;
    std::unique_ptr<GroupPlan> plan;      // mixed rates: the host copy of the index list, alive until the final synchronise
    char* d_all = nullptr;
    // host staging: one H2D for the three index arrays, one D2H for all six outputs
    std::vector<char> h_meta(meta_bytes), h_out(out_bytes_total);
    std::memcpy(h_meta.data(), stream_offset, n * 8);
    std::memcpy(h_meta.data() + n * 8, stream_len, n * 4);
    std::memcpy(h_meta.data() + n * 12, bit_frames, n * 4);
    {
        hipError_t e = lease.acquire(total, &d_all);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-entry scratch)");
    }
    if (total_samples > 0)
        AFSK_HIP(hipMemcpyAsync(d_all, samples, (size_t)total_samples * 2, hipMemcpyHostToDevice, stream),
                 "H2D samples");
    AFSK_HIP(hipMemcpyAsync(d_all + o_meta, h_meta.data(), meta_bytes, hipMemcpyHostToDevice, stream),
             "H2D stream index");
    {
        int32_t* i32 = (int32_t*)(d_all + o_out);
        rc = demod_device_auto(bit_frames, stream_len, (const int16_t*)d_all, (const int64_t*)(d_all + o_meta),
                               (const int32_t*)(d_all + o_meta + n * 8),
                               (const int32_t*)(d_all + o_meta + n * 12), amp_end_threshold, n_streams,
                               (uint8_t*)(d_all + o_out + n * 20), out_stride, i32, i32 + n, i32 + 2 * n,
                               i32 + 3 * n, i32 + 4 * n, stream,
                               (int32_t*)(d_all + o_meta + n * 16), plan);
        if (rc != AFSK_OK) goto done;
    }
    AFSK_HIP(hipMemcpyAsync(h_out.data(), d_all + o_out, out_bytes_total, hipMemcpyDeviceToHost, stream),
             "D2H results");
    AFSK_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize");
    std::memcpy(out_nbytes, h_out.data(), n * 4);
    std::memcpy(out_nbits, h_out.data() + n * 4, n * 4);
    std::memcpy(out_clock_idx, h_out.data() + n * 8, n * 4);
    std::memcpy(out_term_frame, h_out.data() + n * 12, n * 4);
    std::memcpy(out_status, h_out.data() + n * 16, n * 4);
    if (bytes_out > 0) std::memcpy(out_bytes, h_out.data() + n * 20, bytes_out);
done:
    // nothing may still use the scratch or the staging vectors when they are released
    if (rc != AFSK_OK) (void)hipStreamSynchronize(stream);
    return rc;   // the lease returns (or frees) the device scratch
}

int afsk_demod_batch_host(const int16_t* samples, int64_t total_samples,
                          const int64_t* stream_offset, const int32_t* stream_len,
                          const int32_t* bit_frames, int32_t amp_end_threshold,
                          int32_t n_streams, uint8_t* out_bytes, int32_t out_stride,
                          int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                          int32_t* out_term_frame, int32_t* out_status) {
    return no_throw([&] {
        return demod_batch_host_impl(samples, total_samples, stream_offset, stream_len, bit_frames,
                                     amp_end_threshold, n_streams, out_bytes, out_stride, out_nbytes,
                                     out_nbits, out_clock_idx, out_term_frame, out_status);
    });
}

int afsk_host_scratch_release(void) {
    std::lock_guard<std::mutex> lk(g_scratch.mu);
    if (g_scratch.ptr) {
        hipError_t e = hipFree(g_scratch.ptr);
        g_scratch.ptr = nullptr;
        g_scratch.cap = 0;
        if (e != hipSuccess) return hip_fail(e, "hipFree (host-entry scratch)");
    }
    for (int k = 0; k < 2; k++) {
        if (g_scratch.stage[k]) { (void)hipHostFree(g_scratch.stage[k]); g_scratch.stage[k] = nullptr; }
        if (g_scratch.stage_free[k]) { (void)hipEventDestroy(g_scratch.stage_free[k]); g_scratch.stage_free[k] = nullptr; }
        if (g_scratch.ring_extra[k]) { (void)hipHostFree(g_scratch.ring_extra[k]); g_scratch.ring_extra[k] = nullptr; }
    }
    for (hipEvent_t& ev : g_scratch.ring_sent)
        if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    return AFSK_OK;
}

static int demod_streams_host_impl(const int16_t* const* streams, const int32_t* stream_len,
                                   const int32_t* bit_frames, int32_t amp_end_threshold,
                                   int32_t n_streams, uint8_t* out_bytes, int32_t out_stride,
                                   int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                                   int32_t* out_term_frame, int32_t* out_status) {
    if (n_streams < 0 || out_stride < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    if (!streams || !stream_len || !bit_frames || !out_nbytes || !out_nbits || !out_clock_idx ||
        !out_term_frame || !out_status || (!out_bytes && out_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    const size_t n = (size_t)n_streams;
    // device layout: every stream starts on a 16-byte boundary
    std::vector<char> h_meta(n * 16);
    int64_t* h_off = (int64_t*)h_meta.data();
    int64_t total_samples = 0;
    for (size_t s = 0; s < n; s++) {
        const int bf = bit_frames[s];
        if (bf < 4 || (bf & 3) || 2 * bf >= AFSK_SYNC_WINDOW)
            return fail(AFSK_E_INVALID_BAUD, "bit_frames must be a multiple of 4 with 2*bf < 4096");
        if (stream_len[s] < 0 || stream_len[s] > AFSK_MAX_STREAM_LEN || (stream_len[s] > 0 && !streams[s]))
            return fail(AFSK_E_INVALID_ARG, "bad stream length or null stream pointer");
        h_off[s] = total_samples;
        total_samples += ((int64_t)stream_len[s] + 7) & ~(int64_t)7;
    }
    std::memcpy(h_meta.data() + n * 8, stream_len, n * 4);
    std::memcpy(h_meta.data() + n * 12, bit_frames, n * 4);
    if (int rc0 = require_device()) return rc0;

    int rc = AFSK_OK;
    const size_t sample_bytes = (size_t)(total_samples > 0 ? total_samples : 1) * 2;
    const size_t bytes_out = n * (size_t)out_stride;
    const size_t o_meta = (sample_bytes + 255) & ~(size_t)255;
    const size_t o_out = o_meta + n * 16 + n * 8;          // + the index list and bit_frames of a grouped dispatch
    const size_t out_bytes_total = n * 20 + bytes_out;
    hipStream_t stream = nullptr;
    {
        hipError_t e = g_thread_stream.get(&stream);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreateWithFlags (host-entry stream)");
    }
    {
        cpu_set_t cpus;                                   // the packing threads belong on the device's socket
        if (device_node_cpus(&cpus)) io_pool().confine_to(cpus);
    }
    ScratchLease lease;
    std::unique_ptr<GroupPlan> plan;      // mixed rates: the host copy of the index list, alive until the final synchronise
    char* d_all = nullptr;
    char* stage[2];
    hipEvent_t stage_free[2];
    std::vector<char> h_out(out_bytes_total);
    {
        hipError_t e = lease.acquire(o_out + out_bytes_total, &d_all, /*block=*/true);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-entry scratch)");
        e = lease.staging(&stage[0], &stage[1], &stage_free[0], &stage_free[1]);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (staging)");
    }
    {
        // windows of the device sample range; pieces of streams are packed into a pinned window
        // by the copy threads while the previous window is on the wire
        const size_t all = (size_t)total_samples * 2;
        // small batches use smaller windows so that packing and DMA still overlap
        const size_t win = std::min(kStageBytes, std::max((size_t)1 << 20, ((all / 4) + 65535) & ~(size_t)65535));
        size_t s_cur = 0;                                  // first stream that may reach into the window
        std::vector<CopyJob> jobs;
        for (size_t w0 = 0, k = 0; w0 < all; w0 += win, k++) {
            const size_t w1 = std::min(all, w0 + win);
            char* st = stage[k & 1];
            if (k >= 2) AFSK_HIP(hipEventSynchronize(stage_free[k & 1]), "hipEventSynchronize");
            jobs.clear();
            size_t moved = 0;
            for (size_t s = s_cur; s < n; s++) {
                const size_t b0 = (size_t)h_off[s] * 2, b1 = b0 + (size_t)stream_len[s] * 2;
                if (b0 >= w1) break;
                if (b1 <= w0) { s_cur = s + 1; continue; }
                const size_t lo = std::max(b0, w0), hi = std::min(b1, w1);
                jobs.push_back({st + (lo - w0), (const char*)streams[s] + (lo - b0), hi - lo});
                moved += hi - lo;
            }
            parallel_copy(jobs, moved);
            AFSK_HIP(hipMemcpyAsync(d_all + w0, st, w1 - w0, hipMemcpyHostToDevice, stream), "H2D samples");
            AFSK_HIP(hipEventRecord(stage_free[k & 1], stream), "hipEventRecord");
        }
    }
    AFSK_HIP(hipMemcpyAsync(d_all + o_meta, h_meta.data(), n * 16, hipMemcpyHostToDevice, stream),
             "H2D stream index");
    {
        int32_t* i32 = (int32_t*)(d_all + o_out);
        rc = demod_device_auto(bit_frames, stream_len, (const int16_t*)d_all, (const int64_t*)(d_all + o_meta),
                               (const int32_t*)(d_all + o_meta + n * 8),
                               (const int32_t*)(d_all + o_meta + n * 12), amp_end_threshold, n_streams,
                               (uint8_t*)(d_all + o_out + n * 20), out_stride, i32, i32 + n, i32 + 2 * n,
                               i32 + 3 * n, i32 + 4 * n, stream,
                               (int32_t*)(d_all + o_meta + n * 16), plan);
        if (rc != AFSK_OK) goto done;
    }
    AFSK_HIP(hipMemcpyAsync(h_out.data(), d_all + o_out, out_bytes_total, hipMemcpyDeviceToHost, stream),
             "D2H results");
    AFSK_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize");
    std::memcpy(out_nbytes, h_out.data(), n * 4);
    std::memcpy(out_nbits, h_out.data() + n * 4, n * 4);
    std::memcpy(out_clock_idx, h_out.data() + n * 8, n * 4);
    std::memcpy(out_term_frame, h_out.data() + n * 12, n * 4);
    std::memcpy(out_status, h_out.data() + n * 16, n * 4);
    if (bytes_out > 0) std::memcpy(out_bytes, h_out.data() + n * 20, bytes_out);
done:
    if (rc != AFSK_OK) (void)hipStreamSynchronize(stream);   // nothing may still read the staging windows
    return rc;
}

int afsk_demod_streams_host(const int16_t* const* streams, const int32_t* stream_len,
                            const int32_t* bit_frames, int32_t amp_end_threshold,
                            int32_t n_streams, uint8_t* out_bytes, int32_t out_stride,
                            int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                            int32_t* out_term_frame, int32_t* out_status) {
    return no_throw([&] {
        return demod_streams_host_impl(streams, stream_len, bit_frames, amp_end_threshold, n_streams,
                                       out_bytes, out_stride, out_nbytes, out_nbits, out_clock_idx,
                                       out_term_frame, out_status);
    });
}

int afsk_wav_probe(const char* const* paths, int32_t n_files, int64_t* out_data_offset,
                   int64_t* out_data_bytes, int32_t* out_status) {
    if (n_files < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_files == 0) return AFSK_OK;
    if (!paths || !out_data_offset || !out_data_bytes || !out_status)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    for (int32_t i = 0; i < n_files; i++)
        if (!paths[i]) return fail(AFSK_E_INVALID_ARG, "null path");
    return no_throw([&] {
        parallel_for((size_t)n_files, io_threads(), [&](size_t i) {
            out_status[i] = wav_probe_one(paths[i], &out_data_offset[i], &out_data_bytes[i]);
        });
        return AFSK_OK;
    });
}

static int wav_upload_impl(const char* const* paths, const int64_t* data_offset, const int64_t* data_bytes,
                           const int64_t* stream_offset, int32_t n_files, int16_t* d_samples,
                           int64_t capacity_samples) {
    if (n_files < 0 || capacity_samples < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_files == 0) return AFSK_OK;
    if (!paths || !data_offset || !data_bytes || !stream_offset || !d_samples)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    const size_t n = (size_t)n_files;
    int64_t prev_end = 0;
    for (size_t s = 0; s < n; s++) {
        if (!paths[s] || data_offset[s] < 0 || data_bytes[s] < 0 || stream_offset[s] < prev_end)
            return fail(AFSK_E_INVALID_ARG, "streams must be ascending and must not overlap");
        prev_end = stream_offset[s] + data_bytes[s] / 2;
        if (prev_end > capacity_samples) return fail(AFSK_E_INVALID_ARG, "stream outside the device buffer");
    }
    if (int rc0 = require_device()) return rc0;
    int rc = AFSK_OK;
    hipStream_t stream = nullptr;
    {
        hipError_t e = g_thread_stream.get(&stream);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreateWithFlags (host-entry stream)");
    }
    ScratchLease lease;
    char* d_unused = nullptr;
    char* stage[2];
    hipEvent_t stage_free[2];
    {
        hipError_t e = lease.acquire(1, &d_unused, /*block=*/true);    // owns the staging windows
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-entry scratch)");
        e = lease.staging(&stage[0], &stage[1], &stage_free[0], &stage_free[1]);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (staging)");
    }
    std::vector<int> fds(n, -1);
    std::atomic<int> io_failed{-1};
    struct Piece { size_t file; int64_t file_off; char* dst; size_t bytes; bool last; };
    constexpr size_t kZeroPiece = ~(size_t)0;             // Piece::file of a zero-filled alignment gap
    constexpr size_t kUploadGapFill = 256;
    {
        // windows over the device byte range the streams cover; a window holds whole or partial
        // streams, each read by one pread straight into pinned memory
        const size_t first_b = (size_t)stream_offset[0] * 2;
        const size_t all_end = (size_t)prev_end * 2;
        const size_t span = all_end - first_b;
        const size_t win = std::min(kStageBytes, std::max((size_t)1 << 20, ((span / 4) + 65535) & ~(size_t)65535));
        size_t s_cur = 0;
        std::vector<Piece> pieces;
        for (size_t w0 = first_b, k = 0; w0 < all_end; w0 += win, k++) {
            const size_t w1 = std::min(all_end, w0 + win);
            char* st = stage[k & 1];
            if (k >= 2) AFSK_HIP(hipEventSynchronize(stage_free[k & 1]), "hipEventSynchronize");
            pieces.clear();
            for (size_t s = s_cur; s < n; s++) {
                const size_t b0 = (size_t)stream_offset[s] * 2, b1 = b0 + ((size_t)data_bytes[s] & ~(size_t)1);
                if (b0 >= w1) break;
                // the alignment gap behind the stream (up to kGapFill bytes to the next stream's start) is a
                // piece of its own, filled with zeros -- in whichever window(s) it falls
                size_t g1 = b1;
                if (s + 1 < n) {
                    const size_t nb0 = (size_t)stream_offset[s + 1] * 2;
                    if (nb0 > b1 && nb0 - b1 <= kUploadGapFill) g1 = nb0;
                }
                if (g1 <= w0) { s_cur = s + 1; continue; }
                if (b1 > w0) {
                    const size_t lo = std::max(b0, w0), hi = std::min(b1, w1);
                    if (hi > lo) pieces.push_back({s, data_offset[s] + (int64_t)(lo - b0), st + (lo - w0), hi - lo, hi == b1});
                }
                const size_t zlo = std::max(b1, w0), zhi = std::min(g1, w1);
                if (zhi > zlo) pieces.push_back({kZeroPiece, 0, st + (zlo - w0), zhi - zlo, false});
            }
            parallel_for(pieces.size(), io_threads(), [&](size_t i) {
                const Piece& p = pieces[i];
                if (p.file == kZeroPiece) { std::memset(p.dst, 0, p.bytes); return; }
                if (fds[p.file] < 0) fds[p.file] = open(paths[p.file], O_RDONLY | O_CLOEXEC);
                if (fds[p.file] < 0 || !pread_all(fds[p.file], p.dst, p.bytes, p.file_off))
                    io_failed.store((int)p.file);
                if (p.last && fds[p.file] >= 0) {           // closed here, in parallel, not in a serial loop at the end
                    close(fds[p.file]);
                    fds[p.file] = -1;
                }
            });
            if (io_failed.load() >= 0) {
                rc = fail(AFSK_E_HOST, std::string("cannot read ") + paths[io_failed.load()]);
                goto done;
            }
            // Only the stream pieces of the window were filled.  Send it as contiguous RUNS of pieces:
            // an alignment gap of up to kGapFill bytes between two streams is zero-filled and travels with
            // the run (one copy per stream would cost more than the transfer); a larger gap -- room the
            // caller keeps for streams it fills some other way -- ends the run and is never written.
            {
                size_t run0 = 0, run1 = 0;                              // window-relative byte range of the open run
                bool open = false;
                for (const Piece& p : pieces) {
                    const size_t a0 = (size_t)(p.dst - st), a1 = a0 + p.bytes;
                    if (open && a0 == run1) {                           // (gaps to be zeroed are pieces themselves)
                        run1 = a1;
                        continue;
                    }
                    if (open)
                        AFSK_HIP(hipMemcpyAsync((char*)d_samples + w0 + run0, st + run0, run1 - run0, hipMemcpyHostToDevice, stream), "H2D samples");
                    run0 = a0; run1 = a1; open = true;
                }
                if (open)
                    AFSK_HIP(hipMemcpyAsync((char*)d_samples + w0 + run0, st + run0, run1 - run0, hipMemcpyHostToDevice, stream), "H2D samples");
            }
            AFSK_HIP(hipEventRecord(stage_free[k & 1], stream), "hipEventRecord");
        }
    }
    AFSK_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize");
done:
    if (rc != AFSK_OK) (void)hipStreamSynchronize(stream);   // nothing may still read the staging windows
    for (int fd : fds)
        if (fd >= 0) close(fd);
    return rc;
}

int afsk_wav_upload(const char* const* paths, const int64_t* data_offset, const int64_t* data_bytes,
                    const int64_t* stream_offset, int32_t n_files, int16_t* d_samples,
                    int64_t capacity_samples) {
    return no_throw([&] {
        return wav_upload_impl(paths, data_offset, data_bytes, stream_offset, n_files, d_samples, capacity_samples);
    });
}

int afsk_file_sizes(const char* const* paths, int32_t n_files, int64_t* out_size_bytes) {
    if (n_files < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_files == 0) return AFSK_OK;
    if (!paths || !out_size_bytes) return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    for (int32_t i = 0; i < n_files; i++)
        if (!paths[i]) return fail(AFSK_E_INVALID_ARG, "null path");
    return no_throw([&] {
        parallel_for((size_t)n_files, io_threads(), [&](size_t i) {
            struct stat st;
            out_size_bytes[i] = stat(paths[i], &st) == 0 ? (int64_t)st.st_size : -1;
        });
        return AFSK_OK;
    });
}

// afsk_wav_ingest: ONE pass per file -- open, header walk, pread of the data chunk straight into pinned
// memory, close -- pipelined against the H2D copies.  The device range is cut into windows of whole
// slots (at most kIngestWindow bytes; a gap of more than kGapFill bytes between two slots ends a
// window); a window lives in one of kIngestSlots staging buffers.  Pool threads take the pieces in
// order and fill them; the calling thread is the coordinator: when the last piece of window w is in, it
// sends the window, and window w + kIngestSlots may be filled once the copy of window w has completed.
static int wav_ingest_impl(const char* const* paths, int32_t n_files, const int64_t* slot_offset,
                           const int64_t* slot_samples, int16_t* d_samples, int64_t capacity_samples,
                           int64_t* out_data_offset, int64_t* out_data_bytes, int32_t* out_status) {
    if (n_files < 0 || capacity_samples < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_files == 0) return AFSK_OK;
    if (!paths || !slot_offset || !slot_samples || !d_samples || !out_data_offset || !out_data_bytes || !out_status)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    const size_t n = (size_t)n_files;
    int64_t prev_end = 0;
    for (size_t s = 0; s < n; s++) {
        if (!paths[s] || slot_samples[s] < 0 || slot_offset[s] < prev_end)
            return fail(AFSK_E_INVALID_ARG, "slots must be ascending and must not overlap");
        prev_end = slot_offset[s] + slot_samples[s];
        if (prev_end > capacity_samples) return fail(AFSK_E_INVALID_ARG, "slot outside the device buffer");
    }
    if (int rc0 = require_device()) return rc0;
    // Staging ring: kSlots pinned buffers of kWindow bytes (default 8 x 16 MiB = the two 32 MiB windows of the
    // gather entry plus two more chunks allocated on first use; AFSK_INGEST_WINDOW_MB = 4 / 8 / 16 / 32 and
    // AFSK_INGEST_SLOTS = 2 ... 16 for experiments, at most 128 MiB in all).
    static const size_t kWindow = [] {
        const char* e = std::getenv("AFSK_INGEST_WINDOW_MB");
        const int mb = e ? std::atoi(e) : 16;
        return (size_t)((mb == 4 || mb == 8 || mb == 16 || mb == 32) ? mb : 16) << 20;
    }();
    static const int kSlots = [] {
        const char* e = std::getenv("AFSK_INGEST_SLOTS");
        const int want = e ? std::atoi(e) : 8;
        const int most = (int)std::min<size_t>((size_t)kMaxRingSlots, kRingChunks * kStageBytes / kWindow);
        return std::max(2, std::min(want, most));
    }();
    static const bool kStats = std::getenv("AFSK_INGEST_STATS") != nullptr;
    constexpr size_t kGapFill = 256;
    int rc = AFSK_OK;
    {
        cpu_set_t cpus;                                   // (one process per GPU: the current device's socket)
        if (device_node_cpus(&cpus)) io_pool().confine_to(cpus);
    }
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    {
        hipError_t e = g_thread_stream.get(&copy_stream[0]);
        if (e == hipSuccess) e = g_thread_stream2.get(&copy_stream[1]);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreateWithFlags (host-entry streams)");
    }
    ScratchLease lease;
    char* d_unused = nullptr;
    char* stage[kMaxRingSlots];
    hipEvent_t sent[kMaxRingSlots];
    {
        hipError_t e = lease.acquire(1, &d_unused, /*block=*/true);    // owns the staging ring
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-entry scratch)");
        e = lease.ring(kWindow, kSlots, stage, sent);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc / hipEventCreate (staging ring)");
    }

    struct Piece { size_t file; size_t win; size_t slot_lo; size_t bytes; size_t stage_off; };
    struct Window { size_t dev_b0; size_t bytes; };
    std::vector<Piece> pieces;
    std::vector<Window> wins;
    pieces.reserve(n + 8);
    {
        size_t w_b0 = 0, w_b1 = 0;                    // open window's device byte range
        bool open_w = false;
        size_t last_end = 0;                          // device byte where the last closed window ended
        bool have_end = false;
        auto close_window = [&] {
            if (open_w) { wins.push_back({w_b0, w_b1 - w_b0}); open_w = false; last_end = w_b1; have_end = true; }
        };
        // the first windows are SMALL (1, 2, 4 ... MiB up to the full window): the first transfer starts after a
        // fraction of a millisecond of file reading instead of after a whole 16 MiB of it
        auto win_cap = [&] { return std::min(kWindow, ((size_t)1 << 20) << std::min<size_t>(wins.size(), 6)); };
        for (size_t s = 0; s < n; s++) {
            const size_t b0 = (size_t)slot_offset[s] * 2, cap = (size_t)slot_samples[s] * 2;
            if (cap == 0) {                           // nothing to copy, but the file is still probed
                // (an empty slot is a slot: a small gap in front of it travels as zeros like any other)
                if (open_w && b0 > w_b1 && b0 - w_b1 <= kGapFill && b0 - w_b0 <= win_cap()) w_b1 = b0;
                if (!open_w) {
                    w_b0 = (have_end && b0 > last_end && b0 - last_end <= kGapFill) ? last_end : b0;
                    w_b1 = b0;
                    open_w = true;
                }
                pieces.push_back({s, wins.size(), 0, 0, 0});
                continue;
            }
            size_t done = 0;
            while (done < cap) {
                const size_t p0 = b0 + done;
                const size_t wcap = win_cap();
                if (open_w && (p0 > w_b1 + kGapFill || p0 - w_b0 >= wcap)) close_window();
                if (!open_w) {
                    // a small alignment gap behind the previous window's last slot becomes the head of this one
                    // (the sender zero-fills everything of a window that no piece covers)
                    w_b0 = (have_end && p0 > last_end && p0 - last_end <= kGapFill) ? last_end : p0;
                    w_b1 = p0;
                    open_w = true;
                }
                const size_t wcap2 = win_cap();       // (closing a window may have moved on to a larger one)
                const size_t room = wcap2 - (p0 - w_b0);
                const size_t take = std::min(cap - done, room);
                // a whole slot that does not fit the rest of this window starts the next one
                if (take < cap - done && done == 0 && cap <= wcap2 && p0 != w_b0) { close_window(); continue; }
                pieces.push_back({s, wins.size(), done, take, p0 - w_b0});
                done += take;
                w_b1 = p0 + take;
                if (w_b1 - w_b0 >= wcap2) close_window();
            }
        }
        close_window();
    }
    const size_t nwin = wins.size(), npieces = pieces.size();
    std::vector<std::atomic<int>> remaining(std::max<size_t>(nwin, 1));
    for (auto& r : remaining) r.store(0);
    for (const Piece& pc : pieces) if (pc.win < nwin) remaining[pc.win].fetch_add(1);
    // Hand-over between the pool threads (fill) and the calling thread (send), under ONE mutex and two
    // condition variables -- nobody spins: `released` = windows whose staging buffer may be filled (a filler
    // whose piece lies beyond it sleeps on cv_released); a filler that completes a window wakes the sender
    // (cv_filled).
    std::mutex hand_mu;
    std::condition_variable cv_released, cv_filled;
    long released = (long)std::min<size_t>((size_t)kSlots, nwin);
    std::atomic<int> failed{0};
    std::atomic<long> io_failed_file{-1};
    // stats (AFSK_INGEST_STATS): nanoseconds summed over the fillers / spent by the sender
    std::atomic<long long> ns_wait_release{0}, ns_open{0}, ns_read{0}, ns_close{0}, n_fast{0};
    long long ns_wait_fill = 0, ns_wait_sent = 0, ns_issue = 0;
    auto now_ns = [] { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(
                           std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const long long t_begin = now_ns();

    auto fill_piece = [&](size_t i) {
        const Piece& pc = pieces[i];
        {
            const long long t0 = kStats ? now_ns() : 0;
            std::unique_lock<std::mutex> lk(hand_mu);
            cv_released.wait(lk, [&] { return (long)pc.win < released || failed.load(std::memory_order_relaxed); });
            if (kStats) ns_wait_release.fetch_add(now_ns() - t0, std::memory_order_relaxed);
        }
        char* dst = pc.win < nwin ? stage[pc.win % (size_t)kSlots] + pc.stage_off : nullptr;
        if (!failed.load(std::memory_order_relaxed)) {
            int64_t doff = 0, dbytes = 0;
            int st = AFSK_WAV_IO;
            long long t0 = kStats ? now_ns() : 0;
            const int fd = open(paths[pc.file], O_RDONLY | O_CLOEXEC);
            if (kStats) { const long long t1 = now_ns(); ns_open.fetch_add(t1 - t0, std::memory_order_relaxed); t0 = t1; }
            const size_t cap = (size_t)slot_samples[pc.file] * 2;
            bool have_data = false;
            // The common file -- a canonical 44-byte header ('RIFF' size 'WAVE' 'fmt ' 16 ... 'data' size, what the
            // stdlib writer behind SoundOutput.writeToFile produces, ref:256-263) whose whole data chunk is this
            // one piece -- is read with ONE preadv: header into a local buffer, data straight into the staging
            // slot.  Anything else (more chunks, odd sizes, a piece of a split slot) takes the general walk below.
            if (fd >= 0 && pc.slot_lo == 0 && pc.bytes == cap && cap > 0) {
                unsigned char hdr[44];
                struct iovec iov[2] = {{hdr, sizeof hdr}, {dst, cap}};
                const ssize_t got = preadv(fd, iov, 2, 0);
                struct stat sb;
                // the read is trusted only if it returned everything the file holds up to the end of the slot
                // (a short read that is not the end of the file -- POSIX allows it -- takes the general walk)
                if (got >= 44 && std::memcmp(hdr, "RIFF", 4) == 0 && std::memcmp(hdr + 8, "WAVEfmt ", 8) == 0 &&
                    le32(hdr + 16) == 16 && std::memcmp(hdr + 36, "data", 4) == 0 && fstat(fd, &sb) == 0 &&
                    (int64_t)got == std::min<int64_t>((int64_t)sb.st_size, 44 + (int64_t)cap)) {
                    const uint32_t tag = le16(hdr + 20), channels = le16(hdr + 22), width = (le16(hdr + 34) + 7) / 8;
                    const int64_t framesize = (int64_t)channels * width;
                    const int64_t csize = (int64_t)le32(hdr + 40);
                    // (a RIFF size field below 36 puts a chunk header outside the form: the stdlib reader -- and
                    // wav_probe_fd, which demands pos + 8 <= form_end for both chunk headers -- rejects such a file,
                    // so it must not be reported OK with 0 data bytes here: the general walk below gives its status)
                    const int64_t form_end = std::min<int64_t>((int64_t)sb.st_size, 8 + (int64_t)le32(hdr + 4));
                    if (tag == 1 && framesize > 0 && form_end >= 44) {
                        const int64_t want = (csize / framesize) * framesize;          // getnframes() whole frames
                        const int64_t avail = std::max<int64_t>(0, form_end - 44);
                        doff = 44; dbytes = std::min(want, avail);
                        const size_t usable = (size_t)dbytes & ~(size_t)1;
                        if (usable <= cap && 44 + (int64_t)usable <= (int64_t)got) {
                            st = AFSK_WAV_OK;
                            have_data = true;
                            out_data_offset[pc.file] = doff; out_data_bytes[pc.file] = dbytes; out_status[pc.file] = st;
                            if (cap > usable) std::memset(dst + usable, 0, cap - usable);   // rest of the slot: zeros
                            if (kStats) n_fast.fetch_add(1, std::memory_order_relaxed);
                        }
                    }
                }
            }
            if (!have_data) {
                if (fd >= 0) st = wav_probe_fd(fd, &doff, &dbytes);
                size_t usable = st == AFSK_WAV_OK ? ((size_t)dbytes & ~(size_t)1) : 0;
                if (usable > cap) { st = AFSK_WAV_SLOT; usable = 0; }          // the caller's slot is too small: left to the caller
                if (pc.slot_lo == 0) { out_data_offset[pc.file] = doff; out_data_bytes[pc.file] = dbytes; out_status[pc.file] = st; }
                const size_t lo = std::min(usable, pc.slot_lo), hi = std::min(usable, pc.slot_lo + pc.bytes);
                size_t have = 0;
                if (hi > lo) {
                    // a file that shrank between the walk and the read is that FILE's problem, not the batch's:
                    // zeroed slot, status AFSK_WAV_IO, the caller's fallback reader reports it
                    char* q = dst;
                    size_t left = hi - lo;
                    int64_t off = doff + (int64_t)lo;
                    while (left > 0) {
                        const ssize_t r = pread(fd, q, left, (off_t)off);
                        if (r <= 0) break;
                        q += r; off += r; left -= (size_t)r; have += (size_t)r;
                    }
                    if (left > 0) { out_status[pc.file] = AFSK_WAV_IO; have = 0; }
                }
                if (dst && pc.bytes > have) std::memset(dst + have, 0, pc.bytes - have);   // rest of the slot: zeros
            }
            if (kStats) { const long long t1 = now_ns(); ns_read.fetch_add(t1 - t0, std::memory_order_relaxed); t0 = t1; }
            if (fd >= 0) close(fd);
            if (kStats) ns_close.fetch_add(now_ns() - t0, std::memory_order_relaxed);
        }
        if (pc.win < nwin && remaining[pc.win].fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(hand_mu);               // last piece of the window: wake the sender
            cv_filled.notify_one();
        }
    };

    hipError_t herr = hipSuccess;
    const char* hwhat = "";
    auto coordinator = [&](bool have_workers) {
        size_t next_serial = 0;                       // without pool threads the caller fills the pieces itself
        size_t gap_scan = 0;
        for (size_t w = 0; w < nwin && herr == hipSuccess && !failed.load(); w++) {
            if (!have_workers)
                while (next_serial < npieces && pieces[next_serial].win <= w) fill_piece(next_serial++);
            {
                const long long t0 = kStats ? now_ns() : 0;
                std::unique_lock<std::mutex> lk(hand_mu);
                cv_filled.wait(lk, [&] { return remaining[w].load(std::memory_order_acquire) == 0 || failed.load(); });
                if (kStats) ns_wait_fill += now_ns() - t0;
            }
            if (failed.load()) break;
            const long long t1 = kStats ? now_ns() : 0;
            char* st = stage[w % (size_t)kSlots];
            // gaps of up to kGapFill bytes between two slots of the window travel with it as zeros
            size_t cur = 0;
            for (; gap_scan < npieces && pieces[gap_scan].win <= w; gap_scan++) {   // pieces are in window order
                const Piece& pc = pieces[gap_scan];
                if (pc.win != w || pc.bytes == 0) continue;
                if (pc.stage_off > cur) std::memset(st + cur, 0, pc.stage_off - cur);
                cur = std::max(cur, pc.stage_off + pc.bytes);
            }
            if (cur < wins[w].bytes) std::memset(st + cur, 0, wins[w].bytes - cur);   // (a window that is all gap)
            // windows alternate between two copy streams (disjoint device ranges: no order needed between them),
            // so the next transfer is already queued when one ends
            hipStream_t cs = copy_stream[w & 1];
            if (wins[w].bytes > 0) {
                herr = hipMemcpyAsync((char*)d_samples + wins[w].dev_b0, st, wins[w].bytes, hipMemcpyHostToDevice, cs);
                hwhat = "H2D samples";
                if (herr != hipSuccess) break;
            }
            herr = hipEventRecord(sent[w % (size_t)kSlots], cs);
            hwhat = "hipEventRecord";
            if (herr != hipSuccess) break;
            const long long t2 = kStats ? now_ns() : 0;
            if (kStats) ns_issue += t2 - t1;
            // keep two transfers queued: only when window w - 1 has left its staging buffer is that buffer handed
            // to the fillers of window w - 1 + kSlots
            if (w >= 1) {
                herr = hipEventSynchronize(sent[(w - 1) % (size_t)kSlots]);
                hwhat = "hipEventSynchronize";
                if (kStats) ns_wait_sent += now_ns() - t2;
                {
                    std::lock_guard<std::mutex> lk(hand_mu);
                    released = (long)(w + (size_t)kSlots);
                }
                cv_released.notify_all();
            }
        }
        if (herr != hipSuccess || failed.load()) failed.store(1);
        {
            std::lock_guard<std::mutex> lk(hand_mu);
            released = std::numeric_limits<long>::max();           // nobody waits any more
        }
        cv_released.notify_all();
        if (!have_workers)
            while (next_serial < npieces) fill_piece(next_serial++);
    };
    // as many fillers as can work on released windows at once, within the pool's size
    const unsigned width = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, npieces));
    io_pool().run_split(npieces, width, fill_piece, coordinator);
    if (herr != hipSuccess) rc = hip_fail(herr, hwhat);
    else if (io_failed_file.load() >= 0) rc = fail(AFSK_E_HOST, std::string("cannot read ") + paths[io_failed_file.load()]);
    for (int k = 0; k < 2; k++) {
        hipError_t e = hipStreamSynchronize(copy_stream[k]);  // nothing may still read the staging buffers
        if (e != hipSuccess && rc == AFSK_OK) rc = hip_fail(e, "hipStreamSynchronize");
    }
    if (kStats) {
        const double ms = 1e-6;
        std::fprintf(stderr, "afsk_wav_ingest: %zu files, %zu windows of %zu MiB x %d slots, %u fillers, %.2f ms total | sender: "
                     "wait-filled %.2f issue %.2f wait-sent %.2f ms | fillers (sum over threads): wait-release %.2f open %.2f "
                     "read %.2f close %.2f ms, one-preadv files %lld\n", n, nwin, kWindow >> 20, kSlots, width,
                     (now_ns() - t_begin) * ms, ns_wait_fill * ms, ns_issue * ms, ns_wait_sent * ms,
                     ns_wait_release.load() * ms, ns_open.load() * ms, ns_read.load() * ms, ns_close.load() * ms,
                     (long long)n_fast.load());
    }
    return rc;
}

int afsk_wav_ingest(const char* const* paths, int32_t n_files, const int64_t* slot_offset,
                    const int64_t* slot_samples, int16_t* d_samples, int64_t capacity_samples,
                    int64_t* out_data_offset, int64_t* out_data_bytes, int32_t* out_status) {
    return no_throw([&] {
        return wav_ingest_impl(paths, n_files, slot_offset, slot_samples, d_samples, capacity_samples,
                               out_data_offset, out_data_bytes, out_status);
    });
}

// ---- .wav egress at scale: device streams -> canonical RIFF/WAVE files (Transmitter.save for many payloads) ----
// The mirror image of afsk_wav_ingest: the calling thread copies windows of the device range into the pinned
// staging ring (D2H, two alternating streams) and hands every window to the pool as soon as its copy has
// completed; pool threads write the file pieces of the window -- one open / pwritev (header + data) / close for a
// file that lies inside one window -- and a staging buffer is reused once all its pieces are on their way.
static int wav_egress_impl(const char* const* paths, int32_t n_files, const int16_t* d_samples,
                           const int64_t* stream_offset, const int32_t* stream_len, int32_t* out_status) {
    if (n_files < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_files == 0) return AFSK_OK;
    if (!paths || !d_samples || !stream_offset || !stream_len || !out_status)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    const size_t n = (size_t)n_files;
    int64_t prev_end = 0;
    for (size_t s = 0; s < n; s++) {
        if (!paths[s] || stream_len[s] < 0 || stream_offset[s] < prev_end)
            return fail(AFSK_E_INVALID_ARG, "streams must be ascending and must not overlap");
        if ((int64_t)stream_len[s] > (int64_t)((0xFFFFFFFFll - 36) / 2))
            return fail(AFSK_E_INVALID_ARG, "a stream too long for a RIFF file");
        prev_end = stream_offset[s] + stream_len[s];
    }
    if (int rc0 = require_device()) return rc0;
    {
        cpu_set_t cpus;
        if (device_node_cpus(&cpus)) io_pool().confine_to(cpus);
    }
    static const size_t kWindow = (size_t)16 << 20;
    static const int kSlots = 8;
    int rc = AFSK_OK;
    hipStream_t copy_stream[2] = {nullptr, nullptr};
    {
        hipError_t e = g_thread_stream.get(&copy_stream[0]);
        if (e == hipSuccess) e = g_thread_stream2.get(&copy_stream[1]);
        if (e != hipSuccess) return hip_fail(e, "hipStreamCreateWithFlags (host-entry streams)");
    }
    ScratchLease lease;
    char* d_unused = nullptr;
    char* stage[kMaxRingSlots];
    hipEvent_t landed_ev[kMaxRingSlots];
    {
        hipError_t e = lease.acquire(1, &d_unused, /*block=*/true);    // owns the staging ring
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (host-entry scratch)");
        e = lease.ring(kWindow, kSlots, stage, landed_ev);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc / hipEventCreate (staging ring)");
    }
    // windows over the device byte range; a gap of more than 64 KiB between two streams is not copied
    struct Piece { size_t file; size_t win; size_t lo; size_t bytes; size_t stage_off; bool whole; };
    struct Window { size_t dev_b0; size_t bytes; };
    std::vector<Piece> pieces;
    std::vector<Window> wins;
    pieces.reserve(n + 8);
    {
        constexpr size_t kGapCopy = (size_t)64 << 10;
        size_t w_b0 = 0, w_b1 = 0;
        bool open_w = false;
        auto close_window = [&] { if (open_w) { wins.push_back({w_b0, w_b1 - w_b0}); open_w = false; } };
        auto win_cap = [&] { return std::min(kWindow, ((size_t)1 << 20) << std::min<size_t>(wins.size(), 6)); };
        for (size_t s = 0; s < n; s++) {
            const size_t b0 = (size_t)stream_offset[s] * 2, total = (size_t)stream_len[s] * 2;
            if (total == 0) {                         // an empty stream still becomes a (header-only) file
                if (!open_w) { w_b0 = w_b1 = b0; open_w = true; }
                pieces.push_back({s, wins.size(), 0, 0, 0, true});
                continue;
            }
            size_t done = 0;
            while (done < total) {
                const size_t p0 = b0 + done;
                if (open_w && (p0 > w_b1 + kGapCopy || p0 - w_b0 >= win_cap())) close_window();
                if (!open_w) { w_b0 = w_b1 = p0; open_w = true; }
                const size_t cap = win_cap();
                const size_t room = cap - (p0 - w_b0);
                const size_t take = std::min(total - done, room);
                if (take < total - done && done == 0 && total <= cap && p0 != w_b0) { close_window(); continue; }
                pieces.push_back({s, wins.size(), done, take, p0 - w_b0, take == total});
                done += take;
                w_b1 = p0 + take;
                if (w_b1 - w_b0 >= cap) close_window();
            }
        }
        close_window();
    }
    const size_t nwin = wins.size(), npieces = pieces.size();
    std::vector<std::atomic<int>> remaining(std::max<size_t>(nwin, 1));
    for (auto& r : remaining) r.store(0);
    for (const Piece& pc : pieces) if (pc.win < nwin) remaining[pc.win].fetch_add(1);
    for (size_t s = 0; s < n; s++) out_status[s] = AFSK_WAV_OK;
    std::mutex hand_mu;
    std::condition_variable cv_landed, cv_written;
    long landed = 0;                                        // windows whose D2H copy has completed
    std::atomic<int> failed{0};

    auto header = [](unsigned char* h, uint32_t data_bytes) {   // what wave.Wave_write emits for 1 ch / 16 bit / 48000 Hz
        auto p32 = [](unsigned char* q, uint32_t v) { q[0] = (unsigned char)v; q[1] = (unsigned char)(v >> 8); q[2] = (unsigned char)(v >> 16); q[3] = (unsigned char)(v >> 24); };
        auto p16 = [](unsigned char* q, uint32_t v) { q[0] = (unsigned char)v; q[1] = (unsigned char)(v >> 8); };
        std::memcpy(h, "RIFF", 4); p32(h + 4, 36u + data_bytes); std::memcpy(h + 8, "WAVEfmt ", 8); p32(h + 16, 16u);
        p16(h + 20, 1u); p16(h + 22, 1u); p32(h + 24, AFSK_SAMPLE_RATE); p32(h + 28, AFSK_SAMPLE_RATE * 2u); p16(h + 32, 2u);
        p16(h + 34, 16u); std::memcpy(h + 36, "data", 4); p32(h + 40, data_bytes);
    };
    auto write_piece = [&](size_t i) {
        const Piece& pc = pieces[i];
        if (pc.win < nwin) {
            std::unique_lock<std::mutex> lk(hand_mu);
            cv_landed.wait(lk, [&] { return (long)pc.win < landed || failed.load(std::memory_order_relaxed); });
        }
        if (!failed.load(std::memory_order_relaxed)) {
            const uint32_t total = (uint32_t)stream_len[pc.file] * 2u;
            // No O_TRUNC: an existing file is overwritten IN PLACE (its page-cache pages are reused instead of freed and
            // allocated again) and every piece sets the final size -- the pieces of a file that spans windows are
            // written concurrently, in any order, and ftruncate to the size a file already has costs nothing.
            const int fd = open(paths[pc.file], O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
            bool ok = fd >= 0;
            if (ok && !pc.whole) ok = ftruncate(fd, (off_t)(44 + (int64_t)total)) == 0;
            if (ok) {
                unsigned char hdr[44];
                struct iovec iov[2];
                int niov = 0;
                int64_t off = 44 + (int64_t)pc.lo;
                if (pc.lo == 0) { header(hdr, total); iov[niov++] = {hdr, sizeof hdr}; off = 0; }
                if (pc.bytes > 0) iov[niov++] = {stage[pc.win % (size_t)kSlots] + pc.stage_off, pc.bytes};
                size_t want = (pc.lo == 0 ? 44 : 0) + pc.bytes;
                while (ok && want > 0) {                          // pwritev may be partial
                    const ssize_t r = pwritev(fd, iov, niov, (off_t)off);
                    if (r <= 0) { ok = false; break; }
                    want -= (size_t)r; off += r;
                    size_t adv = (size_t)r;
                    while (adv > 0 && niov > 0) {
                        if (adv >= iov[0].iov_len) { adv -= iov[0].iov_len; iov[0] = iov[1]; niov--; }
                        else { iov[0].iov_base = (char*)iov[0].iov_base + adv; iov[0].iov_len -= adv; adv = 0; }
                    }
                }
            }
            if (ok && pc.whole) ok = ftruncate(fd, (off_t)(44 + (int64_t)total)) == 0;   // (an existing longer file)
            if (fd >= 0 && close(fd) != 0) ok = false;
            if (!ok) out_status[pc.file] = AFSK_WAV_IO;           // that file's problem; the batch goes on
        }
        if (pc.win < nwin && remaining[pc.win].fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(hand_mu);
            cv_written.notify_one();
        }
    };

    hipError_t herr = hipSuccess;
    const char* hwhat = "";
    auto coordinator = [&](bool have_workers) {
        size_t issued = 0, next_serial = 0;
        for (size_t w = 0; w < nwin && herr == hipSuccess; w++) {
            for (;;) {                                            // keep copies ahead of the writers, one per free buffer
                if (issued >= nwin || issued >= w + (size_t)kSlots) break;
                if (issued >= (size_t)kSlots) {
                    const size_t dep = issued - (size_t)kSlots;   // the window whose buffer this copy reuses (dep < w: released)
                    if (remaining[dep].load(std::memory_order_acquire) != 0) {
                        if (issued > w) break;                    // window w itself is on its way: look again later
                        if (!have_workers) while (next_serial < npieces && pieces[next_serial].win <= dep) write_piece(next_serial++);
                        std::unique_lock<std::mutex> lk(hand_mu);
                        cv_written.wait(lk, [&] { return remaining[dep].load(std::memory_order_acquire) == 0; });
                    }
                }
                hipStream_t cs = copy_stream[issued & 1];
                if (wins[issued].bytes > 0) {
                    herr = hipMemcpyAsync(stage[issued % (size_t)kSlots], (const char*)d_samples + wins[issued].dev_b0,
                                          wins[issued].bytes, hipMemcpyDeviceToHost, cs);
                    hwhat = "D2H samples";
                    if (herr != hipSuccess) break;
                }
                herr = hipEventRecord(landed_ev[issued % (size_t)kSlots], cs);
                hwhat = "hipEventRecord";
                if (herr != hipSuccess) break;
                issued++;
            }
            if (herr != hipSuccess) break;
            herr = hipEventSynchronize(landed_ev[w % (size_t)kSlots]);
            hwhat = "hipEventSynchronize";
            if (herr != hipSuccess) break;
            {
                std::lock_guard<std::mutex> lk(hand_mu);
                landed = (long)w + 1;
            }
            cv_landed.notify_all();
            if (!have_workers) while (next_serial < npieces && pieces[next_serial].win <= w) write_piece(next_serial++);
        }
        if (herr != hipSuccess) failed.store(1);
        {
            std::lock_guard<std::mutex> lk(hand_mu);
            landed = std::numeric_limits<long>::max();
        }
        cv_landed.notify_all();
        if (!have_workers) while (next_serial < npieces) write_piece(next_serial++);
    };
    const unsigned width = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, npieces));
    io_pool().run_split(npieces, width, write_piece, coordinator);
    if (herr != hipSuccess) rc = hip_fail(herr, hwhat);
    for (int k = 0; k < 2; k++) {
        hipError_t e = hipStreamSynchronize(copy_stream[k]);
        if (e != hipSuccess && rc == AFSK_OK) rc = hip_fail(e, "hipStreamSynchronize");
    }
    return rc;
}

int afsk_wav_egress(const char* const* paths, int32_t n_files, const int16_t* d_samples,
                    const int64_t* stream_offset, const int32_t* stream_len, int32_t* out_status) {
    return no_throw([&] { return wav_egress_impl(paths, n_files, d_samples, stream_offset, stream_len, out_status); });
}

int afsk_modulate_batch(const uint8_t* payload, int32_t payload_stride,
                        const int32_t* payload_len, const int32_t* bit_frames,
                        const int32_t* ts_cycles, const int64_t* stream_offset,
                        const int32_t* stream_len, int32_t max_stream_len, int32_t n_streams,
                        int32_t wav_quirk, int16_t* samples, void* hip_stream) {
    if (n_streams < 0 || payload_stride < 0 || max_stream_len < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0 || max_stream_len == 0) return AFSK_OK;
    if (!payload_len || !bit_frames || !ts_cycles || !stream_offset || !stream_len || !samples ||
        (!payload && payload_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::ModulateArgs a;
    a.payload = payload; a.payload_stride = payload_stride; a.payload_len = payload_len;
    a.bit_frames = bit_frames; a.ts_cycles = ts_cycles; a.stream_offset = stream_offset;
    a.stream_len = stream_len; a.n_streams = n_streams; a.wav_quirk = wav_quirk;
    a.samples = samples; a.chunks = 0;
    hipError_t e = afsk::launch_modulate(a, max_stream_len, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch modulate_kernel");
}

static int gate_batch_impl(const int16_t* samples, const int64_t* stream_offset,
                           const int32_t* stream_len, int32_t max_stream_len,
                           int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                           int32_t max_bursts, int32_t* block_amp, int32_t* out_n_bursts,
                           int32_t* out_burst_start, int32_t* out_burst_len, int32_t* out_open_end,
                           int64_t* out_slot_offset, int32_t* out_slot_len, void* hip_stream) {
    if (n_streams < 0 || max_stream_len < 0 || max_bursts < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    const int32_t max_blocks = max_stream_len / 2048;
    if (!samples || !stream_offset || !stream_len || !out_n_bursts || !out_open_end ||
        (max_blocks > 0 && !block_amp) || (max_bursts > 0 && (!out_burst_start || !out_burst_len)))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if ((out_slot_offset == nullptr) != (out_slot_len == nullptr))
        return fail(AFSK_E_INVALID_ARG, "out_slot_offset and out_slot_len go together");
    if (int rc = require_device()) return rc;
    afsk::GateArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.amp_start = amp_start_threshold; a.amp_end = amp_end_threshold; a.n_streams = n_streams;
    a.max_len = max_stream_len;
    a.max_blocks = max_blocks; a.max_bursts = max_bursts; a.block_amp = block_amp;
    a.out_n_bursts = out_n_bursts; a.out_burst_start = out_burst_start;
    a.out_burst_len = out_burst_len; a.out_open_end = out_open_end;
    a.out_slot_offset = max_bursts > 0 ? out_slot_offset : nullptr;
    a.out_slot_len = max_bursts > 0 ? out_slot_len : nullptr;
    hipError_t e = afsk::launch_gate(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch gate kernels");
}

int afsk_gate_batch(const int16_t* samples, const int64_t* stream_offset,
                    const int32_t* stream_len, int32_t max_stream_len,
                    int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                    int32_t max_bursts, int32_t* block_amp, int32_t* out_n_bursts,
                    int32_t* out_burst_start, int32_t* out_burst_len, int32_t* out_open_end,
                    void* hip_stream) {
    return gate_batch_impl(samples, stream_offset, stream_len, max_stream_len, amp_start_threshold, amp_end_threshold,
                           n_streams, max_bursts, block_amp, out_n_bursts, out_burst_start, out_burst_len, out_open_end,
                           nullptr, nullptr, hip_stream);
}

int afsk_gate_batch_slots(const int16_t* samples, const int64_t* stream_offset,
                          const int32_t* stream_len, int32_t max_stream_len,
                          int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                          int32_t max_bursts, int32_t* block_amp, int32_t* out_n_bursts,
                          int32_t* out_burst_start, int32_t* out_burst_len, int32_t* out_open_end,
                          int64_t* out_slot_offset, int32_t* out_slot_len, void* hip_stream) {
    if (max_bursts > 0 && (!out_slot_offset || !out_slot_len)) return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    return gate_batch_impl(samples, stream_offset, stream_len, max_stream_len, amp_start_threshold, amp_end_threshold,
                           n_streams, max_bursts, block_amp, out_n_bursts, out_burst_start, out_burst_len, out_open_end,
                           out_slot_offset, out_slot_len, hip_stream);
}

int afsk_add_noise_batch(int16_t* samples, const int64_t* stream_offset,
                         const int32_t* stream_len, int32_t max_stream_len,
                         const int32_t* scale_q24, int32_t n_streams, uint32_t seed,
                         uint32_t stream_idx_base, void* hip_stream) {
    if (n_streams < 0 || max_stream_len < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0 || max_stream_len == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !scale_q24)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::NoiseArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.scale_q24 = scale_q24; a.n_streams = n_streams; a.seed = seed;
    a.stream_idx_base = stream_idx_base; a.chunks = 0;
    hipError_t e = afsk::launch_noise(a, max_stream_len, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch noise_kernel");
}

}  // extern "C"
