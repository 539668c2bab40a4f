// hip_stub_runtime.cpp -- TEST INFRASTRUCTURE ONLY: a fake HIP runtime for CPU-side tests of afsk_capi.hip's
// HOST logic (the staging ring of afsk_wav_ingest, the window packing of afsk_wav_upload /
// afsk_demod_streams_host, scratch leases, thread hand-over) under ThreadSanitizer / AddressSanitizer and in
// the plain CPU suite.  "Device memory" is host memory; every stream is an in-order worker thread, so
// hipMemcpyAsync really is asynchronous and a staging buffer handed back to the fillers before its copy has
// run IS a data race the sanitizer sees.  Linked (with -Bsymbolic) into a test-only build of the library
// together with stub kernel launchers (tools/capi_asan.sh, tests/test_capi_host_logic.py); it is never part of
// libafsk_amd.so and nothing in afskmodem_amd/ can reach it.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace {

struct Event {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long recorded = 0, completed = 0;      // record / completion counters
};

struct Stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool quit = false, busy = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        th.join();
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || !q.empty(); });
            if (q.empty()) return;
            auto fn = std::move(q.front());
            q.pop_front();
            busy = true;
            lk.unlock();
            fn();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void push(std::function<void()> fn) {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(fn)); }
        cv.notify_all();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.empty() && !busy; });
    }
};

Stream& null_stream() { static Stream* s = new Stream(); return *s; }
Stream& of(hipStream_t s) { return s ? *reinterpret_cast<Stream*>(s) : null_stream(); }

}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int*, hipDeviceAttribute_t, int) { return hipErrorInvalidValue; }   // NUMA node unknown:
hipError_t hipDeviceGetPCIBusId(char*, int, int) { return hipErrorInvalidValue; }                    // no binding
const char* hipGetErrorString(hipError_t) { return "stub runtime error"; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { null_stream().drain(); std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
    of(st).push([=] { std::memcpy(d, s, n); });
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = reinterpret_cast<hipStream_t>(new Stream()); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<Stream*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { of(s).drain(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { *e = reinterpret_cast<hipEvent_t>(new Event()); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Event*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    Event* ev = reinterpret_cast<Event*>(e);
    unsigned long long ticket;
    { std::lock_guard<std::mutex> lk(ev->mu); ticket = ++ev->recorded; }
    of(s).push([ev, ticket] {
        { std::lock_guard<std::mutex> lk(ev->mu); if (ev->completed < ticket) ev->completed = ticket; }
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    Event* ev = reinterpret_cast<Event*>(e);
    std::unique_lock<std::mutex> lk(ev->mu);
    const unsigned long long want = ev->recorded;          // the most recent record, like the real call
    ev->cv.wait(lk, [&] { return ev->completed >= want; });
    return hipSuccess;
}

}  // extern "C"
