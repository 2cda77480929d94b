// common.hpp — context, error plumbing, device buffers and kernel-launch timing for libcomet_hip.
// gfx950 only; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/comet_gpu.h"

namespace comet {

// thread-local error text (comet_last_error)
inline std::string& last_error() { static thread_local std::string e; return e; }
inline int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    last_error() = buf;
    return code;
}

struct HipError { hipError_t e; const char* what; const char* file; int line; };
#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) throw ::comet::HipError{_e, #expr, __FILE__, __LINE__};        \
    } while (0)

struct StatusError { int code; };  // thrown after set_error()
#define COMET_FAIL(code, ...) throw ::comet::StatusError{::comet::set_error((code), __VA_ARGS__)}

// Wrap a C-ABI body: translate exceptions into status codes.
template <class F> inline int guarded(F&& f) {
    try { return f(); }
    catch (const StatusError& s) { return s.code; }
    catch (const HipError& h) {
        return set_error(COMET_ERR_HIP, "HIP error %d (%s) in %s at %s:%d", (int)h.e, hipGetErrorString(h.e), h.what, h.file, h.line);
    }
    catch (const std::bad_alloc&) { return set_error(COMET_ERR_HIP, "host allocation failed"); }
    catch (const std::exception& e) { return set_error(COMET_ERR_INVALID_ARG, "%s", e.what()); }
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

struct Ctx;

// Growable device allocation owned by an index (doubling growth, preserves contents).
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t bytes, hipStream_t s, size_t keep_bytes) {
        if (bytes <= cap) return;
        size_t ncap = cap ? cap : 4096;
        while (ncap < bytes) ncap = ncap + ncap / 2 + 4096;
        void* np = nullptr;
        HIP_CHECK(hipMalloc(&np, ncap));
        if (p && keep_bytes) { HIP_CHECK(hipMemcpyAsync(np, p, keep_bytes, hipMemcpyDeviceToDevice, s)); HIP_CHECK(hipStreamSynchronize(s)); }
        if (p) HIP_CHECK(hipFree(p));
        p = np; cap = ncap;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <class T> T* as() const { return (T*)p; }
};

struct ProfEntry { double ms = 0; int64_t n = 0; };

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    std::recursive_mutex mu;  // serialises calls that share the scratch arena / stream
    // scratch arena: bump allocator over one device allocation, reset at the start of every call.
    void* scratch = nullptr; size_t scratch_cap = 0, scratch_off = 0;
    std::vector<void*> retired;  // old arenas kept alive until the call that outgrew them finishes
    // Execution lanes (round 3). Two batches in flight on ONE stream do not overlap on the GPU, and a search step is a chain of
    // short latency-bound kernels around one big scan: with every other asynchronous search on a second stream (and a scratch arena
    // of its own — the arena is recycled in stream order) the small kernels of batch i+1 run beside batch i's post stage
    // (tools/two_ctx_probe.py: IVF 854 k -> 1.14 M q/s, IVFPQ 843 k -> 1.29 M, Flat 783 k -> 883 k with two contexts).
    // Up to four lanes: index kinds whose step has no kernel that fills the GPU (HNSW: 256 waves per search; IVFPQ) keep gaining up to four
    // searches in flight (tools/two_ctx_probe.py, one stream per context: HNSW 672 k / 1.19 M / 1.60 M / 1.89 M q/s with 1 / 2 / 3 / 4, IVFPQ
    // 828 k / 1.30 M / 1.56 M / 1.69 M; IVF and Flat peak at two: 858 k / 1.16 M / 0.99 M, 800 k / 895 k / 861 k) — comet_index::max_lanes().
    // `stream` / `scratch*` / `retired` always describe the CURRENT lane; `parked` holds the others'. Lane 0 is current whenever no
    // asynchronous search is being enqueued; everything that is not such a search first waits for the other lanes (quiesce_alt).
    static constexpr int kMaxLanes = 8;      // (round 6: eight — an HNSW search is one wave per query, 8 x 256 queries are two waves per SIMD; the other kinds stop at comet_index::max_lanes())
    struct LaneState { hipStream_t stream = nullptr; void* scratch = nullptr; size_t scratch_cap = 0, scratch_off = 0; std::vector<void*> retired; };
    LaneState parked[kMaxLanes]; // parked[l]: lane l's state while another lane is current (parked[cur_lane] is stale)
    int cur_lane = 0;
    unsigned dirty_mask = 0;     // bit l: work was enqueued on lane l >= 1 since it was last synchronised
    int lanes = [] { const char* e = getenv("COMET_LANES"); const int v = e ? atoi(e) : kMaxLanes; return v < 1 ? 1 : (v > kMaxLanes ? kMaxLanes : v); }();
    void switch_lane(int l) {
        if (l == cur_lane) return;
        LaneState& p = parked[cur_lane];
        p.stream = stream; p.scratch = scratch; p.scratch_cap = scratch_cap; p.scratch_off = scratch_off; p.retired = std::move(retired);
        LaneState& n = parked[l];
        if (!n.stream) HIP_CHECK(hipStreamCreateWithFlags(&n.stream, hipStreamNonBlocking));
        stream = n.stream; scratch = n.scratch; scratch_cap = n.scratch_cap; scratch_off = n.scratch_off; retired = std::move(n.retired); n.retired.clear();
        cur_lane = l;
    }
    void mark_dirty() { if (cur_lane) dirty_mask |= 1u << cur_lane; }
    // Ordering between lane 0 and the other lanes. Lane 0's stream is the one comet_ctx_stream() hands out and the one every call that is
    // not an asynchronous search enqueues on (uploads, comet_synth_*, adds ...). Such a call may return with its work still queued; an
    // asynchronous search that lands on lane 1..3 must start behind it. `lane0_fence` is recorded on lane 0 when a non-search call ends
    // (CallGuard) or when the caller asks for it after enqueueing its own kernels on comet_ctx_stream() (comet_ctx_fence); a search on
    // another lane waits for the newest record. Searches on lane 0 do NOT move the fence: a search on lane 1 must not queue up behind the
    // search that lane 0 is running (that would serialise the lanes).
    hipEvent_t lane0_fence = nullptr; bool lane0_fence_set = false;
    bool async_seen = false;        // an asynchronous search has been issued on this context: only then do non-search calls record the fence (a context that is only
                                    // ever used through the blocking calls re-recorded the event after every call for nobody — thousands of records on one event)
    hipEvent_t fork_ev = nullptr;   // segments_search: the other lanes start behind what lane 0 holds at the call (owned by the context: an event belongs to a device)
    void fence_lane0() {            // lane 0 current
        if (!lane0_fence) HIP_CHECK(hipEventCreateWithFlags(&lane0_fence, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(lane0_fence, stream));
        lane0_fence_set = true;
    }
    void follow_lane0() {           // the current lane (>= 1) starts behind the newest fence of lane 0
        if (cur_lane != 0 && lane0_fence_set) HIP_CHECK(hipStreamWaitEvent(stream, lane0_fence, 0));
    }
    void quiesce_alt() {         // called with lane 0 current: the other lanes idle
        for (int l = 1; l < kMaxLanes; l++) if (((dirty_mask >> l) & 1u) && parked[l].stream) HIP_CHECK(hipStreamSynchronize(parked[l].stream));
        dirty_mask = 0;
    }
    void quiesce_all() {         // every lane idle (whichever is current)
        HIP_CHECK(hipStreamSynchronize(stream));
        for (int l = 0; l < kMaxLanes; l++) if (l != cur_lane && parked[l].stream) HIP_CHECK(hipStreamSynchronize(parked[l].stream));
        dirty_mask = 0;
    }
    // the live indexes of this context and their integrity check (index.hpp: comet_index::guards_ok); set by api.hip
    std::vector<void*> live_indexes;
    void (*check_live)(Ctx*, const char*) = nullptr;
    void check_indexes(const char* where) { if (check_live) check_live(this, where); }
    // pinned host staging for small readbacks
    void* pinned = nullptr; size_t pinned_cap = 0;
    // profiling
    bool profile = false;
    std::string prof_only;       // non-empty: only the scopes named in this '|'-separated list are timed (event records are barrier packets: ~5 us each in a chain of short kernels)
    bool prof_open = false;
    struct Pending { std::string name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::map<std::string, ProfEntry> prof;

    void bind() { HIP_CHECK(hipSetDevice(device)); }
    void scratch_reset() {
        scratch_off = 0;
        for (void* r : retired) (void)hipFree(r);
        retired.clear();
    }
    // 256-byte aligned scratch; contents undefined. Pointers stay valid until the next scratch_reset().
    void* scratch_alloc(size_t bytes) {
        size_t off = (scratch_off + 255) & ~(size_t)255;
        if (off + bytes > scratch_cap) {
            // outgrown: allocate a bigger arena; keep the old one alive (earlier pointers of this call remain valid)
            size_t ncap = scratch_cap ? scratch_cap * 2 : ((size_t)64 << 20);
            while (ncap < bytes + 256) ncap *= 2;
            if (scratch) retired.push_back(scratch);
            HIP_CHECK(hipMalloc(&scratch, ncap));
            scratch_cap = ncap; off = 0;
        }
        scratch_off = off + bytes;
        return (char*)scratch + off;
    }
    template <class T> T* salloc(size_t n) { return (T*)scratch_alloc(n * sizeof(T)); }
    void* pinned_buf(size_t bytes) {
        if (bytes > pinned_cap) {
            if (pinned) (void)hipHostFree(pinned);
            size_t ncap = pinned_cap ? pinned_cap : 4096; while (ncap < bytes) ncap *= 2;
            HIP_CHECK(hipHostMalloc(&pinned, ncap, hipHostMallocDefault)); pinned_cap = ncap;
        }
        return pinned;
    }
    void sync() { quiesce_all(); collect_profile(); }
    // Host <-> device copies between device memory and memory the library does not own (callers' arrays, std::vector storage, stack words) go through a PINNED bounce
    // buffer of the context with a stream synchronisation per piece: the HIP runtime's own staging of pageable asynchronous copies is not used, and the GPU never reads
    // or writes pageable memory. Why: round 6's soaks under HSA_ENABLE_SDMA=0 (copies by shader blits) met wrong results — an index whose rows in one
    // window were not the rows that were added (the same wrong answer from every kernel, again and again), a result row that was wrong once and right on the next
    // identical call, a wrong answer of the CPU ORACLE (host memory that changed under it) — one pass in four on some boxes, with nothing but pageable hipMemcpyAsync
    // between the host arrays and the kernels; and round 4's crash was eight bytes of 0xFF in a heap object (the padding of a result row is 0xFFFFFFFF ids).
    // COMET_COPY_DIRECT=1 restores the direct pageable hipMemcpyAsync (what every round up to 5 did).
    void* bounce = nullptr; static constexpr size_t kBounce = (size_t)8 << 20;
    static bool copy_direct() { static const bool d = getenv("COMET_COPY_DIRECT") != nullptr; return d; }
    void h2d(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        if (copy_direct()) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); return; }
        if (!bounce) HIP_CHECK(hipHostMalloc(&bounce, kBounce, hipHostMallocDefault));
        for (size_t off = 0; off < bytes; off += kBounce) {
            const size_t m = std::min(kBounce, bytes - off);
            std::memcpy(bounce, (const char*)src + off, m);
            HIP_CHECK(hipMemcpyAsync((char*)dst + off, bounce, m, hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));              // the bounce buffer is free again (and the caller's memory was read on the host, now)
        }
    }
    // (device -> host: the same bounce buffer, the host memcpy behind the synchronisation; the copy is SYNCHRONOUS — a caller's own hipStreamSynchronize behind it finds an idle stream)
    void d2h(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        if (copy_direct()) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream)); return; }
        if (!bounce) HIP_CHECK(hipHostMalloc(&bounce, kBounce, hipHostMallocDefault));
        for (size_t off = 0; off < bytes; off += kBounce) {
            const size_t m = std::min(kBounce, bytes - off);
            HIP_CHECK(hipMemcpyAsync(bounce, (const char*)src + off, m, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            std::memcpy((char*)dst + off, bounce, m);
        }
    }
    void d2d(void* dst, const void* src, size_t bytes) { if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream)); }
    void zero(void* dst, size_t bytes) { if (bytes) HIP_CHECK(hipMemsetAsync(dst, 0, bytes, stream)); }

    // prof_only: "" = every scope, otherwise the scopes named in a '|'-separated list
    bool prof_wants(const char* name) const {
        if (prof_only.empty()) return true;
        const size_t ln = std::strlen(name);
        for (size_t a = 0; a <= prof_only.size();) {
            size_t b = prof_only.find('|', a); if (b == std::string::npos) b = prof_only.size();
            if (b - a == ln && prof_only.compare(a, ln, name) == 0) return true;
            a = b + 1;
        }
        return false;
    }
    void prof_begin(const char* name) {
        prof_open = profile && prof_wants(name);
        if (!prof_open) return;
        Pending p; p.name = name;
        HIP_CHECK(hipEventCreate(&p.a)); HIP_CHECK(hipEventCreate(&p.b));
        HIP_CHECK(hipEventRecord(p.a, stream));
        pending.push_back(p);
    }
    void prof_end() {
        if (!prof_open) return;
        prof_open = false;
        (void)hipEventRecord(pending.back().b, stream);
    }
    // Launch a kernel under a profile scope name WITHOUT the two event-record packets of ProfScope: when the scope is being timed the
    // start / stop events ride on the kernel's own dispatch (hipExtLaunchKernelGGL), so the timed regions of a bench see the same
    // dependency chain as an untimed run (an event record is a barrier packet: ~5 us each between short kernels) and the measured time
    // is the kernel's, not the kernel plus the wait for its predecessor.
    template <class K, class... A>
    void launch_timed(const char* name, K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
        if (profile && prof_wants(name)) {
            Pending p; p.name = name;
            HIP_CHECK(hipEventCreate(&p.a)); HIP_CHECK(hipEventCreate(&p.b));
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, p.a, p.b, 0, args...);
            pending.push_back(p);
        } else {
            hipLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, stream, args...);
        }
    }
    void collect_profile() {
        for (auto& p : pending) {
            float ms = 0;
            if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                auto& e = prof[p.name]; e.ms += ms; e.n += 1;
            }
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }
};

// RAII: release scratch taken inside a scope (same-stream ordering makes immediate reuse safe).
struct ScratchMark {
    Ctx* c; void* arena; size_t off;
    explicit ScratchMark(Ctx* c_) : c(c_), arena(c_->scratch), off(c_->scratch_off) {}
    ~ScratchMark() { if (c->scratch == arena) c->scratch_off = off; }
};

// RAII: time one kernel launch under a name when profiling is on.
struct ProfScope {
    Ctx* c;
    ProfScope(Ctx* c_, const char* name) : c(c_) { c->prof_begin(name); }
    ~ProfScope() { c->prof_end(); }
};

// math.Log as Go's portable implementation computes it (FreeBSD e_log.c; src/math/log.go) — restated so
// idf matches the reference bit for bit instead of depending on the host libm.
inline double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (std::isnan(x) || (std::isinf(x) && x > 0)) return x;
    if (x < 0) return std::nan("");
    if (x == 0) return -INFINITY;
    int ki; double f1 = std::frexp(x, &ki);
    if (f1 < 0.70710678118654752440) { f1 *= 2; ki--; }
    const double f = f1 - 1, k = (double)ki;
    const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2, hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

#define LAUNCH_CHECK() HIP_CHECK(hipGetLastError())

}  // namespace comet
