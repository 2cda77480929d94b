// comm.hip — the multi-GPU exchange step behind the C ABI: RCCL over xGMI, one process per GPU.
//
// Reference analogue: the per-segment fan-out + mergeResults of the persistent layer (storage.go:546-626,
// storage_merge.go:13-46), which also searches independent parts and merges their top-K lists. Here the parts are
// index shards on different GPUs: every rank searches its shard for the SAME query batch, the per-shard
// (ids | scores | counts) blocks are exchanged with ONE ncclAllGather per batch, and every rank merges the R
// sorted lists per query (merge_topk_kernel) — exact, because the top-K of a union is the top-K of the parts'
// top-Ks, and identical to the unsharded canonical order as long as lower ranks hold earlier scan positions.
//
// The exchange runs on a second HIP stream of the communicator: search(i+1) on the context's stream overlaps
// all-gather(i) + merge(i); nothing on this path blocks the host except the explicit waits the caller asks for.
// librccl.so.1 is bound at run time (dlopen) the first time a communicator is made, so single-GPU users of the
// library never load it.
#include <dlfcn.h>

#include <atomic>

#include "index.hpp"

namespace comet {

// the few RCCL entry points used (signatures from rccl.h; ncclComm_t / ncclResult_t as opaque pointer / int)
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId { char internal[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
constexpr int NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_MAX = 2, NCCL_MIN = 3, NCCL_SUM = 0;

static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string why;
    std::call_once(once, [] {
        // COMET_RCCL_LIB: another library with the same five entry points (tests/shm_rccl.cpp runs the ranks of a communicator as
        // processes sharing one GPU over POSIX shared memory — real RCCL refuses several ranks on one device)
        const char* env = getenv("COMET_RCCL_LIB");
        const char* names[] = {env && *env ? env : "librccl.so.1", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (int i = 0; i < (env && *env ? 1 : 4) && !r.h; i++) {
            r.h = dlopen(names[i], RTLD_NOW | (env && *env ? RTLD_LOCAL : RTLD_GLOBAL));
            if (!r.h) { const char* e = dlerror(); why = e ? e : "dlopen failed"; }     // dlerror() clears the state: read it once
        }
        if (!r.h) return;
        r.GetUniqueId = (int (*)(void*))dlsym(r.h, "ncclGetUniqueId");
        r.CommDestroy = (int (*)(void*))dlsym(r.h, "ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.h, "ncclAllGather");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.h, "ncclAllReduce");
        r.GetErrorString = (const char* (*)(int))dlsym(r.h, "ncclGetErrorString");
    });
    if (!r.h || !r.GetUniqueId || !r.AllGather || !r.AllReduce) COMET_FAIL(COMET_ERR_UNSUPPORTED, "RCCL (librccl.so.1) is not available: %s", r.h ? "missing symbols" : why.c_str());
    return r;
}
#define RCCL_CHECK(expr) do { int _r = (expr); if (_r != 0) COMET_FAIL(COMET_ERR_HIP, "RCCL error %d (%s) in %s", _r, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?", #expr); } while (0)

// run launches of a scope on another stream of the same context (calls are serialised by the context mutex)
struct StreamSwap {
    Ctx* c; hipStream_t saved;
    StreamSwap(Ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};

}  // namespace comet

using namespace comet;

struct comet_comm {
    Ctx* c = nullptr; int rank = 0, world = 1;
    void* comm = nullptr;
    hipStream_t xstream = nullptr;       // exchange + merge (+ the bound all-reduce of sharded IVFPQ searches): every collective is issued here
    hipEvent_t bound_a = nullptr, bound_b = nullptr;
    int exch_done = 0;                   // bound exchanges issued so far by the search being enqueued (a failing search owes its peers the rest)
    DevBuf idle_tq;                      // +inf bounds a failed search contributes
    uint64_t uid = 0;                    // never reused (a later communicator may be allocated at this one's address): what an index remembers its placement check by
    DevBuf scalar;                       // small device scratch for barrier / all-reduce
    double* hscalar = nullptr;           // pinned: the all-reduced value comes back here (the GPU does not write stack or heap memory, common.hpp: Ctx::d2h)
    struct Slot {
        bool active = false; uint64_t ticket = 0, search_ticket = 0;
        comet_index* idx = nullptr; int B = 0, k_cap = 0, k = 0;
        DevBuf pack, gathered;           // this rank's block [B*k_cap ids | B*k_cap scores | B counts], and the R gathered blocks
        DevBuf mergews;                  // the merge's own workspace (beyond 8192 candidates per query it sorts in global memory): the merge runs on the
                                         // exchange stream while the context's stream is already recycling the scratch arena for the next search
        uint32_t* out_ids = nullptr; float* out_scores = nullptr; int32_t* out_counts = nullptr;
        hipEvent_t searched = nullptr, merged = nullptr;
        int failed = 0; std::string why; // the local search threw after the slot was taken: this rank still joins every collective of the batch (its block carries
                                         // counts = -code, which the merge hands to every rank), _wait reports the error here
    };
    static constexpr int kSlots = 4;
    Slot slots[kSlots];
    uint64_t next_ticket = 1;
};

extern "C" {

int comet_comm_unique_id(uint8_t* out_id128) {
    return guarded([&] {
        UniqueId id; std::memset(&id, 0, sizeof(id));
        RCCL_CHECK(rccl().GetUniqueId(&id));
        std::memcpy(out_id128, &id, sizeof(id));
        return (int)COMET_OK;
    });
}

int comet_comm_create(comet_ctx* c, const uint8_t* id128, int32_t rank, int32_t world, comet_comm** out) {
    return guarded([&] {
        *out = nullptr;
        if (world <= 0 || rank < 0 || rank >= world || !id128) COMET_FAIL(COMET_ERR_INVALID_ARG, "bad communicator shape: rank %d of %d", rank, world);
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        Rccl& r = rccl();
        // ncclCommInitRank takes the 128-byte id by value: on the SysV x86-64 ABI a struct that large travels in memory, i.e.
        // exactly like the dereferenced pointer the typed declaration below passes
        using InitFn = int (*)(void**, int, UniqueId, int);
        InitFn init = (InitFn)dlsym(r.h, "ncclCommInitRank");
        if (!init) COMET_FAIL(COMET_ERR_UNSUPPORTED, "ncclCommInitRank not found in librccl");
        UniqueId id; std::memcpy(&id, id128, sizeof(id));
        auto* cm = new comet_comm();
        static std::atomic<uint64_t> next_uid{1};
        cm->c = c; cm->rank = rank; cm->world = world; cm->uid = next_uid++;
        int rc = init(&cm->comm, world, id, rank);
        if (rc != 0) { delete cm; COMET_FAIL(COMET_ERR_HIP, "ncclCommInitRank failed: %d (%s)", rc, r.GetErrorString ? r.GetErrorString(rc) : "?"); }
        HIP_CHECK(hipStreamCreateWithFlags(&cm->xstream, hipStreamNonBlocking));
        cm->scalar.reserve(256, c->stream, 0);
        *out = cm;
        return (int)COMET_OK;
    });
}

int comet_comm_destroy(comet_comm* cm) {
    return guarded([&] {
        if (!cm) return (int)COMET_OK;
        Ctx* c = cm->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        c->quiesce_all(); (void)hipStreamSynchronize(cm->xstream);
        for (auto& s : cm->slots) { if (s.searched) (void)hipEventDestroy(s.searched); if (s.merged) (void)hipEventDestroy(s.merged); }
        if (cm->bound_a) { (void)hipEventDestroy(cm->bound_a); (void)hipEventDestroy(cm->bound_b); }
        if (cm->hscalar) (void)hipHostFree(cm->hscalar);
        if (cm->comm && rccl().CommDestroy) (void)rccl().CommDestroy(cm->comm);
        (void)hipStreamDestroy(cm->xstream);
        delete cm;
        return (int)COMET_OK;
    });
}
int comet_comm_rank(const comet_comm* cm) { return cm->rank; }
int comet_comm_world(const comet_comm* cm) { return cm->world; }

// host-value all-reduce on the exchange stream (context mutex held): op NCCL_MAX / NCCL_MIN / NCCL_SUM; returns when every rank's value is in
static double allreduce_host(comet_comm* cm, double v, int op) {
    double* d = cm->scalar.as<double>();
    if (!cm->hscalar) HIP_CHECK(hipHostMalloc((void**)&cm->hscalar, 16, hipHostMallocDefault));
    cm->hscalar[1] = v;
    HIP_CHECK(hipMemcpyAsync(d, cm->hscalar + 1, 8, hipMemcpyHostToDevice, cm->xstream));
    RCCL_CHECK(rccl().AllReduce(d, d + 1, 1, NCCL_FLOAT64, op, cm->comm, cm->xstream));
    HIP_CHECK(hipMemcpyAsync(cm->hscalar, d + 1, 8, hipMemcpyDeviceToHost, cm->xstream));
    HIP_CHECK(hipStreamSynchronize(cm->xstream));
    return cm->hscalar[0];
}
// all ranks: *inout = max over ranks (op 0) or sum over ranks (op 1) of a host double; blocks until complete (a barrier).
int comet_comm_allreduce_f64(comet_comm* cm, double* inout, int32_t op) {
    return guarded([&] {
        Ctx* c = cm->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        c->quiesce_all();                                                 // everything enqueued so far on the search streams is part of "before the barrier"
        *inout = allreduce_host(cm, *inout, op == 0 ? NCCL_MAX : NCCL_SUM);
        return (int)COMET_OK;
    });
}
int comet_comm_barrier(comet_comm* cm) { double x = 0; return comet_comm_allreduce_f64(cm, &x, 1); }

// Sharded search, enqueue only: this rank's shard is searched for the batch (results land in the slot's packed block).
int comet_index_search_sharded_async(comet_index* idx, comet_comm* cm, const float* queries_dev, int32_t B, const comet_search_params* p,
                                     uint32_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, int32_t k_cap, uint64_t* out_ticket) {
    return guarded([&] {
        if (!p || B <= 0 || k_cap <= 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "bad batch size / k_cap");
        if (idx->c != cm->c) COMET_FAIL(COMET_ERR_INVALID_ARG, "index and communicator live on different contexts");
        Ctx* c = idx->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        // A list-sharded index: every rank must deal the lists to the ranks the same way (the placement comes from host-side training state, which a rank that
        // LOADED its quantisers does not have — comet_index_set_list_owners). The first sharded search of an index on a communicator compares a fingerprint of
        // the placement over the ranks (two blocking all-reduces, once): ranks that disagree would own some lists twice and others not at all, silently.
        // EVERY rank of the communicator joins the two all-reduces, whatever its own state (round-5 advisor: a rank that failed its local check before them, or
        // skipped them because its own shard was unset, left its peers waiting in them for ever): a rank whose shard does not match the communicator contributes
        // a sentinel instead of a fingerprint, and all ranks fail together AFTER the collectives.
        if (cm->world > 1 && (idx->kind == COMET_KIND_IVF || idx->kind == COMET_KIND_IVFPQ) && idx->owners_checked_on != cm->uid) {
            const bool unset = idx->shard_world <= 1;
            const bool mismatch = !unset && (idx->shard_world != cm->world || idx->shard_rank != cm->rank);
            const double fp = mismatch ? -2.0 : unset ? -1.0 : (double)(idx->owners_fingerprint() & ((1ull << 52) - 1));
            const double hi = allreduce_host(cm, fp, NCCL_MAX), lo = allreduce_host(cm, fp, NCCL_MIN);
            if (mismatch) COMET_FAIL(COMET_ERR_INVALID_ARG, "index is shard %d of %d, the communicator is rank %d of %d", idx->shard_rank, idx->shard_world, cm->rank, cm->world);
            if (lo == -2.0) COMET_FAIL(COMET_ERR_INVALID_ARG, "a rank of this communicator holds a shard that does not match its rank / the communicator's size (comet_index_set_shard)");
            if (hi != lo) {
                if (lo == -1.0) COMET_FAIL(COMET_ERR_INVALID_ARG, "some ranks of this communicator hold a list shard of the index and others the whole index (comet_index_set_shard on every rank, or on none)");
                COMET_FAIL(COMET_ERR_INVALID_ARG, "the ranks of this communicator disagree on which rank owns which inverted list (placement fingerprints %.0f .. %.0f): "
                                                  "train every rank on the same vectors or hand every rank the same placement (comet_index_set_list_owners)", lo, hi);
            }
            idx->owners_checked_on = cm->uid;
        }
        // every other sharded search of an index on the context's second lane, like comet_index_search_dev_async: a rank's
        // short kernels (query preparation, post stage, coarse ranking) are the part of its step that does not shrink with the shard
        struct LaneBack { Ctx* c; ~LaneBack() { if (c->cur_lane != 0) { c->mark_dirty(); c->switch_lane(0); } } } lane_back{c};
        // the first asynchronous search of the context (same sequence as api.hip's CallGuard(c, lane)): everything lane 0 holds so far — the queries'
        // upload or generation — is what a search on lanes 1.. must start behind; from here on non-search calls keep the fence current
        if (!c->async_seen) { c->switch_lane(0); c->fence_lane0(); c->async_seen = true; }
        { const int m = std::min(c->lanes, idx->max_lanes()); c->switch_lane(m > 1 ? (idx->lane_toggle = (idx->lane_toggle + 1) % m) : 0); }
        c->scratch_reset();
        c->follow_lane0();          // lanes 1..: behind the non-search work lane 0 was last given (the queries' upload / generation)
        comet_comm::Slot* s = &cm->slots[cm->next_ticket % comet_comm::kSlots];   // round robin: a slot is reused four searches later
        if (s->active) s = nullptr;
        if (!s) COMET_FAIL(COMET_ERR_INVALID_ARG, "more than %d sharded searches in flight: wait for one first", comet_comm::kSlots);
        if (!s->searched) { HIP_CHECK(hipEventCreateWithFlags(&s->searched, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&s->merged, hipEventDisableTiming)); }
        const size_t words = (size_t)2 * B * k_cap + B;
        // the merge of the previous user of this slot must be done before its buffers are overwritten
        HIP_CHECK(hipEventSynchronize(s->merged));
        s->pack.reserve(words * 4, c->stream, 0); s->gathered.reserve(words * 4 * cm->world, c->stream, 0);
        if (const size_t wsb = merge_topk_workspace_bytes(cm->world, B, k_cap)) s->mergews.reserve(wsb, c->stream, 0);
        uint32_t* pids = s->pack.as<uint32_t>(); float* psc = reinterpret_cast<float*>(pids + (size_t)B * k_cap); int32_t* pcn = reinterpret_cast<int32_t*>(pids + (size_t)2 * B * k_cap);
        // list-sharded two-stage IVFPQ search: the ranks' stage-1 bounds are reduced to their minimum on the search stream (in place, B floats;
        // every rank runs the same sub-batching, so the calls pair up)
        struct Guard { comet_index* i; ~Guard() { i->bound_exchange = nullptr; i->bound_exchange_user = nullptr; } } guard{idx};
        if (cm->world > 1) {
            idx->bound_exchange_user = cm;
            // every collective of the communicator is issued on ONE stream (the exchange stream): the search stream hands over with an event
            // and waits for the reduced bounds with another
            idx->bound_exchange = [](void* user, uint32_t* tq, int n) {
                comet_comm* m = static_cast<comet_comm*>(user);
                if (!m->bound_a) { HIP_CHECK(hipEventCreateWithFlags(&m->bound_a, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&m->bound_b, hipEventDisableTiming)); }
                HIP_CHECK(hipEventRecord(m->bound_a, m->c->stream));
                HIP_CHECK(hipStreamWaitEvent(m->xstream, m->bound_a, 0));
                RCCL_CHECK(rccl().AllReduce(tq, tq, (size_t)n, NCCL_FLOAT32, NCCL_MIN, m->comm, m->xstream));
                HIP_CHECK(hipEventRecord(m->bound_b, m->xstream));
                HIP_CHECK(hipStreamWaitEvent(m->c->stream, m->bound_b, 0));
                m->exch_done++;
            };
        }
        // From here on the peers are (or will be) inside this batch's collectives: a failure of THIS rank's search (out of memory, a shape its kernels refuse)
        // must not leave them waiting. The rank issues the bound exchanges it still owes (+inf: "no bound from me"), hands its peers a block whose counts are
        // -code (the merge passes a negative count through to every rank), and reports the error from _wait.
        int per = 0; const int owed = cm->world > 1 ? idx->sharded_exchanges(B, *p, k_cap, &per) : 0;
        cm->exch_done = 0; s->failed = 0; s->why.clear(); s->search_ticket = 0;
        auto failed_search = [&](int code, const std::string& msg) {
            s->failed = code ? code : (int)COMET_ERR_HIP; s->why = msg;
            if (cm->exch_done < owed && idx->bound_exchange) {
                cm->idle_tq.reserve((size_t)std::max(B, 1) * 4, c->stream, 0);
                HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)cm->idle_tq.p, 0x7F800000, (size_t)B, c->stream));
                for (int e = cm->exch_done; e < owed; e++) { const int b0 = e * per; idx->bound_exchange(cm, cm->idle_tq.as<uint32_t>() + b0, std::min(per, B - b0)); }
            }
            HIP_CHECK(hipMemsetAsync(pids, 0, (size_t)2 * B * k_cap * 4, c->stream));
            HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)pcn, (int)(-s->failed), (size_t)B, c->stream));
        };
        // test hook (read once): COMET_TEST_FAIL_SEARCH="rank:ticket" makes that rank's search with that ticket throw before it enqueues anything
        static const std::pair<int, long> inject = [] { const char* e = getenv("COMET_TEST_FAIL_SEARCH"); int r = -1; long t = -1; if (e) sscanf(e, "%d:%ld", &r, &t); return std::make_pair(r, t); }();
        try {
            if (inject.first == cm->rank && inject.second == (long)cm->next_ticket) COMET_FAIL(COMET_ERR_UNSUPPORTED, "injected failure (COMET_TEST_FAIL_SEARCH)");
            s->search_ticket = idx->search_begin(queries_dev, B, *p, pids, psc, pcn, k_cap);
        }
        catch (const StatusError& e) { failed_search(e.code, last_error()); }
        catch (const HipError& h) { char m[384]; snprintf(m, sizeof(m), "HIP error %d (%s) in %s at %s:%d", (int)h.e, hipGetErrorString(h.e), h.what, h.file, h.line); failed_search(COMET_ERR_HIP, m); }
        catch (const std::bad_alloc&) { failed_search(COMET_ERR_HIP, "host allocation failed"); }
        s->idx = idx; s->B = B; s->k_cap = k_cap; s->k = p->k; s->out_ids = out_ids_dev; s->out_scores = out_scores_dev; s->out_counts = out_counts_dev;
        HIP_CHECK(hipEventRecord(s->searched, c->stream));          // the exchange of this batch depends on THIS search only, not on later ones
        s->ticket = cm->next_ticket++; s->active = true;
        if (out_ticket) *out_ticket = s->ticket;
        return (int)COMET_OK;
    });
}

// Finish a sharded search: make the local results final (the one host-side decision the Flat fast path defers), then enqueue
// all-gather + merge on the communicator's stream. block != 0: return when the merged results are in the output buffers;
// block == 0: return at once (the outputs are final after a later blocking wait, comet_comm_barrier or comet_comm_sync).
int comet_index_search_sharded_wait(comet_index* idx, comet_comm* cm, uint64_t ticket, int32_t block) {
    return guarded([&] {
        Ctx* c = idx->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        comet_comm::Slot* s = nullptr;
        for (auto& sl : cm->slots) if (sl.active && sl.ticket == ticket) { s = &sl; break; }
        if (!s) { if (block) HIP_CHECK(hipStreamSynchronize(cm->xstream)); return (int)COMET_OK; }
        // waits for THIS search only (its event), not for later ones; a rare strict re-run of overflowed queries lands behind the
        // later searches on the stream, so the dependency is re-recorded in that case
        if (!s->failed && idx->search_finish(s->search_ticket)) HIP_CHECK(hipEventRecord(s->searched, c->stream));
        HIP_CHECK(hipStreamWaitEvent(cm->xstream, s->searched, 0));
        const size_t words = (size_t)2 * s->B * s->k_cap + s->B;
        {
            StreamSwap sw(c, cm->xstream);
            { ProfScope ps(c, "shard_allgather"); RCCL_CHECK(rccl().AllGather(s->pack.p, s->gathered.p, words, NCCL_INT32, cm->comm, cm->xstream)); }
            const uint32_t* g = s->gathered.as<uint32_t>();
            launch_merge_topk(c, g, reinterpret_cast<const float*>(g + (size_t)s->B * s->k_cap), reinterpret_cast<const int32_t*>(g + (size_t)2 * s->B * s->k_cap),
                              cm->world, s->B, s->k_cap, s->k, s->out_ids, s->out_scores, s->out_counts, (int64_t)words, (int64_t)words,
                              merge_topk_workspace_bytes(cm->world, s->B, s->k_cap) ? s->mergews.p : nullptr);
        }
        HIP_CHECK(hipEventRecord(s->merged, cm->xstream));
        s->active = false;
        if (block) HIP_CHECK(hipEventSynchronize(s->merged));
        if (s->failed) { const int code = s->failed; s->failed = 0; COMET_FAIL(code, "sharded search failed on rank %d (every rank's counts for the batch are %d): %s", cm->rank, -code, s->why.c_str()); }
        return (int)COMET_OK;
    });
}
int comet_comm_sync(comet_comm* cm) {
    return guarded([&] { Ctx* c = cm->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->quiesce_all(); HIP_CHECK(hipStreamSynchronize(cm->xstream)); c->collect_profile(); return (int)COMET_OK; });
}

}  // extern "C"
