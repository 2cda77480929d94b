// index.hpp — host-side index objects behind the C ABI. They mirror the reference's containers
// (FlatIndex flat_index.go:82-100, IVFIndex ivf_index.go:98-118, PQIndex pq_index.go:99-120,
// IVFPQIndex ivfpq_index.go:40-100) but keep the data resident in HBM in layouts chosen for the
// kernels (padded row-major fp32 matrices, list-major inverted lists, interleaved PQ code blocks).
#pragma once
#include <algorithm>
#include <memory>
#include <unordered_map>
#include <unordered_set>

#include "kernels.hpp"
#include <cstdio>
#include "io.hpp"

struct comet_ctx : comet::Ctx {};

struct comet_index {
    comet::Ctx* c = nullptr;
    int kind = 0, dim = 0, ld = 0, metric = 0;
    bool trained = false;
    // Integrity guards around the host-side containers of the base object (a soak run of 7 000+ index lifetimes once ended in free(-1) inside this
    // destructor: something had written 0xFF bytes over a vector's pointers). Checked on entry to and exit from every call on the index's context
    // (CallGuard / comet_index_destroy): a mismatch aborts with the name of the call that first saw it.
    static constexpr uint64_t kGuard = 0xC0FEE1D5A5A5F00Dull;
    uint64_t guard0 = kGuard;
    bool guards_ok() const {
        auto sane = [](const std::vector<int32_t>& v) { return v.data() == nullptr ? (v.size() == 0 && v.capacity() == 0) : (v.size() <= v.capacity() && v.capacity() < ((size_t)1 << 40)); };
        return guard0 == kGuard && guard1 == kGuard && guard2 == kGuard && sane(train_counts) && sane(list_owner) && deleted.bucket_count() < ((size_t)1 << 40);
    }
    void guards_dump(const char* where) const {
        const uint64_t* w = reinterpret_cast<const uint64_t*>(this);
        std::fprintf(stderr, "comet: index object %p (kind %d) corrupted, first seen at %s; guards %llx %llx %llx; words:", (const void*)this, kind, where,
                     (unsigned long long)guard0, (unsigned long long)guard1, (unsigned long long)guard2);
        for (int i = 0; i < 48; i++) std::fprintf(stderr, " %llx", (unsigned long long)w[i]);
        std::fprintf(stderr, "\n");
    }
    // soft deletes (deletedNodes roaring bitmap in the reference, flat_index.go:88)
    std::unordered_set<uint32_t> deleted;
    comet::DevBuf deleted_dev; int n_deleted_dev = 0; bool deleted_dirty = false;

    // multi-GPU list sharding (IVF / IVFPQ): this rank keeps only the members of lists l with l % shard_world == shard_rank;
    // centroids / codebooks stay replicated, every rank ranks ALL centroids and probes the same lists, and the lists it does
    // not own are simply empty here — table build, scan and selection all shrink with the rank count.
    int shard_rank = 0, shard_world = 1;
    // Which rank owns which list. Lists are dealt by LENGTH, not by index: the members per list seen in training (`train_counts`, the same on every
    // rank: training is deterministic and replicated) estimate the lists' final lengths, and the lists go longest first to the rank with the
    // least estimated rows (LPT; ties: fewer lists, then the lower rank) — a corpus with a few giant lists (the bench's clustered corpus: longest
    // list 23 x the mean) no longer piles them on the ranks their indices happen to name. Empty until trained: l % world then.
    std::vector<int32_t> train_counts, list_owner;
    uint64_t guard1 = kGuard;
    int owner_of(int32_t l) const { return list_owner.empty() ? l % shard_world : list_owner[(size_t)l]; }
    // what the ranks of a communicator compare before the first sharded search (comm.hip): FNV-1a over (world, the placement; "l % world" when there is none)
    uint64_t owners_checked_on = 0;      // comet_comm::uid of the communicator the placement was last compared on (0: none)
    uint64_t owners_fingerprint() const {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint32_t v) { for (int b = 0; b < 4; b++) { h ^= (v >> (8 * b)) & 0xFF; h *= 1099511628211ull; } };
        mix((uint32_t)shard_world); mix((uint32_t)list_owner.size());
        for (int32_t o : list_owner) mix((uint32_t)o);
        return h;
    }
    // the placement is not part of the reference's on-disk layouts: an index that loads its quantisers (read_from) starts without one (l % world on every
    // rank that loaded) until the host hands it the placement it saved (comet_index_get_list_owners / _set_list_owners)
    void forget_placement() { train_counts.clear(); list_owner.clear(); owners_checked_on = 0; }
    void assign_list_owners() {
        list_owner.clear();
        if (shard_world <= 1 || train_counts.empty()) return;
        const int nl = (int)train_counts.size();
        std::vector<int32_t> order(nl);
        for (int l = 0; l < nl; l++) order[l] = l;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return train_counts[a] > train_counts[b]; });
        std::vector<int64_t> load(shard_world, 0); std::vector<int32_t> cnt(shard_world, 0);
        list_owner.assign(nl, 0);
        for (int32_t l : order) {
            int best = 0;
            for (int r = 1; r < shard_world; r++) if (load[r] < load[best] || (load[r] == load[best] && cnt[r] < cnt[best])) best = r;
            list_owner[l] = best; load[best] += train_counts[l]; cnt[best]++;
        }
    }
    // "shard_est_rows" (training-set members of this rank's lists), "shard_est_rows_max" / "_mean" over the ranks, "shard_lists" (lists owned)
    bool shard_stat(const char* name, double* out) const {
        if (std::strncmp(name, "shard_", 6) != 0) return false;
        std::vector<double> load(std::max(1, shard_world), 0.0); double lists = 0;
        for (size_t l = 0; l < train_counts.size(); l++) { const int o = owner_of((int32_t)l); load[o] += train_counts[l]; if (o == shard_rank) lists += 1; }
        double mx = 0, sum = 0; for (double v : load) { mx = std::max(mx, v); sum += v; }
        if (!std::strcmp(name, "shard_est_rows")) { *out = load[shard_rank]; return true; }
        if (!std::strcmp(name, "shard_est_rows_max")) { *out = mx; return true; }
        if (!std::strcmp(name, "shard_est_rows_mean")) { *out = sum / load.size(); return true; }
        if (!std::strcmp(name, "shard_lists")) { *out = lists; return true; }
        return false;
    }
    void note_training_assignment(const std::vector<int32_t>& assign, int nl) {
        train_counts.assign(nl, 0);
        for (int32_t a : assign) if (a >= 0 && a < nl) train_counts[a]++;
        assign_list_owners();
    }
    // set by comet_index_search_sharded_async for the duration of the call: all-reduce(min) of n floats on the context's stream across the
    // ranks of the communicator (the stage-1 bound exchange of the sharded two-stage IVFPQ search); null outside a sharded search
    void (*bound_exchange)(void* user, uint32_t* tq, int n) = nullptr; void* bound_exchange_user = nullptr;
    // how many bound exchanges a sharded search of B queries issues on EVERY rank (whatever the rank's own lists look like), *per = queries per exchange:
    // a rank whose search fails half way still owes its peers the rest of them (comm.hip)
    virtual int sharded_exchanges(int /*B*/, const comet_search_params& /*p*/, int /*k_cap*/, int* per) const { if (per) *per = 0; return 0; }

    virtual ~comet_index() {
        for (auto& r : done_ring) if (r.ev) (void)hipEventDestroy(r.ev);
        if (c) { auto& L = c->live_indexes; for (size_t i = 0; i < L.size(); i++) if (L[i] == this) { L.erase(L.begin() + i); break; } }
    }
    virtual int64_t size() const = 0;
    virtual int default_nprobes() const { return 0; }
    virtual int n_lists() const { return 0; }
    virtual void train_dev(const float* /*vecs_dev*/, int64_t /*n*/) {}   // VectorIndex.Train; no-op for Flat (flat_index.go:150)
    // add n rows already on device (dense n x dim); returns rows added; sets *zero_at = index of first
    // zero-norm vector (or -1). normalized_dev (nullable): receives the preprocessed rows (dense).
    virtual int64_t add_dev(const uint32_t* ids_dev, const uint32_t* ids_host, const float* vecs_dev, int64_t n,
                            int64_t* zero_at, float* normalized_dev) = 0;
    virtual bool contains_id(uint32_t id) const = 0;
    // row of the stored (preprocessed) vector of `id` in rows_dev(), the first one in the reference's lookup order
    // (lookupNodeVectors flat_index_search.go:171-196, ivf_index_search.go:176-206, hnsw_index_search.go:212-226); -1: no such id.
    // Index kinds that keep no vectors (PQ / IVFPQ store codes only) return nullptr from rows_dev().
    virtual int64_t row_of_id(uint32_t /*id*/) { return -1; }
    virtual const float* rows_dev() const { return nullptr; }
    virtual void flush() = 0;
    virtual void search_dev(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids,
                            float* out_scores, int32_t* out_counts, int k_cap) = 0;
    // asynchronous form: begin enqueues the search and returns a ticket; finish(ticket) makes the results final
    // (index kinds without deferred work run everything in begin)
    virtual uint64_t search_begin(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                                  int32_t* out_counts, int k_cap) { search_dev(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap); return record_done(); }
    // returns true if it had to enqueue further device work (the Flat fast path's rare strict re-run)
    virtual bool search_finish(uint64_t ticket) { wait_done(ticket); return false; }
    // how many of the context's execution lanes this kind's asynchronous searches rotate through (> 1: a search touches nothing persistent on
    // the device except read-only index data, its own ring slot and per-lane buffers). Two where one kernel of the step fills the GPU
    // (Flat, IVF: a third search in flight only adds contention), four where none does (PQ / IVFPQ, HNSW).
    virtual int max_lanes() const { return 1; }
    int lane_toggle = 0;
    // completion tickets of the index kinds without deferred work: an event behind the search on the stream it was enqueued on
    struct DoneEv { uint64_t ticket = 0; hipEvent_t ev = nullptr; bool active = false; };
    DoneEv done_ring[8]; uint64_t done_next = 1;
    uint64_t guard2 = kGuard;
    uint64_t record_done() {
        DoneEv* slot = nullptr;
        for (auto& r : done_ring) if (!r.active) { slot = &r; break; }
        if (!slot) { slot = &done_ring[0]; for (auto& r : done_ring) if (r.ticket < slot->ticket) slot = &r; HIP_CHECK(hipEventSynchronize(slot->ev)); }
        if (!slot->ev) HIP_CHECK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(slot->ev, c->stream));
        slot->ticket = done_next++; slot->active = true;
        return slot->ticket;
    }
    void wait_done(uint64_t ticket) {
        for (auto& r : done_ring) if (r.active && r.ticket == ticket) { HIP_CHECK(hipEventSynchronize(r.ev)); r.active = false; return; }
    }
    virtual int64_t list_size(int /*list*/) const { return size(); }
    virtual void list_read(int /*list*/, uint32_t* /*ids*/, uint8_t* /*codes*/, float* /*vecs*/) const {}
    virtual void export_all(uint32_t* /*ids*/, int32_t* /*lists*/, uint8_t* /*codes*/) const { COMET_FAIL(COMET_ERR_UNSUPPORTED, "export not supported for this index kind"); }
    // io.WriterTo / io.ReaderFrom in the reference's byte layout (io.hpp); write_to flushes first like the reference
    virtual void write_to(comet::Sink& /*s*/) { COMET_FAIL(COMET_ERR_UNSUPPORTED, "serialisation not supported for this index kind"); }
    virtual void read_from(comet::Source& /*s*/) { COMET_FAIL(COMET_ERR_UNSUPPORTED, "deserialisation not supported for this index kind"); }
    virtual bool get_stat(const char* /*name*/, double* /*out*/) const { return false; }
    virtual void get_centroids(float*) const { COMET_FAIL(COMET_ERR_UNSUPPORTED, "index has no centroids"); }
    virtual void get_codebooks(float*) const { COMET_FAIL(COMET_ERR_UNSUPPORTED, "index has no codebooks"); }

    // sorted device copy of the soft-delete set (rebuilt lazily)
    const uint32_t* deleted_sorted_dev() {
        if (deleted_dirty) {
            c->quiesce_all();      // a search in flight on either lane may still be reading the list that is about to be replaced
            std::vector<uint32_t> v(deleted.begin(), deleted.end());
            std::sort(v.begin(), v.end());
            deleted_dev.reserve(std::max<size_t>(4, v.size() * 4), c->stream, 0);
            c->h2d(deleted_dev.p, v.data(), v.size() * 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));   // v is a temporary
            n_deleted_dev = (int)v.size(); deleted_dirty = false;
        }
        return deleted_dev.as<uint32_t>();
    }
    // sorted filter ids in scratch (nullptr / 0 when no filter)
    const uint32_t* filter_sorted_scratch(const comet_search_params& p, int* n_out) {
        *n_out = 0;
        if (!p.filter_ids || p.n_filter <= 0) return nullptr;
        std::vector<uint32_t> v(p.filter_ids, p.filter_ids + p.n_filter);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        uint32_t* d = c->salloc<uint32_t>(v.size());
        c->h2d(d, v.data(), v.size() * 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        *n_out = (int)v.size();
        return d;
    }
};

namespace comet {
comet_index* make_flat(Ctx* c, int dim, int metric);
comet_index* make_ivf(Ctx* c, int dim, int metric, int nlist);
comet_index* make_pq(Ctx* c, int dim, int metric, int M, int nbits);
comet_index* make_ivfpq(Ctx* c, int dim, int metric, int nlist, int M, int nbits);

// preprocess B queries (dense B x dim) into padded B x ld; zflag[q] = 1 for zero-norm cosine queries
// may_alias: the caller never offsets zflag and accepts Qp == queries_dev (+ zflag == nullptr) when no preprocessing is needed
void prepare_queries(Ctx* c, int metric, const float* queries_dev, int B, int dim, int ld, float** Qp, int32_t** zflag, bool may_alias = false);
// out_ids[q][i] = ids[pos[q][i]] ; counts[q] = -COMET_ERR_ZERO_VECTOR where zflag[q]
void launch_finalize(Ctx* c, const uint32_t* ids_table, const uint32_t* pos, int B, int k_cap, const int32_t* zflag,
                     uint32_t* out_ids, int32_t* counts);
}  // namespace comet
