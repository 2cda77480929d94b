// index_quant.hip — k-means driver and the IVF / PQ / IVFPQ indexes (reference: clustering.go,
// ivf_index.go + ivf_index_search.go, pq_index.go + pq_index_search.go, ivfpq_index.go +
// ivfpq_index_search.go).
//
// HBM layout
//   * vectors / centroids: padded row-major fp32 (kernels.hpp ROW_PAD);
//   * inverted lists: elements are kept in arrival (Add) order and "compiled" lazily into list-major
//     SLOTS — a stable counting sort by list, so each list keeps the reference's append order. For the
//     raw-vector IVF a slot indexes the arrival-order row (row indirection is free in the gather kernel);
//     for PQ / IVFPQ each list is padded to 64-code blocks and its codes are stored word-interleaved per
//     block so a wave reads 256 contiguous bytes per load (see adc_scan_kernel).
#include "index.hpp"

namespace comet {

// ------------------------------------------------------------------------------------------------
// exact nearest-centroid assignment and k-means (clustering.go:119-272)
// ------------------------------------------------------------------------------------------------
// out_idx[v] = FindNearestCentroidIndex(V[v], C) for n padded rows (device). Lowest index wins ties.
static void assign_nearest(Ctx* c, int metric, const float* V, int64_t n, int ld, const float* C, int k, int32_t* out_idx) {
    if (n <= 0) return;
    ScratchMark sm(c);
    const int64_t vchunk = std::min<int64_t>(n, 262144);
    const int64_t ldD = round_up(vchunk, 16);
    const size_t budget = (size_t)1 << 30;
    int kb = (int)std::min<int64_t>(k, std::max<int64_t>(16, (int64_t)(budget / ((size_t)ldD * 4)) / 16 * 16));
    float* D = c->salloc<float>((size_t)kb * ldD);
    float* best = c->salloc<float>(vchunk);
    for (int64_t v0 = 0; v0 < n; v0 += vchunk) {
        const int64_t nv = std::min(vchunk, n - v0);
        for (int c0 = 0; c0 < k; c0 += kb) {
            const int kn = std::min(kb, k - c0);
            ScratchMark sm2(c);
            // Calculate(vector, centroid): (v-c)^2 == (c-v)^2 and v*c == c*v bitwise, so the centroid block
            // plays the "query" role of the exact distance kernel and the vectors the "row" role.
            launch_dist_exact(c, metric, V + (size_t)v0 * ld, nv, ld, C + (size_t)c0 * ld, kn, D, ldD, nullptr);
            launch_argmin_update(c, D, ldD, kn, c0, nv, best, out_idx + v0, c0 == 0);
        }
    }
}

// kmeansInternal (clustering.go:119-243). V: n x ld padded rows on device. centroids: >= min(k,n) x ld,
// assign: n int32 (device). Returns the effective k (0 for the reference's (nil, nil) cases).
static int kmeans_device(Ctx* c, int metric, const float* V, int64_t n, int ld, int k, int max_iter, float* centroids, int32_t* assign) {
    if (n <= 0 || k <= 0) return 0;
    if (k > n) k = (int)n;
    if (max_iter <= 0) max_iter = 20;   // DefaultMaxIter clustering.go:14
    ScratchMark sm(c);
    {   // uniform-stride initialisation clustering.go:147-162
        int64_t step = n / k; if (step == 0) step = 1;
        std::vector<int32_t> idx(k);
        for (int i = 0; i < k; i++) { int64_t vi = (int64_t)i * step; if (vi >= n) vi = n - 1; idx[i] = (int32_t)vi; }
        int32_t* didx = c->salloc<int32_t>(k);
        c->h2d(didx, idx.data(), k * sizeof(int32_t));
        launch_gather_rows(c, V, ld, didx, k, centroids);
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    launch_fill_i32(c, assign, n, -1);   // UnassignedCluster
    int32_t* new_idx = c->salloc<int32_t>(n);
    int32_t* changed = c->salloc<int32_t>(1);
    for (int it = 0; it < max_iter; it++) {
        assign_nearest(c, metric, V, n, ld, centroids, k, new_idx);
        c->zero(changed, sizeof(int32_t));
        launch_apply_assign(c, new_idx, assign, n, changed);
        int32_t h = 0;
        c->d2h(&h, changed, sizeof(int32_t));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (!h) break;                    // converged (clustering.go:203)
        ScratchMark sm2(c);
        launch_kmeans_update(c, V, n, ld, assign, k, centroids);
    }
    return k;
}

// ------------------------------------------------------------------------------------------------
// list layout shared by IVF / PQ / IVFPQ
// ------------------------------------------------------------------------------------------------
struct ListLayout {
    int nlist = 1, align = 1;
    std::vector<uint32_t> ids;       // arrival order
    std::vector<int32_t> list_of;    // arrival order
    int64_t n = 0;
    bool dirty = true;
    // compiled (host)
    std::vector<int32_t> len_h; std::vector<int64_t> base_h; std::vector<uint32_t> row_of_slot_h;
    int64_t nslots = 0; int max_len = 0;
    uint64_t version = 0;                  // bumped by every compile (derived device state, e.g. the IVF fp16 shadow, follows it)
    int64_t total_tiles256 = 0;            // sum over lists of ceil(len / 256)
    std::map<int, std::pair<int64_t, int64_t>> bound_cache;   // np -> (max candidates, max 64-row units) of one query
    // compiled (device)
    DevBuf row_of_slot, ids_slot, list_base, list_len;
    std::unordered_map<uint32_t, int> id_count;

    void append(const uint32_t* new_ids, const int32_t* lists, int64_t m) {
        ids.insert(ids.end(), new_ids, new_ids + m);
        if (lists) list_of.insert(list_of.end(), lists, lists + m); else list_of.insert(list_of.end(), m, 0);
        for (int64_t i = 0; i < m; i++) id_count[new_ids[i]]++;
        n += m; dirty = true;
    }
    // stable counting sort by list -> slots
    void compile(Ctx* c) {
        if (!dirty) return;
        len_h.assign(nlist, 0);
        for (int64_t i = 0; i < n; i++) len_h[list_of[i]]++;
        base_h.assign(nlist, 0);
        int64_t run = 0; max_len = 0;
        total_tiles256 = 0;
        for (int l = 0; l < nlist; l++) { base_h[l] = run; run += round_up(len_h[l], align); max_len = std::max(max_len, len_h[l]); total_tiles256 += ceil_div(len_h[l], 256); }
        nslots = run; version++; bound_cache.clear();
        row_of_slot_h.assign(std::max<int64_t>(nslots, 1), 0xFFFFFFFFu);
        std::vector<uint32_t> ids_slot_h(std::max<int64_t>(nslots, 1), 0);
        std::vector<int64_t> cur(base_h);
        for (int64_t i = 0; i < n; i++) { int64_t s = cur[list_of[i]]++; row_of_slot_h[s] = (uint32_t)i; ids_slot_h[s] = ids[i]; }
        row_of_slot.reserve(row_of_slot_h.size() * 4, c->stream, 0);
        ids_slot.reserve(ids_slot_h.size() * 4, c->stream, 0);
        list_base.reserve((size_t)nlist * 8, c->stream, 0);
        list_len.reserve((size_t)nlist * 4, c->stream, 0);
        c->h2d(row_of_slot.p, row_of_slot_h.data(), row_of_slot_h.size() * 4);
        c->h2d(ids_slot.p, ids_slot_h.data(), ids_slot_h.size() * 4);
        c->h2d(list_base.p, base_h.data(), (size_t)nlist * 8);
        c->h2d(list_len.p, len_h.data(), (size_t)nlist * 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        dirty = false;
    }
    // upper bound on the candidates of one query probing `np` lists: the np longest lists
    int64_t max_candidates(int np) { return bounds(np).first; }
    // upper bound on the 64-row units of one query probing `np` lists
    int64_t max_units(int np) { return bounds(np).second; }
    const std::pair<int64_t, int64_t>& bounds(int np) {
        auto it = bound_cache.find(np);
        if (it != bound_cache.end()) return it->second;
        std::vector<int32_t> l(len_h);
        const int m = std::min(np, nlist);
        if (m < nlist) std::partial_sort(l.begin(), l.begin() + m, l.end(), std::greater<int32_t>());
        int64_t s = 0, u = 0; for (int i = 0; i < m; i++) { s += l[i]; u += ceil_div(l[i], 64); }
        return bound_cache[np] = std::make_pair(s, u);
    }
    // rows (arrival indices) surviving a flush, in arrival order
    std::vector<int64_t> survivors(const std::unordered_set<uint32_t>& deleted) const {
        std::vector<int64_t> keep;
        for (int64_t i = 0; i < n; i++) if (!deleted.count(ids[i])) keep.push_back(i);
        return keep;
    }
    void keep_rows(const std::vector<int64_t>& keep) {
        std::vector<uint32_t> nid(keep.size()); std::vector<int32_t> nl(keep.size());
        for (size_t i = 0; i < keep.size(); i++) { nid[i] = ids[keep[i]]; nl[i] = list_of[keep[i]]; }
        ids.swap(nid); list_of.swap(nl); n = (int64_t)keep.size();
        id_count.clear(); for (auto id : ids) id_count[id]++;
        dirty = true;
    }
};

// compact the surviving rows of a row-major device buffer (row_bytes each) into a fresh buffer
static void compact_rows(Ctx* c, DevBuf& buf, size_t row_bytes, const std::vector<int64_t>& keep) {
    DevBuf nb;
    nb.reserve(std::max<size_t>(4, keep.size() * row_bytes), c->stream, 0);
    size_t i = 0;
    while (i < keep.size()) {
        size_t j = i;
        while (j + 1 < keep.size() && keep[j + 1] == keep[j] + 1) j++;
        c->d2d((char*)nb.p + i * row_bytes, (char*)buf.p + (size_t)keep[i] * row_bytes, (j - i + 1) * row_bytes);
        i = j + 1;
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    std::swap(buf.p, nb.p); std::swap(buf.cap, nb.cap);
}

// coarse step shared by IVF and IVFPQ (ivf_index_search.go:246-261): rank all centroids, keep nprobes.
// probe_list[q][0..np) = centroid indices sorted by (distance, index).
// list_len != nullptr: seg_off / cnts / uoff (see launch_probe_segments, launch_ivf_probe_units) are wanted as well; returns true if they were
// written here (the fast ranking's pick kernel does it on its way out), false if the caller has to launch the bookkeeping kernel itself
static bool coarse_probe(Ctx* c, int metric, const float* centroids, int nlist, int ld, int dim, const float* Qp, int B, int np, uint32_t* probe_list,
                         const int32_t* list_len = nullptr, int32_t* seg_off = nullptr, int32_t* cnts = nullptr, int32_t* uoff = nullptr) {
    if (launch_coarse_probe_fast(c, metric, centroids, nlist, ld, dim, Qp, B, np, probe_list, list_len, seg_off, cnts, uoff, ivf_fast_unit_rows())) return list_len != nullptr;     // approximate ranking + exact re-scoring of the few that matter
    const int64_t ldDc = round_up(nlist, 16);
    float* Dc = c->salloc<float>((size_t)B * ldDc);
    launch_dist_exact(c, metric, centroids, nlist, ld, Qp, B, Dc, ldDc, nullptr);
    float* psc = c->salloc<float>((size_t)B * np);
    int32_t* pcnt = c->salloc<int32_t>(B);
    launch_select_topk(c, Dc, ldDc, B, nlist, nullptr, 0.0f, np, probe_list, psc, pcnt, np);
    launch_probe_complete(c, probe_list, np, pcnt, B);
    return false;
}
static int sanitize_nprobes(int nprobes, int nlist) { return (nprobes <= 0 || nprobes > nlist) ? nlist : nprobes; }  // ivf_index_search.go:233-236

// raw training vectors (dense n x dim on device) -> padded rows without preprocessing
// (Train uses the vectors as given, even for cosine: ivf_index.go:217-224)
static float* pad_raw(Ctx* c, const float* vecs_dev, int64_t n, int dim, int ld) {
    float* V = c->salloc<float>((size_t)std::max<int64_t>(n, 1) * ld);
    launch_ingest_rows(c, COMET_L2SQ, vecs_dev, n, dim, V, ld, nullptr);
    return V;
}

// list sharding: positions (within a batch) of the rows whose list this rank owns
static std::vector<int32_t> owned_rows(const std::vector<int32_t>& lists, const comet_index* ix) {
    std::vector<int32_t> keep;
    for (size_t i = 0; i < lists.size(); i++) if (ix->owner_of(lists[i]) == ix->shard_rank) keep.push_back((int32_t)i);
    return keep;
}
// members per list of the training set (host copy of the device assignment): the length estimate the list -> rank assignment is made from
static void note_training(comet_index* ix, Ctx* c, const int32_t* assign_dev, int64_t n, int nlist) {
    std::vector<int32_t> h((size_t)n);
    c->d2h(h.data(), assign_dev, (size_t)n * sizeof(int32_t));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    ix->note_training_assignment(h, nlist);
}

// ------------------------------------------------------------------------------------------------
// IVFIndex
// ------------------------------------------------------------------------------------------------
struct IVFIndex : comet_index {
    int nlist = 0;
    int n_lists() const override { return nlist; }
    DevBuf centroids;   // nlist x ld
    DevBuf V;           // arrival-order rows, n x ld
    ListLayout lay;
    // fp16 shadow for the MFMA fast path (kernels_ivf.hip): slot-ordered rows, squared norms per slot, magnitude statistics;
    // rebuilt lazily when the slot layout was recompiled (shadow_version != lay.version)
    int ldh = 0;
    DevBuf Vh, rn_slot, stats_dev, scan_counts[Ctx::kMaxLanes];       // scan_counts: written by every search's item builder — one per execution lane
    int last_lane = 0;
    uint64_t shadow_version = 0;
    float xmax_abs = 0.0f, xmax_norm2 = 0.0f;
    // int8 shadow (one scale per 64-slot unit; kernels_ivf.hip ivf_shadow_i8_kernel): half the bytes of the HBM-bound scan. Policy as in
    // FlatIndex: a search whose int8 slices overflowed or proposed more than kI8MaxCand candidates per query sends the index back to the
    // fp16 shadow for a while (doubling back-off); COMET_IVF_I8 = 0 never / 1 always.
    int ld8 = 0;
    DevBuf V8, su;
    float xmax_err2 = 0.0f;
    static constexpr int64_t kI8MaxCand = 1024;
    int i8_policy = [] { const char* e = getenv("COMET_IVF_I8"); return e ? atoi(e) : -1; }();
    int64_t n_searches = 0, i8_resume_at = 0, st_i8_slices = 0, st_i8_backoffs = 0; int i8_strikes = 0;
    bool i8_usable() const {
        if (i8_policy == 0 || !prep_queries_i8_ok(dim) || !std::isfinite(xmax_err2) || !V8.p) return false;
        return i8_policy == 1 || n_searches >= i8_resume_at;
    }
    int64_t st_candidates = 0, st_overflows = 0, st_expansions = 0, st_fast_queries = 0, st_strict_queries = 0;
    // deferred verification of fast-path searches, as FlatIndex does it: the post stage leaves per-query overflow flags (candidate
    // list beyond its LDS capacity: adversarial clustering / mass ties) which travel to pinned memory on a second stream;
    // search_finish() waits for the search's event and re-runs the flagged queries on the exact kernels
    struct Pending {
        bool active = false; uint64_t ticket = 0; hipEvent_t ev = nullptr, ev_post = nullptr;
        int B = 0, k_cap = 0, nfast_slices = 0, slice = 256; uint64_t i8_mask = 0;    // bit sl: slice sl was screened on the int8 shadow
        const float* queries = nullptr; comet_search_params p{}; std::vector<uint32_t> flt;
        uint32_t* out_ids = nullptr; float* out_scores = nullptr; int32_t* out_counts = nullptr;
        int32_t* flags = nullptr;   // pinned: per fast slice [256 overflow flags | 4 stats]
        int32_t* dflags = nullptr;  // the same slices in HBM
    };
    hipStream_t copy_stream = nullptr;
    static constexpr int kRing = 8, kSliceInts = 260, kMaxSlices = 64, kFastBatch = 256;
    Pending ring[kRing];
    uint64_t next_ticket = 1;
    ~IVFIndex() override {
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);      // nothing of this index is in flight on its private stream when its events and pinned slots go
        for (auto& r : ring) { if (r.ev) (void)hipEventDestroy(r.ev); if (r.ev_post) (void)hipEventDestroy(r.ev_post); if (r.flags) (void)hipHostFree(r.flags); if (r.dflags) (void)hipFree(r.dflags); }
        // (the stream belongs to the context since round 5: comet_ctx_create)
    }

    int64_t size() const override { return lay.n; }
    int max_lanes() const override { return 2; }      // searches write scratch, their ring slot / per-lane counters and (opt-in) atomics only
    int default_nprobes() const override { return (int)std::sqrt((double)nlist); }   // ivf_index.go:406-413
    bool contains_id(uint32_t id) const override { return lay.id_count.count(id) != 0; }
    void export_all(uint32_t* oids, int32_t* olists, uint8_t*) const override {
        if (oids) std::copy(lay.ids.begin(), lay.ids.end(), oids);
        if (olists) std::copy(lay.list_of.begin(), lay.list_of.end(), olists);
    }
    // first match in the reference's order: list by list, append order inside a list (ivf_index_search.go:183-198)
    std::unordered_map<uint32_t, int64_t> first_row; uint64_t first_row_version = 0;     // id -> row of its first slot, rebuilt when the layout was recompiled
    int64_t row_of_id(uint32_t id) override {
        lay.compile(c);
        if (first_row_version != lay.version) {
            first_row.clear(); first_row.reserve((size_t)lay.n);
            for (int64_t s = 0; s < lay.nslots; s++) { const uint32_t r = lay.row_of_slot_h[s]; if (r != 0xFFFFFFFFu) first_row.emplace(lay.ids[r], (int64_t)r); }   // emplace keeps the first
            first_row_version = lay.version;
        }
        auto it = first_row.find(id);
        return it == first_row.end() ? -1 : it->second;
    }
    const float* rows_dev() const override { return V.as<float>(); }

    // IVFIndex.WriteTo ivf_index.go:468-588: Flush; "IVFX", version, dim, kind, nlist, trained byte, centroids
    // {size, floats} when trained, list count, per list {size, per vector {id, dim floats}}, empty bitmap.
    void write_to(Sink& s) override {
        flush();
        lay.compile(c);
        write_header(s, "IVFX", dim, metric);
        s.u32((uint32_t)nlist);
        s.u8(trained ? 1 : 0);
        if (trained) {
            std::vector<float> cen((size_t)nlist * dim);
            get_centroids(cen.data());
            for (int l = 0; l < nlist; l++) { s.u32((uint32_t)dim); s.put(&cen[(size_t)l * dim], (size_t)dim * 4); }
        }
        s.u32((uint32_t)nlist);
        const int64_t chunk = 16384;
        ScratchMark sm(c);
        const size_t cap = (size_t)std::min<int64_t>(chunk, std::max<int64_t>(lay.n, 1));
        int32_t* didx = c->salloc<int32_t>(cap);
        float* padded = c->salloc<float>(cap * ld);
        float* dense = c->salloc<float>(cap * dim);
        std::vector<float> host(cap * dim); std::vector<int32_t> hidx(cap);
        for (int l = 0; l < nlist; l++) {
            const int len = lay.len_h[l];
            s.u32((uint32_t)len);
            for (int j0 = 0; j0 < len; j0 += (int)chunk) {
                const int m = (int)std::min<int64_t>(chunk, len - j0);
                for (int j = 0; j < m; j++) hidx[j] = (int32_t)lay.row_of_slot_h[lay.base_h[l] + j0 + j];
                c->h2d(didx, hidx.data(), (size_t)m * 4);
                launch_gather_rows(c, V.as<float>(), ld, didx, m, padded);
                launch_unpad_rows(c, padded, m, ld, dense, dim);
                c->d2h(host.data(), dense, (size_t)m * dim * 4);
                HIP_CHECK(hipStreamSynchronize(c->stream));
                for (int j = 0; j < m; j++) { s.u32(lay.ids[hidx[j]]); s.put(&host[(size_t)j * dim], (size_t)dim * 4); }
            }
        }
        write_empty_bitmap(s);
        s.flush();
    }
    // IVFIndex.ReadFrom ivf_index.go:620-785 (parsed into scratch state, committed at the end like :778-782)
    void read_from(Source& s) override {
        read_header(s, "IVFX", dim, metric);
        const uint32_t nl = s.u32("nlist");
        if ((int)nl != nlist) COMET_FAIL(COMET_ERR_FORMAT, "nlist mismatch: index has nlist=%d, serialized data has nlist=%u", nlist, nl);   // :676
        const bool tr = s.u8("trained flag") == 1;
        DevBuf ncent, nV; ListLayout nlay; nlay.nlist = nlist; nlay.align = 1;
        if (tr) {
            std::vector<float> cen((size_t)nlist * dim);
            for (int l = 0; l < nlist; l++) {
                const uint32_t cs = s.u32("centroid size");
                if ((int)cs != dim) COMET_FAIL(COMET_ERR_FORMAT, "centroid %d has %u components, expected %d", l, cs, dim);
                s.get(&cen[(size_t)l * dim], (size_t)dim * 4, "centroid data");
            }
            ScratchMark sm(c);
            float* raw = c->salloc<float>((size_t)nlist * dim);
            c->h2d(raw, cen.data(), cen.size() * 4);
            ncent.reserve((size_t)nlist * ld * 4, c->stream, 0);
            launch_ingest_rows(c, COMET_L2SQ, raw, nlist, dim, ncent.as<float>(), ld, nullptr);
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        const uint32_t lc = s.u32("list count");
        if ((int)lc != nlist) COMET_FAIL(COMET_ERR_FORMAT, "list count %u does not match nlist %d", lc, nlist);
        const int64_t chunk = 16384;
        std::vector<float> host((size_t)chunk * dim); std::vector<uint32_t> hid(chunk); std::vector<int32_t> hl(chunk);
        int64_t fill = 0;
        auto spill = [&]() {
            if (!fill) return;
            ScratchMark sm(c);
            float* dv = c->salloc<float>((size_t)fill * dim);
            c->h2d(dv, host.data(), (size_t)fill * dim * 4);
            nV.reserve((size_t)(nlay.n + fill) * ld * 4, c->stream, (size_t)nlay.n * ld * 4);
            launch_ingest_rows(c, COMET_L2SQ, dv, fill, dim, nV.as<float>() + (size_t)nlay.n * ld, ld, nullptr);   // stored vectors are already preprocessed
            HIP_CHECK(hipStreamSynchronize(c->stream));
            nlay.append(hid.data(), hl.data(), fill);
            fill = 0;
        };
        for (uint32_t l = 0; l < lc; l++) {
            const uint32_t len = s.u32("list size");
            for (uint32_t j = 0; j < len; j++) {
                hid[fill] = s.u32("vector ID"); hl[fill] = (int32_t)l;
                s.get(&host[(size_t)fill * dim], (size_t)dim * 4, "vector data");
                if (++fill == chunk) spill();
            }
        }
        spill();
        const std::vector<uint32_t> del = read_bitmap(s);
        // commit
        trained = tr;
        forget_placement();       // host-side training state, not in the stream: stale counts of an earlier Train must not outlive it (every loader falls back to l % world)
        std::swap(centroids.p, ncent.p); std::swap(centroids.cap, ncent.cap);
        std::swap(V.p, nV.p); std::swap(V.cap, nV.cap);
        lay.ids.swap(nlay.ids); lay.list_of.swap(nlay.list_of); lay.id_count.swap(nlay.id_count); lay.n = nlay.n; lay.dirty = true;
        deleted.clear(); deleted.insert(del.begin(), del.end()); deleted_dirty = true;
    }

    // IVFIndex.Train ivf_index.go:206-235
    void train_dev(const float* vecs_dev, int64_t n) override {
        if (n < nlist) COMET_FAIL(COMET_ERR_TRAIN_DATA, "need at least %d training vectors for %d clusters (got %lld)", nlist, nlist, (long long)n);
        float* Vt = pad_raw(c, vecs_dev, n, dim, ld);
        centroids.reserve((size_t)nlist * ld * sizeof(float), c->stream, 0);
        int32_t* assign = c->salloc<int32_t>(n);
        if (kmeans_device(c, metric, Vt, n, ld, nlist, 20, centroids.as<float>(), assign) == 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "k-means clustering failed");
        assign_nearest(c, metric, Vt, n, ld, centroids.as<float>(), nlist, assign);     // against the FINAL centroids (what Add will do): the list-length estimate of the shard assignment
        note_training(this, c, assign, n, nlist);
        trained = true;
    }
    // IVFIndex.Add ivf_index.go:251-280
    int64_t add_dev(const uint32_t*, const uint32_t* ids_h, const float* vecs_dev, int64_t m, int64_t* zero_at, float* normalized_dev) override {
        *zero_at = -1;
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained before adding vectors");
        if (m <= 0) return 0;
        const bool sharded = shard_world > 1;
        if (!sharded) V.reserve((size_t)(lay.n + m) * ld * sizeof(float), c->stream, (size_t)lay.n * ld * sizeof(float));
        float* dst = sharded ? c->salloc<float>((size_t)m * ld) : V.as<float>() + (size_t)lay.n * ld;
        int32_t* zf = c->salloc<int32_t>(m);
        launch_ingest_rows(c, metric, vecs_dev, m, dim, dst, ld, zf);
        int64_t added = m;
        if (metric == COMET_COSINE) {
            std::vector<int32_t> h(m);
            c->d2h(h.data(), zf, m * sizeof(int32_t));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < m; i++) if (h[i]) { *zero_at = i; added = i; break; }
        }
        if (added > 0) {
            int32_t* a = c->salloc<int32_t>(added);
            assign_nearest(c, metric, dst, added, ld, centroids.as<float>(), nlist, a);   // FindNearestCentroidIndex
            std::vector<int32_t> ah(added);
            c->d2h(ah.data(), a, added * sizeof(int32_t));
            if (normalized_dev) launch_unpad_rows(c, dst, added, ld, normalized_dev, dim);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (sharded) {     // keep the rows of the lists this rank owns
                const std::vector<int32_t> keep = owned_rows(ah, this);
                if (!keep.empty()) {
                    int32_t* kd = c->salloc<int32_t>(keep.size());
                    c->h2d(kd, keep.data(), keep.size() * 4);
                    V.reserve((size_t)(lay.n + keep.size()) * ld * sizeof(float), c->stream, (size_t)lay.n * ld * sizeof(float));
                    launch_gather_rows(c, dst, ld, kd, (int64_t)keep.size(), V.as<float>() + (size_t)lay.n * ld);
                    std::vector<uint32_t> kid(keep.size()); std::vector<int32_t> kl(keep.size());
                    for (size_t i = 0; i < keep.size(); i++) { kid[i] = ids_h[keep[i]]; kl[i] = ah[keep[i]]; }
                    HIP_CHECK(hipStreamSynchronize(c->stream));
                    lay.append(kid.data(), kl.data(), (int64_t)keep.size());
                }
            } else {
                lay.append(ids_h, ah.data(), added);
            }
        }
        return added;
    }
    void flush() override {
        if (deleted.empty()) return;
        auto keep = lay.survivors(deleted);
        compact_rows(c, V, (size_t)ld * sizeof(float), keep);
        lay.keep_rows(keep);
        deleted.clear(); deleted_dirty = true;
    }
    // ---- MFMA fast path (kernels_ivf.hip) ----
    void build_shadow() {
        lay.compile(c);
        if (shadow_version == lay.version) return;
        const int64_t ns = std::max<int64_t>(lay.nslots, 64);
        Vh.reserve((size_t)ns * ldh * 2, c->stream, 0);
        rn_slot.reserve((size_t)ns * 4, c->stream, 0);
        stats_dev.reserve(16, c->stream, 0);
        c->zero(stats_dev.p, 16);
        launch_ivf_shadow(c, V.as<float>(), ld, lay.row_of_slot.as<uint32_t>(), lay.nslots, Vh.p, ldh, rn_slot.as<float>(), stats_dev.as<uint32_t>());
        if (i8_policy != 0 && prep_queries_i8_ok(dim)) {
            const int64_t nu = ceil_div(ns, 64);
            V8.reserve((size_t)nu * 64 * ld8, c->stream, 0);
            su.reserve((size_t)nu * 4, c->stream, 0);
            launch_ivf_shadow_i8(c, V.as<float>(), ld, lay.row_of_slot.as<uint32_t>(), lay.nslots, V8.p, ld8, su.as<float>(), stats_dev.as<uint32_t>());
        }
        uint32_t hs[4] = {0, 0, 0, 0};
        c->d2h(hs, stats_dev.p, 16);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        std::memcpy(&xmax_abs, &hs[0], 4); std::memcpy(&xmax_norm2, &hs[1], 4); std::memcpy(&xmax_err2, &hs[2], 4);
        shadow_version = lay.version;
    }
    bool fast_usable(int B, const comet_search_params& p) {
        if (p.mode == 1 || lay.n < 1 || B < 1) return false;
        static const bool off = getenv("COMET_IVF_STRICT") != nullptr;
        if (off && p.mode != 2) return false;
        if (nlist > ivf_fast_max_lists() || p.k < 1 || p.k > 1024) return false;
        build_shadow();
        return std::isfinite(xmax_abs) && xmax_abs <= 60000.0f && std::isfinite(xmax_norm2);
    }
    // up to 256 raw queries through the fast path; writes the final ids / scores / counts of the slice
    void search_fast(const float* queries_dev, int bn, const comet_search_params& p, const uint8_t* elig, uint32_t* out_ids, float* out_scores,
                     int32_t* out_counts, int k_cap, Pending* pend) {
        ScratchMark sm(c);
        const int np = sanitize_nprobes(p.nprobes, nlist);
        const int NB = kFastBatch;
        float* Qp = c->salloc<float>((size_t)bn * ld);
        int32_t* zflag = c->salloc<int32_t>(bn);
        const bool i8 = i8_usable();
        void* Qh = i8 ? c->scratch_alloc((size_t)NB * ld8 * 2) : c->scratch_alloc((size_t)NB * ldh * 2 * 2);      // row-major copy (the scan gathers its query rows from it) + the fragment-ordered copy of the Flat scan
        void* Q8R = i8 ? (char*)Qh + (size_t)NB * ld8 : nullptr;
        float* sqv = i8 ? c->salloc<float>(NB) : nullptr;
        float* qn = c->salloc<float>(NB);
        float* err = c->salloc<float>(NB);
        int32_t* flags = pend->dflags + (size_t)pend->nfast_slices * kSliceInts;
        int32_t* ovf = flags; int32_t* st = flags + 256;
        const int fmode = metric == COMET_COSINE ? 0 : 1;
        const float xn2 = metric == COMET_COSINE ? 1.0002f : xmax_norm2;
        if (i8) launch_prep_queries_i8(c, metric, queries_dev, bn, dim, Qp, ld, zflag, Qh, Q8R, ld8, sqv, qn, err, fmode, xn2, std::sqrt(xmax_err2), st);
        else if (prep_queries_fused_ok(dim)) launch_prep_queries_fused(c, metric, queries_dev, bn, dim, Qp, ld, zflag, Qh, ldh, qn, err, fmode, xn2, st);
        else { launch_ingest_rows(c, metric, queries_dev, bn, dim, Qp, ld, zflag); launch_prep_queries_fast(c, Qp, bn, ld, dim, Qh, ldh, qn, err, fmode, xn2, st); }
        uint32_t* probe_list = c->salloc<uint32_t>((size_t)bn * np);
        int32_t* seg_off = c->salloc<int32_t>((size_t)bn * (np + 1));
        int32_t* uoff = c->salloc<int32_t>((size_t)bn * (np + 1));
        if (!coarse_probe(c, metric, centroids.as<float>(), nlist, ld, dim, Qp, bn, np, probe_list, lay.list_len.as<int32_t>(), seg_off, nullptr, uoff))
            launch_ivf_probe_units(c, probe_list, np, lay.list_len.as<int32_t>(), bn, np, seg_off, uoff);
        const int64_t umax = std::max<int64_t>(lay.max_units(np), 1);
        const int64_t ldD = umax * ivf_fast_unit_rows();             // score row of a query: its probed lists' 64-row units in probe order
        const int64_t P = (int64_t)bn * np;
        const int64_t gmax = P / 64 + std::min<int64_t>(nlist, P) + 1;
        const int64_t imax = ceil_div(P, 64) * ceil_div(std::max(lay.max_len, 1), 256) + lay.total_tiles256 + 1;
        void* groups = c->scratch_alloc((size_t)gmax * ivf_group_bytes());
        void* items = c->scratch_alloc((size_t)imax * ivf_item_bytes());
        DevBuf& sc_buf = scan_counts[c->cur_lane]; last_lane = c->cur_lane;
        sc_buf.reserve(16, c->stream, 0);
        int32_t* counts = sc_buf.as<int32_t>();            // persistent: get_stat reads the last launch's figures
        launch_ivf_items(c, probe_list, np, np, uoff, (int)P, nlist, lay.list_len.as<int32_t>(), lay.list_base.as<int64_t>(), groups, items, counts);
        float* D = c->salloc<float>((size_t)bn * ldD);
        float* umin = c->salloc<float>((size_t)bn * umax);          // per (query, unit): the unit's smallest approximate distance
        if (i8) { launch_ivf_scan_i8(c, fmode, V8.p, ld8, Q8R, rn_slot.as<float>(), qn, su.as<float>(), sqv, elig, groups, items, counts, D, ldD, umin, umax); pend->i8_mask |= 1ull << pend->nfast_slices; }
        else launch_ivf_scan_f16(c, fmode, Vh.p, ldh, Qh, rn_slot.as<float>(), qn, elig, groups, items, counts, D, ldD, umin, umax);
        launch_ivf_post(c, metric, D, ldD, umin, umax, uoff, np, probe_list, np, lay.list_base.as<int64_t>(), lay.list_len.as<int32_t>(), lay.row_of_slot.as<uint32_t>(),
                        lay.ids_slot.as<uint32_t>(), elig, err, p.k, p.threshold, V.as<float>(), ld, Qp, bn, zflag, out_ids, out_scores, out_counts, k_cap, ovf, st);
        pend->nfast_slices++;
    }
    void search_core(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores, int32_t* out_counts, int k_cap, Pending* pend) {
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained before searching");
        lay.compile(c);
        // queries per slice: the score rows of a slice (slice x max units of a query x 64 floats) stay under 4 GiB
        int slice = kFastBatch;
        bool fast = pend && fast_usable(B, p);
        if (fast) {
            const int64_t row_bytes = std::max<int64_t>(lay.max_units(sanitize_nprobes(p.nprobes, nlist)), 1) * ivf_fast_unit_rows() * 4;
            slice = (int)std::min<int64_t>(kFastBatch, ((int64_t)4 << 30) / row_bytes);
            fast = slice >= 1 && ceil_div(B, slice) <= kMaxSlices;
        }
        if (p.mode == 2 && !fast) COMET_FAIL(COMET_ERR_UNSUPPORTED, "fast path unavailable for this index / k (values beyond fp16 range, k < 1 or k > 1024, too many lists)");
        if (!fast) { search_strict(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap); st_strict_queries += B; return; }
        const uint8_t* elig = nullptr;
        int nf = 0;
        const uint32_t* flt = filter_sorted_scratch(p, &nf);
        const uint32_t* del = deleted.empty() ? nullptr : deleted_sorted_dev();
        const int nd = deleted.empty() ? 0 : n_deleted_dev;
        if (nd > 0 || nf > 0) {
            uint8_t* e = c->salloc<uint8_t>(lay.nslots);
            launch_build_elig(c, lay.ids_slot.as<uint32_t>(), lay.nslots, del, nd, flt, nf, e);
            elig = e;
        }
        pend->slice = slice;
        for (int b0 = 0; b0 < B; b0 += slice) {
            const int bn = std::min(slice, B - b0);
            search_fast(queries_dev + (size_t)b0 * dim, bn, p, elig, out_ids + (size_t)b0 * k_cap, out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap, pend);
        }
    }
    uint64_t search_begin(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                          int32_t* out_counts, int k_cap) override {
        Pending* slot = nullptr;
        for (auto& r : ring) if (!r.active) { slot = &r; break; }
        if (!slot) {
            Pending* oldest = &ring[0];
            for (auto& r : ring) if (r.ticket < oldest->ticket) oldest = &r;
            search_finish(oldest->ticket);
            slot = oldest;
        }
        if (!slot->ev) HIP_CHECK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
        if (!slot->ev_post) HIP_CHECK(hipEventCreateWithFlags(&slot->ev_post, hipEventDisableTiming));
        if (!slot->flags) HIP_CHECK(hipHostMalloc((void**)&slot->flags, sizeof(int32_t) * kSliceInts * kMaxSlices, hipHostMallocDefault));
        if (!slot->dflags) HIP_CHECK(hipMalloc((void**)&slot->dflags, sizeof(int32_t) * kSliceInts * kMaxSlices));
        copy_stream = c->stream;      // the flag copy follows the search on its own lane (round 5; a private stream until then: see below)
        slot->ticket = next_ticket++; slot->B = B; slot->k_cap = k_cap; slot->nfast_slices = 0; slot->i8_mask = 0; slot->queries = queries_dev;
        n_searches++;
        slot->p = p; slot->flt.clear();
        if (p.filter_ids && p.n_filter > 0) { slot->flt.assign(p.filter_ids, p.filter_ids + p.n_filter); slot->p.filter_ids = slot->flt.data(); }
        slot->out_ids = out_ids; slot->out_scores = out_scores; slot->out_counts = out_counts;
        st_candidates = st_overflows = st_expansions = st_fast_queries = st_strict_queries = st_i8_slices = 0;
        search_core(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap, slot);
        if (slot->nfast_slices > 0) {
            HIP_CHECK(hipEventRecord(slot->ev_post, c->stream));
            HIP_CHECK(hipStreamWaitEvent(copy_stream, slot->ev_post, 0));
            HIP_CHECK(hipMemcpyAsync(slot->flags, slot->dflags, sizeof(int32_t) * kSliceInts * slot->nfast_slices, hipMemcpyDeviceToHost, copy_stream));
            HIP_CHECK(hipEventRecord(slot->ev, copy_stream));
        } else HIP_CHECK(hipEventRecord(slot->ev, c->stream));
        slot->active = true;
        return slot->ticket;
    }
    bool search_finish(uint64_t ticket) override {
        Pending* slot = nullptr;
        for (auto& r : ring) if (r.active && r.ticket == ticket) { slot = &r; break; }
        if (!slot) return false;
        HIP_CHECK(hipEventSynchronize(slot->ev));
        slot->active = false;
        std::vector<int> redo;
        for (int sl = 0; sl < slot->nfast_slices; sl++) {
            const int32_t* hf = slot->flags + (size_t)sl * kSliceInts;
            const int bn = std::min(slot->slice, slot->B - sl * slot->slice);
            st_candidates += hf[256]; st_overflows += hf[257]; st_expansions += hf[258];
            if ((slot->i8_mask >> sl) & 1ull) {
                st_i8_slices++;
                if (i8_policy != 1 && (hf[257] > 0 || hf[256] > kI8MaxCand * bn) && n_searches >= i8_resume_at) {
                    i8_resume_at = n_searches + ((int64_t)32 << std::min(i8_strikes, 10)); i8_strikes++; st_i8_backoffs++;
                }
            }
            int nfast = bn;
            for (int q = 0; q < bn; q++) if (hf[q]) { redo.push_back(sl * slot->slice + q); nfast--; }
            st_fast_queries += nfast;
        }
        comet_search_params sp = slot->p; sp.mode = 1;
        for (int q : redo) {
            ScratchMark sm(c);
            search_core(slot->queries + (size_t)q * dim, 1, sp, slot->out_ids + (size_t)q * slot->k_cap, slot->out_scores + (size_t)q * slot->k_cap,
                        slot->out_counts + q, slot->k_cap, nullptr);
        }
        if (!redo.empty()) HIP_CHECK(hipStreamSynchronize(c->stream));
        return !redo.empty();
    }
    void search_dev(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                    int32_t* out_counts, int k_cap) override {
        search_finish(search_begin(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap));
    }
    bool get_stat(const char* name, double* out) const override {
        std::string k(name);
        if (k == "fast_candidates") *out = (double)st_candidates;
        else if (k == "fast_overflows") *out = (double)st_overflows;
        else if (k == "fast_expansions") *out = (double)st_expansions;
        else if (k == "fast_queries") *out = (double)st_fast_queries;
        else if (k == "strict_queries") *out = (double)st_strict_queries;
        else if (k == "i8_slices") *out = (double)st_i8_slices;
        else if (k == "i8_backoffs") *out = (double)st_i8_backoffs;
        else if (k == "i8_max_residual") *out = std::sqrt((double)xmax_err2);
        else if (k == "max_list_len") { const_cast<ListLayout&>(lay).compile(c); *out = (double)lay.max_len; }
        else if (k == "ivf_scan_rows" || k == "ivf_scan_items") {       // rows the last fast slice's scan streamed (x ldh x 2 = its algorithmic bytes) / its work items
            int32_t h[4] = {0, 0, 0, 0};
            if (scan_counts[last_lane].p) { c->quiesce_all(); c->d2h(h, scan_counts[last_lane].p, 16); HIP_CHECK(hipStreamSynchronize(c->stream)); }
            *out = k == "ivf_scan_rows" ? (double)h[2] * 64.0 : (double)h[0];
        }
        else return false;
        return true;
    }

    // ivfIndexSearch.searchSingleQuery ivf_index_search.go:217-322 — the exact kernels (search mode 1, and whatever the fast path does not take)
    void search_strict(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                       int32_t* out_counts, int k_cap) {
        float* Qp; int32_t* zflag;
        prepare_queries(c, metric, queries_dev, B, dim, ld, &Qp, &zflag, true);
        const int np = sanitize_nprobes(p.nprobes, nlist);
        uint32_t* probe_list = c->salloc<uint32_t>((size_t)B * np);
        int32_t* seg_off = c->salloc<int32_t>((size_t)B * (np + 1));
        int32_t* cnts = c->salloc<int32_t>(B);
        if (!coarse_probe(c, metric, centroids.as<float>(), nlist, ld, dim, Qp, B, np, probe_list, lay.list_len.as<int32_t>(), seg_off, cnts))
            launch_probe_segments(c, probe_list, np, nullptr, lay.list_len.as<int32_t>(), B, np, seg_off, cnts);
        const int64_t Cmax = lay.max_candidates(np);
        uint32_t* pos = c->salloc<uint32_t>((size_t)B * k_cap);
        if (Cmax > 0) {
            const uint8_t* elig = nullptr;
            int nf = 0;
            const uint32_t* flt = filter_sorted_scratch(p, &nf);
            const uint32_t* del = deleted.empty() ? nullptr : deleted_sorted_dev();
            const int nd = deleted.empty() ? 0 : n_deleted_dev;
            if (nd > 0 || nf > 0) {
                uint8_t* e = c->salloc<uint8_t>(lay.nslots);
                launch_build_elig(c, lay.ids_slot.as<uint32_t>(), lay.nslots, del, nd, flt, nf, e);
                elig = e;
            }
            const int64_t ldR = round_up(Cmax, 16);
            const size_t budget = (size_t)2 << 30;
            int qb = (int)std::max<int64_t>(1, std::min<int64_t>(B, (int64_t)(budget / ((size_t)ldR * 4))));
            float* D = c->salloc<float>((size_t)qb * ldR);
            uint32_t* order = c->salloc<uint32_t>((size_t)qb * np);
            uint32_t* olist = c->salloc<uint32_t>((size_t)qb * np);
            for (int b0 = 0; b0 < B; b0 += qb) {
                const int bn = std::min(qb, B - b0);
                const uint32_t* pl = probe_list + (size_t)b0 * np;
                const int32_t* so = seg_off + (size_t)b0 * (np + 1);
                // (query, list) pairs in list order: the scans of one inverted list run back to back and share its rows through L2
                // groups pay off once lists are shared: on average at least two (query, list) pairs per list
                const bool grouped = launch_order_pairs(c, pl, np, np, so, bn * np, nlist, order, olist) && (int64_t)bn * np >= 2 * (int64_t)nlist;
                launch_dist_list(c, metric, V.as<float>(), ld, Qp + (size_t)b0 * ld, order, grouped ? olist : nullptr, bn * np, nlist, np, pl, np, so, lay.list_base.as<int64_t>(),
                                 lay.list_len.as<int32_t>(), lay.row_of_slot.as<uint32_t>(), elig, lay.max_len, D, ldR);
                launch_select_topk(c, D, ldR, bn, Cmax, cnts + b0, p.threshold, p.k, pos + (size_t)b0 * k_cap,
                                   out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap);
            }
        } else {
            launch_select_topk(c, nullptr, 0, B, 0, nullptr, 0.0f, p.k, pos, out_scores, out_counts, k_cap);
        }
        launch_finalize_probe(c, pos, B, k_cap, probe_list, np, seg_off, np, lay.list_base.as<int64_t>(), lay.ids_slot.as<uint32_t>(),
                              zflag, out_ids, out_counts);
    }
    void get_centroids(float* out) const override {
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained");
        float* tmp = c->salloc<float>((size_t)nlist * dim);
        launch_unpad_rows(c, centroids.as<float>(), nlist, ld, tmp, dim);
        c->d2h(out, tmp, (size_t)nlist * dim * sizeof(float));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    int64_t list_size(int l) const override { const_cast<ListLayout&>(lay).compile(c); return (l >= 0 && l < nlist) ? lay.len_h[l] : 0; }
    void list_read(int l, uint32_t* oids, uint8_t*, float* ovecs) const override {
        auto& L = const_cast<ListLayout&>(lay); L.compile(c);
        if (l < 0 || l >= nlist) COMET_FAIL(COMET_ERR_INVALID_ARG, "list out of range");
        const int len = L.len_h[l];
        std::vector<float> row(ld);
        for (int j = 0; j < len; j++) {
            uint32_t r = L.row_of_slot_h[L.base_h[l] + j];
            if (oids) oids[j] = L.ids[r];
            if (ovecs) { c->d2h(row.data(), V.as<float>() + (size_t)r * ld, (size_t)ld * 4); HIP_CHECK(hipStreamSynchronize(c->stream)); std::copy(row.begin(), row.begin() + dim, ovecs + (size_t)j * dim); }
        }
    }
};

comet_index* make_ivf(Ctx* c, int dim, int metric, int nlist) {
    auto* f = new IVFIndex();
    f->c = c; f->kind = COMET_KIND_IVF; f->dim = dim; f->ld = padded_dim(dim); f->metric = metric; f->nlist = nlist;
    f->ldh = (int)round_up(dim, 64); f->ld8 = (int)round_up(dim, 128);
    f->lay.nlist = nlist; f->lay.align = ivf_fast_unit_rows();   // every list starts on a 64-slot key unit of the fast path's shadow
    return f;
}

// ------------------------------------------------------------------------------------------------
// PQIndex and IVFPQIndex share one implementation: PQ is the one-list, no-centroid case.
// ------------------------------------------------------------------------------------------------
struct PQFamilyIndex : comet_index {
    bool ivf = false;
    int nlist = 1, M = 0, nbits = 0, Ksub = 0, dsub = 0, M4 = 0;
    int n_lists() const override { return nlist; }
    DevBuf centroids;   // nlist x ld (IVFPQ only)
    DevBuf codebooks;   // M x Ksub x dsub dense fp32 (pq_index.go:99-101 layout)
    DevBuf codes_arr;   // arrival-order codes, n x M4 words (byte m of a row = code[m])
    DevBuf codes_il;    // compiled, block-interleaved
    DevBuf adc_stats;   // two-stage search counters (AdcFilter::stats), read by get_stat
    mutable bool stats_on = false;   // adc_* counters are collected (get_stat "adc_stats_on" / "adc_stats_off")
    // Adaptive staging of the fused search (IVFPQ, one GPU). The two-stage search pays a lower-bound kernel and a second set of launches to remove the (query, list)
    // pairs that cannot matter — 97 % of them on clustered data, NONE on data without cluster structure (SURVEY 8d's uniform rows: pairs_left_alive 1.0), where it
    // is pure cost. Every kAutoEvery-th search runs two-stage with the counters on (a buffer of its own); their read-back is asynchronous (pinned slot + event,
    // polled by later searches, never waited for); when more than half of the pairs behind the nearest lists survived, the searches in between run single-stage.
    // Results are bit-identical in every mode (tests/test_quant_gpu.py), so the switch is invisible. Never on a list shard: the ranks' collectives must pair up.
    static constexpr int64_t kAutoEvery = 32;
    DevBuf auto_stats; int32_t* auto_host = nullptr; hipEvent_t auto_ev = nullptr; bool auto_pending = false, auto_one = false; int64_t auto_n = 0;
    ~PQFamilyIndex() override {
        if (auto_ev) { (void)hipEventSynchronize(auto_ev); (void)hipEventDestroy(auto_ev); }
        if (auto_host) (void)hipHostFree(auto_host);
    }
    DevBuf bound_tab3, bound_cmax2;   // the lower bound's transposed codebook (IVFPQ, Ksub == 256; rebuilt with the interleaved codes)
    DevBuf list_rmax;   // per list: upper bound on the norm of its members' decoded residuals (IVFPQ; rebuilt with the interleaved codes)
    bool il_dirty = true;
    ListLayout lay;

    int64_t size() const override { return lay.n; }
    int max_lanes() const override { return 4; }      // searches write scratch, their ring slot / per-lane counters and (opt-in) atomics only
    int default_nprobes() const override { return ivf ? (int)std::sqrt((double)nlist) : 0; }   // ivfpq_index.go:442-449
    bool contains_id(uint32_t id) const override { return lay.id_count.count(id) != 0; }

    // learn M codebooks on the rows of R (n x ld): KMeansSubspace per subspace (pq_index.go:218-246, ivfpq_index.go:232-256)
    void train_codebooks(const float* R, int64_t n) {
        codebooks.reserve((size_t)M * Ksub * dsub * sizeof(float), c->stream, 0);
        const int lds = padded_dim(dsub);
        float* sub = c->salloc<float>((size_t)n * lds);
        float* cent = c->salloc<float>((size_t)std::min<int64_t>(Ksub, n) * lds);
        int32_t* assign = c->salloc<int32_t>(n);
        for (int m = 0; m < M; m++) {
            launch_extract_sub(c, R, ld, n, m * dsub, dsub, sub, lds);
            int k = kmeans_device(c, COMET_L2SQ, sub, n, lds, Ksub, 20, cent, assign);
            if (k == 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "k-means failed for subspace %d", m);
            if (k < Ksub) COMET_FAIL(COMET_ERR_TRAIN_DATA, "need at least %d vectors for training", Ksub);  // the reference would index past the centroids slice
            launch_unpad_rows(c, cent, Ksub, lds, codebooks.as<float>() + (size_t)m * Ksub * dsub, dsub);
        }
    }
    void train_dev(const float* vecs_dev, int64_t n) override {
        if (ivf) {   // IVFPQIndex.Train ivfpq_index.go:180-259
            if (n < (int64_t)nlist * 10) COMET_FAIL(COMET_ERR_TRAIN_DATA, "need at least %d vectors for training", nlist * 10);
            float* Vt = pad_raw(c, vecs_dev, n, dim, ld);
            centroids.reserve((size_t)nlist * ld * sizeof(float), c->stream, 0);
            int32_t* assign = c->salloc<int32_t>(n);
            if (kmeans_device(c, metric, Vt, n, ld, nlist, 20, centroids.as<float>(), assign) == 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "IVF k-means failed");
            assign_nearest(c, metric, Vt, n, ld, centroids.as<float>(), nlist, assign);       // STEP 2 (:205-208)
            note_training(this, c, assign, n, nlist);
            float* R = c->salloc<float>((size_t)n * ld);
            launch_residual_rows(c, Vt, ld, n, centroids.as<float>(), assign, R);              // STEP 3 (:213-224)
            train_codebooks(R, n);                                                              // STEP 4
        } else {     // PQIndex.Train pq_index.go:193-250
            if (n < Ksub) COMET_FAIL(COMET_ERR_TRAIN_DATA, "need at least %d vectors for training", Ksub);
            float* Vt = pad_raw(c, vecs_dev, n, dim, ld);
            train_codebooks(Vt, n);
        }
        trained = true;
    }
    // PQIndex.Add pq_index.go:263-289 / IVFPQIndex.Add ivfpq_index.go:279-319
    int64_t add_dev(const uint32_t*, const uint32_t* ids_h, const float* vecs_dev, int64_t m, int64_t* zero_at, float* normalized_dev) override {
        *zero_at = -1;
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained before adding");
        if (m <= 0) return 0;
        float* P = c->salloc<float>((size_t)m * ld);
        int32_t* zf = c->salloc<int32_t>(m);
        launch_ingest_rows(c, metric, vecs_dev, m, dim, P, ld, zf);
        int64_t added = m;
        if (metric == COMET_COSINE) {
            std::vector<int32_t> h(m);
            c->d2h(h.data(), zf, m * sizeof(int32_t));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < m; i++) if (h[i]) { *zero_at = i; added = i; break; }
        }
        if (added <= 0) return 0;
        if (normalized_dev) launch_unpad_rows(c, P, added, ld, normalized_dev, dim);
        std::vector<int32_t> ah;
        std::vector<uint32_t> kid;                       // ids of the rows this rank keeps (list sharding)
        int32_t* a = nullptr;
        int64_t kept = added;
        if (ivf) {
            a = c->salloc<int32_t>(added);
            assign_nearest(c, metric, P, added, ld, centroids.as<float>(), nlist, a);
            ah.resize(added);
            c->d2h(ah.data(), a, added * sizeof(int32_t));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (shard_world > 1) {                         // encode only the members of the lists this rank owns
                const std::vector<int32_t> keep = owned_rows(ah, this);
                kept = (int64_t)keep.size();
                if (kept == 0) return added;
                int32_t* kd = c->salloc<int32_t>(kept);
                c->h2d(kd, keep.data(), kept * 4);
                float* P2 = c->salloc<float>((size_t)kept * ld);
                launch_gather_rows(c, P, ld, kd, kept, P2);
                std::vector<int32_t> kl(kept); kid.resize(kept);
                for (int64_t i = 0; i < kept; i++) { kid[i] = ids_h[keep[i]]; kl[i] = ah[keep[i]]; }
                ah.swap(kl);
                c->h2d(a, ah.data(), kept * 4);
                P = P2; ids_h = kid.data();
            }
        }
        codes_arr.reserve((size_t)(lay.n + kept) * M4 * 4, c->stream, (size_t)lay.n * M4 * 4);
        uint8_t* dst = (uint8_t*)codes_arr.p + (size_t)lay.n * M4 * 4;
        c->zero(dst, (size_t)kept * M4 * 4);
        if (ivf) {
            float* R = c->salloc<float>((size_t)kept * ld);
            launch_residual_rows(c, P, ld, kept, centroids.as<float>(), a, R);
            launch_pq_encode(c, R, ld, kept, codebooks.as<float>(), M, Ksub, dsub, dst, M4 * 4);
        } else {
            launch_pq_encode(c, P, ld, kept, codebooks.as<float>(), M, Ksub, dsub, dst, M4 * 4);
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        lay.append(ids_h, ivf ? ah.data() : nullptr, kept);
        il_dirty = true;
        return added;
    }
    void flush() override {
        if (deleted.empty()) return;
        auto keep = lay.survivors(deleted);
        compact_rows(c, codes_arr, (size_t)M4 * 4, keep);
        lay.keep_rows(keep);
        il_dirty = true;
        deleted.clear(); deleted_dirty = true;
    }
    void compile() {
        lay.compile(c);
        if (!il_dirty) return;
        il_dirty = false;
        codes_il.reserve((size_t)(lay.nslots + adc_codes_pad()) * M4 * 4 + 4, c->stream, 0);   // + the scan's read-ahead past the last block
        launch_interleave_codes(c, codes_arr.as<uint32_t>(), M4, lay.row_of_slot.as<uint32_t>(), lay.nslots, codes_il.as<uint32_t>());
        if (ivf && Ksub == 256) {
            bound_tab3.reserve((size_t)64 * (dsub + 1) * M * 4 * 4, c->stream, 0); bound_cmax2.reserve((size_t)M * 4, c->stream, 0);
            launch_pq_bound_tab3(c, codebooks.as<float>(), M, dsub, bound_tab3.as<float>(), bound_cmax2.as<float>());
        }
        if (ivf) {      // R(list) for the table-free lower bound of the two-stage search
            list_rmax.reserve((size_t)nlist * 4, c->stream, 0);
            launch_pq_list_rmax(c, codebooks.as<float>(), M, Ksub, dsub, codes_arr.as<uint32_t>(), M4, lay.row_of_slot.as<uint32_t>(), lay.list_base.as<int64_t>(),
                                lay.list_len.as<int32_t>(), nlist, list_rmax.as<float>());
        }
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    int sharded_exchanges(int B, const comet_search_params& p, int k_cap, int* per) const override {
        if (per) *per = 0;
        static const bool fuse_off = getenv("COMET_ADC_NO_FUSE") != nullptr;
        if (!ivf || shard_world <= 1 || fuse_off || p.mode == 1 || !(p.k >= 1 && p.k <= ADC_FILTER_MAX_K && p.k <= k_cap)) return 0;
        return adc_exchange_plan(M, Ksub, sanitize_nprobes(p.nprobes, nlist), B, nlist, per);
    }
    // pqIndexSearch.searchSingleQuery pq_index_search.go:218-325 / ivfpqIndexSearch.searchSingleQuery ivfpq_index_search.go:231-341
    void search_dev(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                    int32_t* out_counts, int k_cap) override {
        if (!trained) { if (ivf) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained before searching"); COMET_FAIL(COMET_ERR_NOT_TRAINED, "index not trained"); }
        compile();
        float* Qp; int32_t* zflag;
        prepare_queries(c, metric, queries_dev, B, dim, ld, &Qp, &zflag, true);
        const int np = ivf ? sanitize_nprobes(p.nprobes, nlist) : 1;
        uint32_t* probe_list = c->salloc<uint32_t>((size_t)B * np);
        int32_t* seg_off = c->salloc<int32_t>((size_t)B * (np + 1));
        int32_t* cnts = c->salloc<int32_t>(B);
        bool booked = false;
        if (ivf) booked = coarse_probe(c, metric, centroids.as<float>(), nlist, ld, dim, Qp, B, np, probe_list, lay.list_len.as<int32_t>(), seg_off, cnts);
        else c->zero(probe_list, sizeof(uint32_t) * (size_t)B);
        if (!booked) launch_probe_segments(c, probe_list, np, nullptr, lay.list_len.as<int32_t>(), B, np, seg_off, cnts);
        const int64_t Cmax = lay.max_candidates(np);
        uint32_t* pos = c->salloc<uint32_t>((size_t)B * k_cap);
        if (Cmax > 0) {
            const uint8_t* elig = nullptr;
            int nf = 0;
            const uint32_t* flt = filter_sorted_scratch(p, &nf);
            const uint32_t* del = deleted.empty() ? nullptr : deleted_sorted_dev();
            const int nd = deleted.empty() ? 0 : n_deleted_dev;
            if (nd > 0 || nf > 0) {
                uint8_t* e = c->salloc<uint8_t>(lay.nslots);
                launch_build_elig(c, lay.ids_slot.as<uint32_t>(), lay.nslots, del, nd, flt, nf, e);
                elig = e;
            }
            const int64_t ldD = round_up(Cmax, 16);
            const size_t budget = (size_t)2 << 30;
            static const bool fuse_off = getenv("COMET_ADC_NO_FUSE") != nullptr;
            const bool fuse = !fuse_off && p.k >= 1 && p.k <= ADC_FILTER_MAX_K && p.k <= k_cap;
            // fused: the scan keeps only the candidates under a running per-query bound (8-byte composites in a row as long as the
            // candidate row, almost all of it never touched); otherwise it writes the distance matrix for the generic selection
            // A sharded search that exchanges stage-1 bounds keeps the whole batch in ONE outer sub-batch: this split depends on the rank's own lists
            // (Cmax), and the ranks' collectives must pair up — the only sub-batching left is launch_adc_scan's, which is the same on every rank
            const bool exchanging = fuse && shard_world > 1 && bound_exchange != nullptr;
            if (exchanging && (size_t)B * (size_t)ldD * 8 > ((size_t)64 << 30))
                COMET_FAIL(COMET_ERR_UNSUPPORTED, "sharded IVFPQ search: %d queries x %lld candidates per query do not fit one sub-batch; search smaller batches", B, (long long)ldD);
            const int qb = exchanging ? B : (int)std::max<int64_t>(1, std::min<int64_t>(B, (int64_t)(budget / ((size_t)ldD * (fuse ? 8 : 4)))));
            float* D = fuse ? nullptr : c->salloc<float>((size_t)qb * ldD);
            AdcFilter afl{};
            bool auto_sample = false;
            if (fuse) {
                afl.cand = c->salloc<unsigned long long>((size_t)qb * ldD); afl.cursor = c->salloc<int32_t>(qb); afl.tq = c->salloc<uint32_t>(qb); afl.K = p.k; afl.thr = p.threshold;
                if (!adc_stats.p) { adc_stats.reserve(32, c->stream, 0); HIP_CHECK(hipMemsetAsync(adc_stats.p, 0, 32, c->stream)); }
                // the counters are atomics on a few words: ~8 k of them per batch from the lower-bound kernel alone — same-address atomics retire
                // one per ~12 ns in the L2, i.e. 0.1 ms of a 0.38 ms search was spent counting. They are taken only while a caller asked for
                // them (get_stat("adc_stats_on") ... get_stat("adc_stats_off")): the bench counts in a pass of its own, outside the timed regions.
                afl.stats = stats_on ? adc_stats.as<int32_t>() : nullptr;
                // mode 1 ("strict"): the reference's literal work — every candidate of every probed list is scored. On a list shard of more
                // than two ranks the single pass is used as well: a rank owns the nearest list of only 1 / world of the queries, the bounds of
                // the others are seeded by a farther list and remove little, and the second set of launches costs more than it saves
                // (tools/shard_probe.py, 1M rows, B = 256: 4 ranks 0.69 ms two-stage vs 0.60 single pass; 2 ranks 0.75 vs 0.80)
                // — unless the ranks exchange their stage-1 bounds (a sharded search through comet_index_search_sharded_async): then every rank
                // prunes with the global bound and the two-stage search pays on every world size
                afl.exchange = shard_world > 1 ? bound_exchange : nullptr; afl.exchange_user = bound_exchange_user;
                static const bool norm_bound_off = getenv("COMET_ADC_NO_NORM_BOUND") != nullptr;
                afl.list_rmax = (ivf && !norm_bound_off && list_rmax.p) ? list_rmax.as<float>() : nullptr;
                afl.bound_tab3 = (ivf && Ksub == 256 && bound_tab3.p) ? bound_tab3.as<float>() : nullptr; afl.bound_cmax2 = afl.bound_tab3 ? bound_cmax2.as<float>() : nullptr;
                afl.one_stage = (p.mode == 1 || (shard_world > 2 && !afl.exchange)) ? 1 : 0;
                static const bool auto_off = getenv("COMET_ADC_NO_AUTO_STAGE") != nullptr;
                if (!auto_off && ivf && shard_world == 1 && p.mode == 0 && !stats_on && np >= 2) {
                    if (auto_pending) {
                        const hipError_t e = hipEventQuery(auto_ev);
                        if (e == hipSuccess) { auto_pending = false; if (auto_host[1] > 0) auto_one = (int64_t)auto_host[0] * 2 > (int64_t)auto_host[1]; }
                        else (void)hipGetLastError();
                    }
                    auto_sample = !auto_pending && (auto_n % kAutoEvery) == 0;
                    auto_n++;
                    if (auto_sample) {
                        auto_stats.reserve(32, c->stream, 0);
                        HIP_CHECK(hipMemsetAsync(auto_stats.p, 0, 32, c->stream));
                        afl.stats = auto_stats.as<int32_t>();
                    } else if (auto_one) afl.one_stage = 1;
                }
            }
            for (int b0 = 0; b0 < B; b0 += qb) {
                const int bn = std::min(qb, B - b0);
                launch_adc_scan(c, Qp + (size_t)b0 * ld, ld, dim, ivf ? centroids.as<float>() : nullptr, codebooks.as<float>(), M, Ksub, dsub,
                                codes_il.as<uint32_t>(), M4, lay.list_base.as<int64_t>(), lay.list_len.as<int32_t>(),
                                probe_list + (size_t)b0 * np, np, np, seg_off + (size_t)b0 * (np + 1), elig, bn, ivf ? nlist : 1, lay.max_len, D, ldD, fuse ? &afl : nullptr, (int64_t)lay.nslots);
                if (fuse) launch_select_composites(c, afl.cand, ldD, afl.cursor, bn, p.k, pos + (size_t)b0 * k_cap, out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap);
                else launch_select_topk(c, D, ldD, bn, Cmax, cnts + b0, p.threshold, p.k, pos + (size_t)b0 * k_cap,
                                        out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap);
            }
            if (auto_sample) {      // the sampled search's counters travel to a pinned slot behind it; a later search polls the event
                if (!auto_host) { HIP_CHECK(hipHostMalloc((void**)&auto_host, 32, hipHostMallocDefault)); HIP_CHECK(hipEventCreateWithFlags(&auto_ev, hipEventDisableTiming)); }
                HIP_CHECK(hipMemcpyAsync(auto_host, auto_stats.p, 32, hipMemcpyDeviceToHost, c->stream));
                HIP_CHECK(hipEventRecord(auto_ev, c->stream));
                auto_pending = true;
            }
        } else {
            // nothing to scan on this rank (an empty shard): its peers still exchange their stage-1 bounds — take part with +inf
            static const bool fuse_off0 = getenv("COMET_ADC_NO_FUSE") != nullptr;
            if (shard_world > 1 && bound_exchange && !fuse_off0 && p.k >= 1 && p.k <= ADC_FILTER_MAX_K && p.k <= k_cap && p.mode != 1) {
                AdcFilter idle{}; idle.tq = c->salloc<uint32_t>(B); idle.exchange = bound_exchange; idle.exchange_user = bound_exchange_user; idle.one_stage = 0;
                adc_exchange_idle(c, &idle, M, Ksub, np, B, ivf ? nlist : 1);
            }
            launch_select_topk(c, nullptr, 0, B, 0, nullptr, 0.0f, p.k, pos, out_scores, out_counts, k_cap);
        }
        launch_finalize_probe(c, pos, B, k_cap, probe_list, np, seg_off, np, lay.list_base.as<int64_t>(), lay.ids_slot.as<uint32_t>(),
                              zflag, out_ids, out_counts);
    }
    void get_centroids(float* out) const override {
        if (!ivf) COMET_FAIL(COMET_ERR_UNSUPPORTED, "PQ index has no centroids");
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained");
        float* tmp = c->salloc<float>((size_t)nlist * dim);
        launch_unpad_rows(c, centroids.as<float>(), nlist, ld, tmp, dim);
        c->d2h(out, tmp, (size_t)nlist * dim * sizeof(float));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    void export_all(uint32_t* oids, int32_t* olists, uint8_t* ocodes) const override {
        if (oids) std::copy(lay.ids.begin(), lay.ids.end(), oids);
        if (olists) std::copy(lay.list_of.begin(), lay.list_of.end(), olists);
        if (ocodes && lay.n > 0) {
            std::vector<uint8_t> all((size_t)lay.n * M4 * 4);
            c->d2h(all.data(), codes_arr.p, all.size());
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < lay.n; i++) std::copy(all.begin() + (size_t)i * M4 * 4, all.begin() + (size_t)i * M4 * 4 + M, ocodes + (size_t)i * M);
        }
    }
    void get_codebooks(float* out) const override {
        if (!trained) COMET_FAIL(COMET_ERR_NOT_TRAINED, "index must be trained");
        c->d2h(out, codebooks.p, (size_t)M * Ksub * dsub * sizeof(float));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }

    // PQIndex.WriteTo pq_index.go:509-640 ("PQIX": dim, kind, M, Nbits, Ksub, dsub, trained, codebooks {size, floats},
    // count, per vector {id, M code bytes}) and IVFPQIndex.WriteTo ivfpq_index.go:544-700 ("IVPQ": dim, kind, nlist, M,
    // Nbits, Ksub, dsub, trained, centroids, codebooks, list count, per list {size, per vector {id, M code bytes}});
    // both Flush first and end with the empty roaring bitmap.
    void write_to(Sink& s) override {
        flush();
        lay.compile(c);
        write_header(s, ivf ? "IVPQ" : "PQIX", dim, metric);
        if (ivf) s.u32((uint32_t)nlist);
        s.u32((uint32_t)M); s.u32((uint32_t)nbits); s.u32((uint32_t)Ksub); s.u32((uint32_t)dsub);
        s.u8(trained ? 1 : 0);
        if (trained) {
            if (ivf) {
                std::vector<float> cen((size_t)nlist * dim);
                get_centroids(cen.data());
                for (int l = 0; l < nlist; l++) { s.u32((uint32_t)dim); s.put(&cen[(size_t)l * dim], (size_t)dim * 4); }
            }
            std::vector<float> cb((size_t)M * Ksub * dsub);
            get_codebooks(cb.data());
            for (int m = 0; m < M; m++) { s.u32((uint32_t)(Ksub * dsub)); s.put(&cb[(size_t)m * Ksub * dsub], (size_t)Ksub * dsub * 4); }
        }
        std::vector<uint8_t> all((size_t)std::max<int64_t>(lay.n, 1) * M4 * 4);
        if (lay.n > 0) { c->d2h(all.data(), codes_arr.p, (size_t)lay.n * M4 * 4); HIP_CHECK(hipStreamSynchronize(c->stream)); }
        if (ivf) s.u32((uint32_t)nlist); else s.u32((uint32_t)lay.n);
        for (int l = 0; l < nlist; l++) {
            const int len = lay.len_h[l];
            if (ivf) s.u32((uint32_t)len);
            for (int j = 0; j < len; j++) {
                const uint32_t r = lay.row_of_slot_h[lay.base_h[l] + j];
                s.u32(lay.ids[r]); s.put(&all[(size_t)r * M4 * 4], (size_t)M);
            }
        }
        write_empty_bitmap(s);
        s.flush();
    }
    // PQIndex.ReadFrom pq_index.go:672-846 / IVFPQIndex.ReadFrom ivfpq_index.go:745-936
    void read_from(Source& s) override {
        read_header(s, ivf ? "IVPQ" : "PQIX", dim, metric);
        if (ivf) { const uint32_t nl = s.u32("nlist"); if ((int)nl != nlist) COMET_FAIL(COMET_ERR_FORMAT, "nlist mismatch: index has nlist=%d, serialized data has nlist=%u", nlist, nl); }
        const uint32_t fM = s.u32("M"), fN = s.u32("Nbits"), fK = s.u32("Ksub"), fd = s.u32("dsub");
        check_param("M", M, fM); check_param("Nbits", nbits, fN); check_param("Ksub", Ksub, fK); check_param("dsub", dsub, fd);   // pq_index.go:737-748
        const bool tr = s.u8("trained flag") == 1;
        DevBuf ncent, ncb, ncodes; ListLayout nlay; nlay.nlist = nlist; nlay.align = 64;
        if (tr) {
            if (ivf) {
                std::vector<float> cen((size_t)nlist * dim);
                for (int l = 0; l < nlist; l++) {
                    const uint32_t cs = s.u32("centroid size");
                    if ((int)cs != dim) COMET_FAIL(COMET_ERR_FORMAT, "centroid %d has %u components, expected %d", l, cs, dim);
                    s.get(&cen[(size_t)l * dim], (size_t)dim * 4, "centroid data");
                }
                ScratchMark sm(c);
                float* raw = c->salloc<float>((size_t)nlist * dim);
                c->h2d(raw, cen.data(), cen.size() * 4);
                ncent.reserve((size_t)nlist * ld * 4, c->stream, 0);
                launch_ingest_rows(c, COMET_L2SQ, raw, nlist, dim, ncent.as<float>(), ld, nullptr);
                HIP_CHECK(hipStreamSynchronize(c->stream));
            }
            std::vector<float> cb((size_t)M * Ksub * dsub);
            for (int m = 0; m < M; m++) {
                const uint32_t sz = s.u32("codebook size");
                if ((int64_t)sz != (int64_t)Ksub * dsub) COMET_FAIL(COMET_ERR_FORMAT, "codebook %d has %u values, expected %d", m, sz, Ksub * dsub);
                s.get(&cb[(size_t)m * Ksub * dsub], (size_t)Ksub * dsub * 4, "codebook data");
            }
            ncb.reserve(cb.size() * 4, c->stream, 0);
            c->h2d(ncb.p, cb.data(), cb.size() * 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        std::vector<uint8_t> codes; std::vector<uint32_t> rid; std::vector<int32_t> rl;
        const uint32_t outer = s.u32(ivf ? "list count" : "vector count");
        if (ivf && (int)outer != nlist) COMET_FAIL(COMET_ERR_FORMAT, "list count %u does not match nlist %d", outer, nlist);
        const uint32_t nl_iter = ivf ? outer : 1;
        for (uint32_t l = 0; l < nl_iter; l++) {
            const uint32_t len = ivf ? s.u32("list size") : outer;
            for (uint32_t j = 0; j < len; j++) {
                rid.push_back(s.u32("vector ID")); rl.push_back((int32_t)l);
                const size_t o = codes.size(); codes.resize(o + (size_t)M4 * 4, 0);
                s.get(&codes[o], (size_t)M, "vector code");
            }
        }
        const std::vector<uint32_t> del = read_bitmap(s);
        if (!rid.empty()) {
            ncodes.reserve(codes.size(), c->stream, 0);
            c->h2d(ncodes.p, codes.data(), codes.size());
            HIP_CHECK(hipStreamSynchronize(c->stream));
            nlay.append(rid.data(), rl.data(), (int64_t)rid.size());
        }
        // commit
        trained = tr;
        forget_placement();       // host-side training state, not in the stream: stale counts of an earlier Train must not outlive it (every loader falls back to l % world)
        std::swap(centroids.p, ncent.p); std::swap(centroids.cap, ncent.cap);
        std::swap(codebooks.p, ncb.p); std::swap(codebooks.cap, ncb.cap);
        std::swap(codes_arr.p, ncodes.p); std::swap(codes_arr.cap, ncodes.cap);
        lay.ids.swap(nlay.ids); lay.list_of.swap(nlay.list_of); lay.id_count.swap(nlay.id_count); lay.n = nlay.n; lay.dirty = true;
        il_dirty = true;
        deleted.clear(); deleted.insert(del.begin(), del.end()); deleted_dirty = true;
    }
    bool get_stat(const char* name, double* out) const override {
        std::string k(name);
        if (k == "max_list_len") { const_cast<ListLayout&>(lay).compile(c); *out = (double)lay.max_len; return true; }
        if (k == "adc_auto_one_stage") { *out = auto_one ? 1.0 : 0.0; return true; }        // adaptive staging: the searches between two samples run single-stage
        if (k == "adc_auto_searches") { *out = (double)auto_n; return true; }
        if (k.rfind("adc_", 0) == 0) {
            // cumulative counters of the fused ADC search. Two-stage search: (query, list) pairs the lower bound left / all pairs behind the nearest
            // lists. What the scan launches were given to move and score: code bytes (every item's blocks, once per duo), table bytes (every item
            // streams its duo's table), candidates (per query), searches. adc_stats_reset zeroes them.
            int32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (adc_stats.p) { c->d2h(h, adc_stats.p, 32); HIP_CHECK(hipStreamSynchronize(c->stream)); }
            const int KL = std::min(Ksub, 256);
            if (k == "adc_pairs_alive") *out = (double)h[0];
            else if (k == "adc_pairs_behind_nearest") *out = (double)h[1];
            else if (k == "adc_code_bytes") *out = (double)h[2] * 64.0 * M4 * 4.0;
            else if (k == "adc_table_bytes") *out = (double)h[3] * 2.0 * M * KL * 4.0;
            else if (k == "adc_candidates") *out = (double)h[4] * 64.0;
            else if (k == "adc_searches") *out = (double)h[5];
            else if (k == "adc_norm_bound_kills") *out = (double)h[6];        // pairs the table-free bound removed before the walk over the subspaces
            else if (k == "adc_stats_on") { stats_on = true; *out = 1.0; }
            else if (k == "adc_stats_off") { stats_on = false; *out = 0.0; }
            else if (k == "adc_stats_reset") { if (adc_stats.p) { HIP_CHECK(hipMemsetAsync(adc_stats.p, 0, 32, c->stream)); HIP_CHECK(hipStreamSynchronize(c->stream)); } *out = 0.0; }
            else return false;
            return true;
        }
        return false;
    }
    int64_t list_size(int l) const override { const_cast<ListLayout&>(lay).compile(c); return (l >= 0 && l < nlist) ? lay.len_h[l] : 0; }
    void list_read(int l, uint32_t* oids, uint8_t* ocodes, float*) const override {
        auto& L = const_cast<ListLayout&>(lay); L.compile(c);
        if (l < 0 || l >= nlist) COMET_FAIL(COMET_ERR_INVALID_ARG, "list out of range");
        const int len = L.len_h[l];
        std::vector<uint8_t> all;
        if (ocodes && L.n > 0) { all.resize((size_t)L.n * M4 * 4); c->d2h(all.data(), codes_arr.p, all.size()); HIP_CHECK(hipStreamSynchronize(c->stream)); }
        for (int j = 0; j < len; j++) {
            uint32_t r = L.row_of_slot_h[L.base_h[l] + j];
            if (oids) oids[j] = L.ids[r];
            if (ocodes) std::copy(all.begin() + (size_t)r * M4 * 4, all.begin() + (size_t)r * M4 * 4 + M, ocodes + (size_t)j * M);
        }
    }
};

static comet_index* make_pq_family(Ctx* c, bool ivf, int dim, int metric, int nlist, int M, int nbits) {
    auto* f = new PQFamilyIndex();
    f->c = c; f->kind = ivf ? COMET_KIND_IVFPQ : COMET_KIND_PQ; f->dim = dim; f->ld = padded_dim(dim); f->metric = metric;
    f->ivf = ivf; f->nlist = ivf ? nlist : 1; f->M = M; f->nbits = nbits; f->Ksub = 1 << nbits; f->dsub = dim / M; f->M4 = (M + 3) / 4;
    f->lay.nlist = f->nlist; f->lay.align = 64;
    return f;
}
comet_index* make_pq(Ctx* c, int dim, int metric, int M, int nbits) { return make_pq_family(c, false, dim, metric, 1, M, nbits); }
comet_index* make_ivfpq(Ctx* c, int dim, int metric, int nlist, int M, int nbits) { return make_pq_family(c, true, dim, metric, nlist, M, nbits); }

}  // namespace comet

// ------------------------------------------------------------------------------------------------
// C ABI: k-means, nearest centroid, multi-GPU merge
// ------------------------------------------------------------------------------------------------
using namespace comet;

extern "C" {

int comet_kmeans(comet_ctx* c, const float* vecs, int64_t n, int d, int k, int metric, int max_iter, float* out_centroids,
                 int32_t* out_assign, int* out_k) {
    return guarded([&] {
        if (out_k) *out_k = 0;
        if (metric < COMET_L2 || metric > COMET_COSINE) COMET_FAIL(COMET_ERR_UNKNOWN_METRIC, "unknown distance kind");
        if (n <= 0 || k <= 0 || d <= 0) return (int)COMET_OK;   // (nil, nil) clustering.go:123-129
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->scratch_reset();
        const int ld = padded_dim(d);
        float* raw = c->salloc<float>((size_t)n * d);
        c->h2d(raw, vecs, (size_t)n * d * sizeof(float));
        float* V = c->salloc<float>((size_t)n * ld);
        launch_ingest_rows(c, COMET_L2SQ, raw, n, d, V, ld, nullptr);
        const int ke = (int)std::min<int64_t>(k, n);
        float* cent = c->salloc<float>((size_t)ke * ld);
        int32_t* assign = c->salloc<int32_t>(n);
        const int kk = kmeans_device(c, metric, V, n, ld, k, max_iter, cent, assign);
        float* dense = c->salloc<float>((size_t)ke * d);
        launch_unpad_rows(c, cent, kk, ld, dense, d);
        c->d2h(out_centroids, dense, (size_t)kk * d * sizeof(float));
        c->d2h(out_assign, assign, (size_t)n * sizeof(int32_t));
        c->sync();
        if (out_k) *out_k = kk;
        return (int)COMET_OK;
    });
}

int comet_nearest_centroid(comet_ctx* c, const float* vecs, int64_t n, int d, const float* centroids, int k, int metric, int32_t* out_index) {
    return guarded([&] {
        if (metric < COMET_L2 || metric > COMET_COSINE) COMET_FAIL(COMET_ERR_UNKNOWN_METRIC, "unknown distance kind");
        if (n <= 0) return (int)COMET_OK;
        if (k <= 0 || d <= 0) { for (int64_t i = 0; i < n; i++) out_index[i] = 0; return (int)COMET_OK; }
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->scratch_reset();
        const int ld = padded_dim(d);
        float* raw = c->salloc<float>((size_t)n * d);
        float* rawc = c->salloc<float>((size_t)k * d);
        c->h2d(raw, vecs, (size_t)n * d * sizeof(float));
        c->h2d(rawc, centroids, (size_t)k * d * sizeof(float));
        float* V = c->salloc<float>((size_t)n * ld);
        float* C = c->salloc<float>((size_t)k * ld);
        launch_ingest_rows(c, COMET_L2SQ, raw, n, d, V, ld, nullptr);
        launch_ingest_rows(c, COMET_L2SQ, rawc, k, d, C, ld, nullptr);
        int32_t* a = c->salloc<int32_t>(n);
        assign_nearest(c, metric, V, n, ld, C, k, a);
        c->d2h(out_index, a, (size_t)n * sizeof(int32_t));
        c->sync();
        return (int)COMET_OK;
    });
}

int comet_merge_topk_dev(comet_ctx* c, const uint32_t* ids_dev, const float* scores_dev, const int32_t* counts_dev, int32_t R, int32_t B,
                         int32_t k_cap, int32_t k, uint32_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev) {
    return guarded([&] {
        if (R <= 0 || B < 0 || k_cap <= 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "bad merge shape");
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        launch_merge_topk(c, ids_dev, scores_dev, counts_dev, R, B, k_cap, k, out_ids_dev, out_scores_dev, out_counts_dev);
        return (int)COMET_OK;
    });
}

/* one packed block per shard: [B*k_cap ids | B*k_cap scores | B counts] (32-bit words), blocks `block_words` apart —
 * the layout a single all-gather of per-rank result blocks produces */
int comet_merge_topk_packed_dev(comet_ctx* c, const uint32_t* packed_dev, int64_t block_words, int32_t R, int32_t B, int32_t k_cap, int32_t k,
                                uint32_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev) {
    return guarded([&] {
        if (R <= 0 || B < 0 || k_cap <= 0 || block_words < (int64_t)2 * B * k_cap + B) COMET_FAIL(COMET_ERR_INVALID_ARG, "bad packed merge shape");
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind();
        const uint32_t* ids = packed_dev;
        const float* scores = reinterpret_cast<const float*>(packed_dev + (size_t)B * k_cap);
        const int32_t* counts = reinterpret_cast<const int32_t*>(packed_dev + (size_t)2 * B * k_cap);
        launch_merge_topk(c, ids, scores, counts, R, B, k_cap, k, out_ids_dev, out_scores_dev, out_counts_dev, block_words, block_words);
        return (int)COMET_OK;
    });
}

}  // extern "C"
