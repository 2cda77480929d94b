// index_flat.hip — FlatIndex on the GPU (reference: flat_index.go, flat_index_search.go) plus the
// helpers every index kind shares (query preprocessing, position -> id finalisation).
#include "index.hpp"

#include <cmath>

namespace comet {

// ------------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------------
void prepare_queries(Ctx* c, int metric, const float* queries_dev, int B, int dim, int ld, float** Qp, int32_t** zflag, bool may_alias) {
    if (may_alias && metric != COMET_COSINE && dim == ld && ((uintptr_t)queries_dev & 15) == 0) {
        // Preprocess is the identity for the L2 family (distance.go:138-147,182-191) and the rows are already padded: search the
        // caller's buffer in place — one launch (and its gap) less in a chain of short kernels; no query can fail (no zero flag)
        *Qp = const_cast<float*>(queries_dev); *zflag = nullptr;
        return;
    }
    *Qp = c->salloc<float>((size_t)B * ld);
    *zflag = c->salloc<int32_t>(B);
    // Distance.Preprocess(query): normalised copy for cosine, unchanged for L2 (flat_index_search.go:236)
    launch_ingest_rows(c, metric, queries_dev, B, dim, *Qp, ld, *zflag);
}

__global__ __launch_bounds__(256) void finalize_kernel(const unsigned* __restrict__ ids_table, const unsigned* __restrict__ pos,
                                                       int B, int k_cap, const int* __restrict__ zflag,
                                                       unsigned* __restrict__ out_ids, int* __restrict__ counts) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * k_cap;
    if (i < total) {
        unsigned p = pos[i];
        out_ids[i] = (p == 0xFFFFFFFFu) ? 0u : (ids_table ? ids_table[p] : p);
    }
    if (i < B && zflag && zflag[i]) counts[i] = -(int)COMET_ERR_ZERO_VECTOR;
}
void launch_finalize(Ctx* c, const uint32_t* ids_table, const uint32_t* pos, int B, int k_cap, const int32_t* zflag,
                     uint32_t* out_ids, int32_t* counts) {
    long total = std::max<long>((long)B * k_cap, B);
    finalize_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, c->stream>>>(ids_table, pos, B, k_cap, zflag, out_ids, counts);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// FlatIndex
// ------------------------------------------------------------------------------------------------
struct FlatIndex : comet_index {
    DevBuf X;        // n x ld fp32, preprocessed rows (idx.vectors, flat_index.go:82)
    DevBuf ids_dev;  // n uint32
    std::vector<uint32_t> ids;  // host mirror of the ids (Remove / lookup)
    std::unordered_map<uint32_t, int> id_count;
    int64_t n = 0;
    // fp16 shadow for the MFMA fast path (kernels_fast.hip): rows, squared norms, magnitude statistics
    int ldh = 0;
    DevBuf Xh, rn, stats_dev;
    float xmax_abs = 0.0f, xmax_norm2 = 0.0f;
    // int8 shadow of the scan tiles: codes in the fp16 shadow's tiled layout, one scale per tile,
    // the largest quantisation-residual norm of any row (squared). The slice falls back to the fp16 shadow while int8 screening is
    // too coarse for the data: a search whose int8 slices overflowed or proposed more than kI8MaxCand candidates per query
    // switches the index to fp16 for a while (doubling back-off) — results are identical either way, only the time differs.
    int ld8 = 0;
    DevBuf X8, sx;
    float xmax_err2 = 0.0f;
    static constexpr int64_t kI8MaxCand = 1536;
    int i8_policy = [] { const char* e = getenv("COMET_FLAT_I8"); return e ? atoi(e) : -1; }();   // 0 never, 1 always, otherwise adaptive
    int64_t n_searches = 0, i8_resume_at = 0, st_i8_slices = 0, st_i8_backoffs = 0; int i8_strikes = 0;
    bool i8_usable(int bn, int64_t keff) const {
        if (i8_policy == 0 || !prep_queries_i8_ok(dim) || !std::isfinite(xmax_err2)) return false;
        if (i8_policy == 1) return true;
        // the int8 screen saves ~0.15-0.2 ns of scan per row and costs the post stage the extra candidates of its wider bound (measured at
        // d 768, B 256: +0.035 ms at K 100, +0.012 ms at K 10): int8 beats fp16 from ~190 k rows at K 100, below 125 k rows at K 10
        // (62.5 k / 125 k / 250 k / 500 k rows, K 100: fp16 0.117 / 0.146 / 0.210 / 0.328 ms per batch, int8 0.149 / 0.160 / 0.191 / 0.256)
        const int64_t need = bn > 64 ? 60000 + 1500 * keff : 30000 + 500 * keff;
        return n >= need && n_searches >= i8_resume_at;
    }
    // counters of the last fast-path search (bench / tests)
    int64_t st_candidates = 0, st_overflows = 0, st_expansions = 0, st_fast_queries = 0, st_strict_queries = 0;
    // Deferred verification of fast-path searches: the candidate-overflow flags of a search are copied to pinned host
    // memory asynchronously; search_finish() waits for the search's event, and re-runs the (rare) overflowed queries on
    // the strict kernels. This keeps the stream busy across batches (no host round trip inside a search).
    struct Pending {
        bool active = false; uint64_t ticket = 0; hipEvent_t ev = nullptr;
        int B = 0, k_cap = 0, nfast_slices = 0; uint64_t i8_mask = 0;   // bit sl: slice sl was screened on the int8 shadow
        const float* queries = nullptr; comet_search_params p{}; std::vector<uint32_t> flt;
        uint32_t* out_ids = nullptr; float* out_scores = nullptr; int32_t* out_counts = nullptr;
        int32_t* flags = nullptr;   // pinned: per fast slice [256 overflow flags | 4 stats]
        int32_t* dflags = nullptr;  // the same slices in HBM: written by the post stage, copied to `flags` on the copy stream
        hipEvent_t ev_post = nullptr;
    };
    hipStream_t copy_stream = nullptr;   // the flag copy runs beside the next batch's kernels instead of between them
    static constexpr int kRing = 8, kSliceInts = 260, kMaxSlices = 64;
    Pending ring[kRing];
    uint64_t next_ticket = 1;
    ~FlatIndex() override {
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);      // nothing of this index is in flight on its private stream when its events and pinned slots go
        for (auto& r : ring) { if (r.ev) (void)hipEventDestroy(r.ev); if (r.ev_post) (void)hipEventDestroy(r.ev_post); if (r.flags) (void)hipHostFree(r.flags); if (r.dflags) (void)hipFree(r.dflags); }
        // (the stream belongs to the context since round 5: comet_ctx_create)
    }

    int64_t size() const override { return n; }
    int max_lanes() const override { return 2; }        // a search writes its own ring slot and scratch only
    bool contains_id(uint32_t id) const override { return id_count.count(id) != 0; }
    std::unordered_map<uint32_t, int64_t> first_row; bool first_row_dirty = true;
    int64_t row_of_id(uint32_t id) override {
        if (first_row_dirty) { first_row.clear(); for (int64_t i = n - 1; i >= 0; i--) first_row[ids[i]] = i; first_row_dirty = false; }
        auto it = first_row.find(id); return it == first_row.end() ? -1 : it->second;
    }
    const float* rows_dev() const override { return X.as<float>(); }
    bool raw_ingest = false;   // ReadFrom: stored vectors are already preprocessed (flat_index.go:592 keeps them as read)

    // FlatIndex.Add flat_index.go:170-186, batched.
    int64_t add_dev(const uint32_t* ids_d, const uint32_t* ids_h, const float* vecs_dev, int64_t m, int64_t* zero_at,
                    float* normalized_dev) override {
        *zero_at = -1;
        if (m <= 0) return 0;
        X.reserve((size_t)(n + m) * ld * sizeof(float), c->stream, (size_t)n * ld * sizeof(float));
        ids_dev.reserve((size_t)(n + m) * 4, c->stream, (size_t)n * 4);
        float* dst = X.as<float>() + (size_t)n * ld;
        int32_t* zf = c->salloc<int32_t>(m);
        const int im = raw_ingest ? (int)COMET_L2SQ : metric;
        launch_ingest_rows(c, im, vecs_dev, m, dim, dst, ld, zf);
        int64_t added = m;
        if (im == COMET_COSINE) {   // ErrZeroVector stops the batch at the first offending vector
            std::vector<int32_t> h(m);
            c->d2h(h.data(), zf, m * sizeof(int32_t));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < m; i++) if (h[i]) { *zero_at = i; added = i; break; }
        }
        if (added > 0) {
            if (ids_d) c->d2d(ids_dev.as<uint32_t>() + n, ids_d, added * 4);
            else c->h2d(ids_dev.as<uint32_t>() + n, ids_h, added * 4);
            if (normalized_dev) launch_unpad_rows(c, dst, added, ld, normalized_dev, dim);
            // half-precision shadow of the new rows + their squared norms and magnitude statistics
            {   // tiled shadow: whole 256-row tiles, new tiles zero-initialised (padding rows must read as zeros)
                const size_t tile_bytes = (size_t)256 * ldh * 2;
                const size_t old_tiles = (size_t)ceil_div(n, 256), new_tiles = (size_t)ceil_div(n + added, 256);
                Xh.reserve(new_tiles * tile_bytes, c->stream, old_tiles * tile_bytes);
                if (new_tiles > old_tiles) c->zero((char*)Xh.p + old_tiles * tile_bytes, (new_tiles - old_tiles) * tile_bytes);
            }
            rn.reserve((size_t)(n + added) * 4, c->stream, (size_t)n * 4);
            stats_dev.reserve(16, c->stream, 0);
            c->zero(stats_dev.p, 16);
            launch_to_half_rows(c, dst, added, ld, Xh.p, ldh, n, rn.as<float>() + n, stats_dev.as<uint32_t>());
            if (i8_policy != 0 && prep_queries_i8_ok(dim)) {
                const size_t tile_bytes = (size_t)256 * ld8;
                const size_t old_tiles = (size_t)ceil_div(n, 256), new_tiles = (size_t)ceil_div(n + added, 256);
                X8.reserve(new_tiles * tile_bytes, c->stream, old_tiles * tile_bytes);
                if (new_tiles > old_tiles) c->zero((char*)X8.p + old_tiles * tile_bytes, (new_tiles - old_tiles) * tile_bytes);
                sx.reserve(new_tiles * 4, c->stream, old_tiles * 4);
                // the tile the batch starts in is quantised again as a whole: its scale follows the largest component of all its rows
                launch_to_i8_tiles(c, X.as<float>(), n + added, ld, X8.p, ld8, n / 256, sx.as<float>(), stats_dev.as<uint32_t>());
            }
            uint32_t hs[4] = {0, 0, 0, 0};
            c->d2h(hs, stats_dev.p, 16);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            float fa, fn, fe; std::memcpy(&fa, &hs[0], 4); std::memcpy(&fn, &hs[1], 4); std::memcpy(&fe, &hs[2], 4);
            if (!(fa <= xmax_abs)) xmax_abs = fa;       // NaN-propagating max
            if (!(fn <= xmax_norm2)) xmax_norm2 = fn;
            if (!(fe <= xmax_err2)) xmax_err2 = fe;
            ids.insert(ids.end(), ids_h, ids_h + added);
            for (int64_t i = 0; i < added; i++) id_count[ids_h[i]]++;
            n += added; first_row_dirty = true;
        }
        return added;
    }

    // FlatIndex.Flush flat_index.go:268-296: compact away soft-deleted rows.
    void flush() override {
        if (deleted.empty()) return;
        std::vector<int64_t> keep;
        for (int64_t i = 0; i < n; i++) if (!deleted.count(ids[i])) keep.push_back(i);
        DevBuf nx, nid;
        size_t nk = keep.size();
        nx.reserve(std::max<size_t>(1, nk) * ld * sizeof(float), c->stream, 0);
        nid.reserve(std::max<size_t>(1, nk) * 4, c->stream, 0);
        std::vector<uint32_t> nids(nk);
        // copy surviving runs (device-to-device, contiguous runs coalesced)
        size_t i = 0;
        while (i < nk) {
            size_t j = i;
            while (j + 1 < nk && keep[j + 1] == keep[j] + 1) j++;
            size_t run = j - i + 1;
            c->d2d(nx.as<float>() + i * ld, X.as<float>() + (size_t)keep[i] * ld, run * ld * sizeof(float));
            for (size_t t = 0; t < run; t++) nids[i + t] = ids[keep[i] + t];
            i = j + 1;
        }
        c->h2d(nid.p, nids.data(), nk * 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        std::swap(X.p, nx.p); std::swap(X.cap, nx.cap);
        std::swap(ids_dev.p, nid.p); std::swap(ids_dev.cap, nid.cap);
        {   // rebuild the fp16 shadow from the compacted rows
            const size_t tiles = std::max<size_t>(1, (size_t)ceil_div((int64_t)nk, 256));
            Xh.reserve(tiles * 256 * ldh * 2, c->stream, 0);
            c->zero(Xh.p, tiles * 256 * ldh * 2);
            rn.reserve(std::max<size_t>(1, nk) * 4, c->stream, 0);
            launch_to_half_rows(c, X.as<float>(), (int64_t)nk, ld, Xh.p, ldh, 0, rn.as<float>(), nullptr);
            if (X8.p) {   // and the int8 one: compaction regroups the rows into other tiles, i.e. other scales — the residual maximum is measured anew
                X8.reserve(tiles * 256 * ld8, c->stream, 0);
                c->zero(X8.p, tiles * 256 * ld8);
                sx.reserve(tiles * 4, c->stream, 0);
                stats_dev.reserve(16, c->stream, 0);
                c->zero(stats_dev.p, 16);
                launch_to_i8_tiles(c, X.as<float>(), (int64_t)nk, ld, X8.p, ld8, 0, sx.as<float>(), stats_dev.as<uint32_t>());
                uint32_t hs[4] = {0, 0, 0, 0};
                c->d2h(hs, stats_dev.p, 16);
                HIP_CHECK(hipStreamSynchronize(c->stream));
                std::memcpy(&xmax_err2, &hs[2], 4);
            }
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        ids.swap(nids); n = (int64_t)nk; first_row_dirty = true;
        id_count.clear(); for (auto id : ids) id_count[id]++;
        deleted.clear(); deleted_dirty = true;
    }

    // exact-arithmetic scan + radix top-K for `bn` prepared queries (the strict path; also the fallback)
    void search_strict(const float* Qp, int bn, const comet_search_params& p, const uint8_t* elig, uint32_t* pos, float* out_scores,
                       int32_t* out_counts, int k_cap) {
        ScratchMark sm(c);
        const int64_t ldD = round_up(n, 16);
        const size_t budget = (size_t)2 << 30;   // bound the distance-matrix scratch: slices of queries
        int qb = (int)std::max<int64_t>(1, std::min<int64_t>(bn, (int64_t)(budget / ((size_t)ldD * sizeof(float)))));
        if (qb >= 16) qb = qb / 16 * 16;
        float* D = c->salloc<float>((size_t)qb * ldD);
        for (int b0 = 0; b0 < bn; b0 += qb) {
            const int m = std::min(qb, bn - b0);
            ScratchMark sm2(c);
            launch_dist_exact(c, metric, X.as<float>(), n, ld, Qp + (size_t)b0 * ld, m, D, ldD, elig);
            launch_select_topk(c, D, ldD, m, n, nullptr, p.threshold, p.k, pos + (size_t)b0 * k_cap,
                               out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap);
        }
        st_strict_queries += bn;
    }

    bool fast_usable(int B, const comet_search_params& p) const {
        if (p.mode == 1) return false;
        const int64_t keff = (p.k <= 0 || p.k > n) ? n : p.k;
        const bool ok = std::isfinite(xmax_abs) && xmax_abs <= 60000.0f && std::isfinite(xmax_norm2) && keff <= 1024 && n >= 1;
        if (p.mode == 2) return ok;
        // auto: the unit-top-2 proposal needs several times more 128-row units than requested results to stay selective
        // (measured at K = 100, B = 256: 125k rows 0.20 ms fast vs 1.19 ms strict, 500k rows 0.38 vs 4.7 — the row shards of a
        // multi-GPU run live in this regime). The batch size does not matter: even ONE query is faster through the half-
        // precision shadow (1.5 GB streamed instead of 3 GB; 1M x 768: 0.61 ms vs 0.72; 100k rows: 0.14 vs 0.22).
        (void)B;
        return ok && n >= (int64_t)128 * 4 * keff;
    }

    // MFMA fast path for up to 256 prepared queries; writes the final ids / scores / counts of the slice.
    // raw_queries != nullptr: the queries are not preprocessed yet — one fused launch does Distance.Preprocess and the fp16 side.
    void search_fast(const float* Qp, int32_t* zflag, int bn, const comet_search_params& p, const uint8_t* elig, uint32_t* out_ids,
                     float* out_scores, int32_t* out_counts, int k_cap, Pending* pend, const float* raw_queries = nullptr) {
        ScratchMark sm(c);
        // key units: the scan emits, per query, 2 keys + 1 bound for every 128-row unit (64-row units on the narrow tile for <= 64 queries
        // and where 128-row units would often be expanded: small indexes / large k)
        const bool i8 = i8_usable(bn, (p.k <= 0 || p.k > n) ? n : p.k) && X8.p != nullptr;
        const int unit_rows = flat_fast_unit_rows(bn, n, p.k, i8 ? ld8 / 2 : ldh);
        const int64_t n_tiles = ceil_div(n, flat_fast_tile_rows()) * (flat_fast_tile_rows() / unit_rows);
        const int64_t ldS = round_up(2 * n_tiles, 16), ldB = round_up(n_tiles, 16);
        const int NB = flat_fast_batch();
        void* Qh = i8 ? c->scratch_alloc((size_t)NB * ld8 * 2) : c->scratch_alloc((size_t)NB * ldh * 2 * 2);   // row-major copy + MFMA-fragment-ordered copy (fp16 values / int8 codes)
        void* Q8R = i8 ? (char*)Qh + (size_t)NB * ld8 : nullptr;
        float* qn = c->salloc<float>(NB);
        float* sqv = i8 ? c->salloc<float>(NB) : nullptr;
        float* err = c->salloc<float>(NB);
        int32_t* flags = pend->dflags + (size_t)pend->nfast_slices * kSliceInts;     // [256 overflow flags | 4 stats] of this slice
        int32_t* ovf = flags; int32_t* st = flags + 256;
        const int fmode = metric == COMET_COSINE ? 0 : 1;
        const float xn2 = metric == COMET_COSINE ? 1.0002f : xmax_norm2;
        if (i8) launch_prep_queries_i8(c, metric, raw_queries, bn, dim, const_cast<float*>(Qp), ld, raw_queries ? zflag : nullptr, Qh, bn <= 64 ? Q8R : nullptr, ld8, sqv, qn, err, fmode, xn2, std::sqrt(xmax_err2), st);
        else if (raw_queries) launch_prep_queries_fused(c, metric, raw_queries, bn, dim, const_cast<float*>(Qp), ld, zflag, Qh, ldh, qn, err, fmode, xn2, st);
        else launch_prep_queries_fast(c, Qp, bn, ld, dim, Qh, ldh, qn, err, fmode, xn2, st);
        float* S0 = c->salloc<float>((size_t)NB * ldS);
        float* bound = c->salloc<float>((size_t)NB * ldB);
        if (i8) { launch_flat_scan_i8(c, fmode, X8.p, n, ld8, Qh, Q8R, bn, rn.as<float>(), qn, sx.as<float>(), sqv, elig, S0, ldS, bound, ldB, unit_rows); pend->i8_mask |= 1ull << pend->nfast_slices; }
        else launch_flat_scan_f16(c, fmode, Xh.p, n, ldh, Qh, bn, rn.as<float>(), qn, elig, S0, ldS, bound, ldB, unit_rows);
        // kappa: exact K-th smallest emitted key per query
        const int64_t keff = (p.k <= 0 || p.k > n) ? n : p.k;
        const int Kq = (int)std::min<int64_t>(keff, 2 * n_tiles);
        // if the unit keys cannot even supply K values (tiny index / huge K) tau = +inf: every unit is expanded
        launch_flat_post(c, metric, S0, ldS, bound, ldB, n_tiles, unit_rows, n, elig, err, p.k, (Kq == keff) ? Kq : 0, p.threshold, X.as<float>(), ld, Qp, bn,
                         ids_dev.as<uint32_t>(), zflag, out_ids, out_scores, out_counts, k_cap, ovf, st);
        pend->nfast_slices++;                        // search_begin copies the slices to pinned host memory on the copy stream
    }

    // flatIndexSearch.searchSingleQuery flat_index_search.go:221-294 for B queries at once (enqueue only).
    void search_core(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                     int32_t* out_counts, int k_cap, Pending* pend) {
        float* Qp; int32_t* zflag;
        const int NB = flat_fast_batch();
        // one fast slice: Distance.Preprocess is fused into the fast path's query kernel (search_fast with the raw queries)
        const bool fuse_prep = pend && n > 0 && B <= NB && fast_usable(B, p) && prep_queries_fused_ok(dim);
        if (fuse_prep) { Qp = c->salloc<float>((size_t)B * ld); zflag = c->salloc<int32_t>(B); }
        else prepare_queries(c, metric, queries_dev, B, dim, ld, &Qp, &zflag);
        if (n == 0) {   // empty index: zero results (sanitizeK(k, 0) == 0)
            uint32_t* pos = c->salloc<uint32_t>((size_t)B * k_cap);
            launch_select_topk(c, nullptr, 0, B, 0, nullptr, 0.0f, p.k, pos, out_scores, out_counts, k_cap);
            launch_finalize(c, nullptr, pos, B, k_cap, zflag, out_ids, out_counts);
            return;
        }
        // eligibility: soft deletes (flat_index_search.go:256) and WithDocumentIDs filter (:261)
        const uint8_t* elig = nullptr;
        int nf = 0;
        const uint32_t* flt = filter_sorted_scratch(p, &nf);
        const uint32_t* del = deleted.empty() ? nullptr : deleted_sorted_dev();
        const int nd = deleted.empty() ? 0 : n_deleted_dev;
        if (nd > 0 || nf > 0) {
            uint8_t* e = c->salloc<uint8_t>(n);
            launch_build_elig(c, ids_dev.as<uint32_t>(), n, del, nd, flt, nf, e);
            elig = e;
        }
        if (p.mode == 2 && !fast_usable(B, p)) COMET_FAIL(COMET_ERR_UNSUPPORTED, "fast path unavailable for this index / k (values beyond fp16 range or k > 1024)");
        if (pend && fast_usable(B, p) && ceil_div(B, NB) <= kMaxSlices) {
            for (int b0 = 0; b0 < B; b0 += NB) {
                const int bn = std::min(NB, B - b0);
                search_fast(Qp + (size_t)b0 * ld, zflag + b0, bn, p, elig, out_ids + (size_t)b0 * k_cap, out_scores + (size_t)b0 * k_cap, out_counts + b0, k_cap, pend,
                            fuse_prep ? queries_dev : nullptr);
            }
            return;                                  // the post stage wrote ids and statuses itself
        }
        uint32_t* pos = c->salloc<uint32_t>((size_t)B * k_cap);
        search_strict(Qp, B, p, elig, pos, out_scores, out_counts, k_cap);
        launch_finalize(c, ids_dev.as<uint32_t>(), pos, B, k_cap, zflag, out_ids, out_counts);
    }

    uint64_t search_begin(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                          int32_t* out_counts, int k_cap) override {
        // pick a free ring slot (retire the oldest if all are in use)
        Pending* slot = nullptr;
        for (auto& r : ring) if (!r.active) { slot = &r; break; }
        if (!slot) {
            Pending* oldest = &ring[0];
            for (auto& r : ring) if (r.ticket < oldest->ticket) oldest = &r;
            search_finish(oldest->ticket);
            slot = oldest;
        }
        if (!slot->ev) HIP_CHECK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
        if (!slot->flags) HIP_CHECK(hipHostMalloc((void**)&slot->flags, sizeof(int32_t) * kSliceInts * kMaxSlices, hipHostMallocDefault));
        if (!slot->dflags) HIP_CHECK(hipMalloc((void**)&slot->dflags, sizeof(int32_t) * kSliceInts * kMaxSlices));
        if (!slot->ev_post) HIP_CHECK(hipEventCreateWithFlags(&slot->ev_post, hipEventDisableTiming));
        copy_stream = c->stream;      // the flag copy follows the search on its own lane (round 5; a private stream until then: see below)
        slot->ticket = next_ticket++; slot->B = B; slot->k_cap = k_cap; slot->nfast_slices = 0; slot->i8_mask = 0; slot->queries = queries_dev;
        n_searches++;
        slot->p = p; slot->flt.clear();
        if (p.filter_ids && p.n_filter > 0) { slot->flt.assign(p.filter_ids, p.filter_ids + p.n_filter); slot->p.filter_ids = slot->flt.data(); }
        slot->out_ids = out_ids; slot->out_scores = out_scores; slot->out_counts = out_counts;
        st_candidates = st_overflows = st_expansions = st_fast_queries = st_strict_queries = st_i8_slices = 0;
        search_core(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap, slot);
        if (slot->nfast_slices > 0) {
            // overflow flags + statistics go to pinned host memory behind the search, on its lane's stream; search_finish() waits for the copy and therefore
            // for everything before it. (Rounds 2-4 sent the copy to a private stream so that it left the chain of the next batch; with the searches alternating
            // between lanes the next batch's chain is on another stream anyway, and a fifth stream beside the four lanes shares a hardware queue with one of them:
            // Flat single stream 855 -> 875 k q/s, IVF 1.09 -> 1.13 M with the copy inline, comet_ctx_create.)
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
        if (!slot) return false;     // already finished (or never deferred)
        HIP_CHECK(hipEventSynchronize(slot->ev));
        slot->active = false;
        const int NB = flat_fast_batch();
        std::vector<int> redo;
        for (int sl = 0; sl < slot->nfast_slices; sl++) {
            const int32_t* hf = slot->flags + (size_t)sl * kSliceInts;
            const int bn = std::min(NB, slot->B - sl * NB);
            st_candidates += hf[256]; st_overflows += hf[257]; st_expansions += hf[258];
            if ((slot->i8_mask >> sl) & 1ull) {
                st_i8_slices++;
                if (i8_policy != 1 && (hf[257] > 0 || hf[256] > kI8MaxCand * bn) && n_searches >= i8_resume_at) {   // too coarse for this data: fp16 for a while
                    i8_resume_at = n_searches + ((int64_t)32 << std::min(i8_strikes, 10)); i8_strikes++; st_i8_backoffs++;
                }
            }
            int nfast = bn;
            for (int q = 0; q < bn; q++) if (hf[q]) { redo.push_back(sl * NB + q); nfast--; }
            st_fast_queries += nfast;
        }
        // overflowed queries (candidate list > cap: adversarial clustering / mass ties): strict kernels, one query at a time
        comet_search_params sp = slot->p; sp.mode = 1;
        for (int q : redo) {
            ScratchMark sm(c);   // on top of whatever the arena holds (the host API keeps queries / outputs of this call there)
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
        else if (k == "max_abs") *out = (double)xmax_abs;
        else if (k == "max_norm2") *out = (double)xmax_norm2;
        else return false;
        return true;
    }
    void export_all(uint32_t* oids, int32_t* olists, uint8_t*) const override {
        if (oids) std::copy(ids.begin(), ids.end(), oids);
        if (olists) std::fill(olists, olists + n, 0);
    }

    // FlatIndex.WriteTo flat_index.go:366-470: Flush, then magic "FLAT", version, dim, distance kind, count,
    // per vector {id, dim, dim float32}, empty roaring bitmap.
    void write_to(Sink& s) override {
        flush();
        write_header(s, "FLAT", dim, metric);
        s.u32((uint32_t)n);
        const int64_t chunk = 16384;
        ScratchMark sm(c);
        float* tmp = c->salloc<float>((size_t)std::min<int64_t>(chunk, std::max<int64_t>(n, 1)) * dim);
        std::vector<float> host((size_t)std::min<int64_t>(chunk, std::max<int64_t>(n, 1)) * dim);
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t m = std::min(chunk, n - r0);
            launch_unpad_rows(c, X.as<float>() + (size_t)r0 * ld, m, ld, tmp, dim);
            c->d2h(host.data(), tmp, (size_t)m * dim * 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < m; i++) { s.u32(ids[r0 + i]); s.u32((uint32_t)dim); s.put(&host[(size_t)i * dim], (size_t)dim * 4); }
        }
        write_empty_bitmap(s);
        s.flush();
    }
    // FlatIndex.ReadFrom flat_index.go:488-614. The stream is parsed into a scratch index; this index is replaced
    // only if everything parsed (the reference assigns idx.vectors / idx.deletedNodes last, :609-611).
    void read_from(Source& s) override {
        read_header(s, "FLAT", dim, metric);
        const uint32_t count = s.u32("vector count");
        FlatIndex t; t.c = c; t.kind = kind; t.dim = dim; t.ld = ld; t.ldh = ldh; t.ld8 = ld8; t.metric = metric; t.trained = true; t.raw_ingest = true;
        const int64_t chunk = 16384;
        std::vector<float> host((size_t)std::min<int64_t>(chunk, std::max<int64_t>(count, 1)) * dim);
        std::vector<uint32_t> hid(std::min<int64_t>(chunk, std::max<int64_t>(count, 1)));
        for (int64_t r0 = 0; r0 < (int64_t)count; r0 += chunk) {
            const int64_t m = std::min<int64_t>(chunk, count - r0);
            for (int64_t i = 0; i < m; i++) {
                hid[i] = s.u32("vector ID");
                const uint32_t vd = s.u32("vector dimension");
                if ((int)vd != dim) COMET_FAIL(COMET_ERR_FORMAT, "vector %lld has dimension %u, expected %d", (long long)(r0 + i), vd, dim);   // :573
                s.get(&host[(size_t)i * dim], (size_t)dim * 4, "vector component");
            }
            ScratchMark sm(c);
            float* dv = c->salloc<float>((size_t)m * dim);
            c->h2d(dv, host.data(), (size_t)m * dim * 4);
            int64_t zero_at = -1;
            t.add_dev(nullptr, hid.data(), dv, m, &zero_at, nullptr);
        }
        const std::vector<uint32_t> del = read_bitmap(s);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        // commit
        for (auto& r : ring) if (r.active) search_finish(r.ticket);
        std::swap(X.p, t.X.p); std::swap(X.cap, t.X.cap); std::swap(ids_dev.p, t.ids_dev.p); std::swap(ids_dev.cap, t.ids_dev.cap);
        std::swap(Xh.p, t.Xh.p); std::swap(Xh.cap, t.Xh.cap); std::swap(rn.p, t.rn.p); std::swap(rn.cap, t.rn.cap);
        std::swap(X8.p, t.X8.p); std::swap(X8.cap, t.X8.cap); std::swap(sx.p, t.sx.p); std::swap(sx.cap, t.sx.cap);
        ids.swap(t.ids); id_count.swap(t.id_count); n = t.n; xmax_abs = t.xmax_abs; xmax_norm2 = t.xmax_norm2; xmax_err2 = t.xmax_err2; first_row_dirty = true;
        deleted.clear(); deleted.insert(del.begin(), del.end()); deleted_dirty = true;
    }

    void list_read(int, uint32_t* oids, uint8_t*, float* ovecs) const override {
        if (oids) std::copy(ids.begin(), ids.end(), oids);
        if (ovecs && n > 0) {
            float* tmp = c->salloc<float>((size_t)n * dim);
            launch_unpad_rows(c, X.as<float>(), n, ld, tmp, dim);
            c->d2h(ovecs, tmp, (size_t)n * dim * sizeof(float));
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
    }
};

comet_index* make_flat(Ctx* c, int dim, int metric) {
    auto* f = new FlatIndex();
    f->c = c; f->kind = COMET_KIND_FLAT; f->dim = dim; f->ld = padded_dim(dim); f->ldh = (int)round_up(dim, 64); f->ld8 = (int)round_up(dim, 256); f->metric = metric; f->trained = true;
    return f;
}

}  // namespace comet
