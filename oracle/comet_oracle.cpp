// comet_oracle.cpp — CPU restatement of wizenheimer/comet's vector-search hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product path (comet_amd/, libcomet_hip.so)
// never links, imports or calls it.
//
// What it is: a line-by-line restatement, in scalar C++ (g++ -O2 -ffp-contract=off, no fast-math), of
// the Go loops the reference runs on its CPU path, with the same float32 / float64 evaluation order
// (Go on amd64 evaluates float32 expressions in float32, never fuses a*b+c, and folds untyped
// constants exactly). The Go reference itself cannot be built in this image (no `go` toolchain, and
// its go.mod dependencies roaring/uax29/x-text/float16 are not vendored), so there is no oracle/_ref.
//
// Pinning status (see tests/test_oracle_golden.py, tests/golden/reference_kats.json):
//   PINNED by the reference's own known-answer tests: distances (distance_test.go:87-145,214-266,
//     335-387), preprocess/normalize (:417-457,:533-785), k-means fixtures (clustering_test.go:9-56,
//     104-134,203-302), nearest-centroid (:649-947), Flat search behaviour
//     (flat_index_search_test.go:10-127,348-389,490-536), Autocut (limiter_test.go:185-256),
//     aggregation (aggregation_test.go:7-115), RRF (fusion_test.go:138-201).
//   PARITY UNPINNED by the reference (it holds no numeric fixtures for them): PQ / IVFPQ distances,
//     BM25 numeric scores, HNSW graph shape (the reference draws levels from an unseeded global RNG),
//     order among exactly tied scores (the reference uses unstable sorts / Go map order). For those
//     this file *is* the specification; ties are broken canonically by scan order (stable sort).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

enum { ORC_L2 = 0, ORC_L2SQ = 1, ORC_COSINE = 2 };
enum { ORC_OK = 0, ORC_ERR_ZERO_VECTOR = -1, ORC_ERR_DIM = -2, ORC_ERR_NOT_TRAINED = -3,
       ORC_ERR_NOT_FOUND = -4, ORC_ERR_ARG = -5, ORC_ERR_ALREADY_DELETED = -6, ORC_ERR_TRAIN_DATA = -7 };

// ---------------------------------------------------------------------------------------------
// distance.go
// ---------------------------------------------------------------------------------------------

// float32(math.Sqrt(float64(x))) — distance.go:120, :258, :316
static inline float go_sqrt32(float x) { return (float)std::sqrt((double)x); }

// euclidean.Calculate distance.go:114-121 ; l2Squared.Calculate :158-165 ; cosine.Calculate :201-216
static float dist_calc(int metric, const float* a, const float* b, int d) {
    if (metric == ORC_COSINE) {
        float dot = 0.0f;
        for (int i = 0; i < d; i++) {
            float p = a[i] * b[i];
            dot = dot + p;
        }
        if (dot > 1.0f) dot = 1.0f; else if (dot < -1.0f) dot = -1.0f;
        return 1.0f - dot;
    }
    float sum = 0.0f;
    for (int i = 0; i < d; i++) {
        float diff = a[i] - b[i];
        float sq = diff * diff;
        sum = sum + sq;
    }
    return metric == ORC_L2 ? go_sqrt32(sum) : sum;
}

ORC_API float orc_distance(int metric, const float* a, const float* b, int d) { return dist_calc(metric, a, b, d); }

// CalculateBatch distance.go:123-135, :167-179, :218-239 — many queries vs one target.
ORC_API void orc_distance_batch(int metric, const float* queries, int nq, const float* target, int d, float* out) {
    for (int i = 0; i < nq; i++) out[i] = dist_calc(metric, queries + (size_t)i * d, target, d);
}

// cosine.Preprocess / PreprocessInPlace distance.go:244-290 (no-ops for L2: :138-147, :182-191).
// out may alias x. Returns ORC_ERR_ZERO_VECTOR for a zero-norm vector under cosine.
ORC_API int orc_preprocess(int metric, const float* x, int d, float* out) {
    if (metric != ORC_COSINE) { if (out != x) std::memcpy(out, x, sizeof(float) * d); return ORC_OK; }
    float sum = 0.0f;
    for (int i = 0; i < d; i++) { float p = x[i] * x[i]; sum = sum + p; }
    float norm = go_sqrt32(sum);
    if (norm == 0.0f) return ORC_ERR_ZERO_VECTOR;
    float scale = 1.0f / norm;  // `scale := 1.0 / norm` is float32 arithmetic (untyped const / float32)
    for (int i = 0; i < d; i++) out[i] = x[i] * scale;
    return ORC_OK;
}

// Norm distance.go:312-318
ORC_API float orc_norm(const float* v, int d) {
    float sum = 0.0f;
    for (int i = 0; i < d; i++) { float p = v[i] * v[i]; sum = sum + p; }
    return go_sqrt32(sum);
}
// Scale distance.go:341-347
ORC_API void orc_scale(const float* v, int d, float s, float* out) { for (int i = 0; i < d; i++) out[i] = v[i] * s; }
// Normalize / NormalizeInPlace distance.go:374-428 — zero vector returned unchanged (no error).
ORC_API void orc_normalize(const float* v, int d, float* out) {
    float norm = orc_norm(v, d);
    if (norm == 0.0f) { if (out != v) std::memcpy(out, v, sizeof(float) * d); return; }
    float scale = 1.0f / norm;
    for (int i = 0; i < d; i++) out[i] = v[i] * scale;
}

// ---------------------------------------------------------------------------------------------
// limiter.go
// ---------------------------------------------------------------------------------------------
// sanitizeK limiter.go:12-17
static inline int sanitize_k(int k, int maxr) { return (k <= 0 || k > maxr) ? maxr : k; }
ORC_API int orc_sanitize_k(int k, int maxr) { return sanitize_k(k, maxr); }

// Autocut limiter.go:81-118
ORC_API int orc_autocut(const float* y, int n, int cutoff) {
    if (n <= 1) return n;
    std::vector<float> diff(n);
    float step = 1.0f / ((float)n - 1.0f);
    for (int i = 0; i < n; i++) {
        float xv = 0.0f + (float)i * step;
        float yn = (y[i] - y[0]) / (y[n - 1] - y[0]);
        diff[i] = yn - xv;
    }
    int extrema = 0;
    for (int i = 0; i < n; i++) {
        if (i == 0) continue;
        if (i == n - 1 && n > 1) {
            // reference reads diff[i-2]; with n==2 Go would panic — n<=1 handled above, n==2: i=1, i-2=-1
            if (i - 2 < 0) continue;  // Go panics here for n==2 only if the first clause is true; diff[1]>diff[0] is false for n==2 (both 0) → short-circuit
            if (diff[i] > diff[i - 1] && diff[i] > diff[i - 2]) { extrema++; if (extrema >= cutoff) return i; }
        } else {
            if (diff[i] > diff[i - 1] && diff[i] > diff[i + 1]) { extrema++; if (extrema >= cutoff) return i; }
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// canonical ordering helper: the reference sorts candidate lists with sort.Slice (unstable) by
// distance only (flat_index_search.go:277 etc.). Order among equal distances is therefore undefined
// in the reference; the oracle (and the GPU path) break ties by scan order == stable sort.
// ---------------------------------------------------------------------------------------------
struct Cand { uint32_t id; float dist; };
static void stable_sort_asc(std::vector<Cand>& v) {
    std::stable_sort(v.begin(), v.end(), [](const Cand& a, const Cand& b) { return a.dist < b.dist; });
}
static int emit_topk(std::vector<Cand>& res, int k, uint32_t* out_ids, float* out_scores, int cap) {
    stable_sort_asc(res);
    k = sanitize_k(k, (int)res.size());
    int n = std::min(k, cap);
    for (int i = 0; i < n; i++) { out_ids[i] = res[i].id; out_scores[i] = res[i].dist; }
    return k;  // number of results the reference would return (may exceed cap)
}

struct Filter {  // document_filter.go:27-66 — nil filter (no ids) == everything eligible
    bool active = false;
    std::unordered_set<uint32_t> ids;
    Filter(const uint32_t* f, int n) { if (n > 0 && f) { active = true; ids.insert(f, f + n); } }
    bool skip(uint32_t id) const { return active && !ids.count(id); }
};

// ---------------------------------------------------------------------------------------------
// flat_index.go / flat_index_search.go
// ---------------------------------------------------------------------------------------------
struct OFlat {
    int dim, metric;
    std::vector<uint32_t> ids;
    std::vector<float> vecs;  // row-major, already preprocessed (flat_index.go:182)
    std::unordered_set<uint32_t> deleted;
};
ORC_API void* orc_flat_new(int dim, int metric) { auto* h = new OFlat(); h->dim = dim; h->metric = metric; return h; }
ORC_API void orc_flat_free(void* p) { delete (OFlat*)p; }
// FlatIndex.Add flat_index.go:170-186
ORC_API int orc_flat_add(void* p, uint32_t id, const float* v) {
    auto* h = (OFlat*)p;
    std::vector<float> tmp(h->dim);
    int rc = orc_preprocess(h->metric, v, h->dim, tmp.data());
    if (rc) return rc;
    h->ids.push_back(id);
    h->vecs.insert(h->vecs.end(), tmp.begin(), tmp.end());
    return ORC_OK;
}
// FlatIndex.Remove flat_index.go:216-249
ORC_API int orc_flat_remove(void* p, uint32_t id) {
    auto* h = (OFlat*)p;
    if (std::find(h->ids.begin(), h->ids.end(), id) == h->ids.end()) return ORC_ERR_NOT_FOUND;
    if (h->deleted.count(id)) return ORC_ERR_ALREADY_DELETED;
    h->deleted.insert(id);
    return ORC_OK;
}
// FlatIndex.Flush flat_index.go:268-296
ORC_API void orc_flat_flush(void* p) {
    auto* h = (OFlat*)p;
    if (h->deleted.empty()) return;
    std::vector<uint32_t> ids; std::vector<float> vecs;
    for (size_t i = 0; i < h->ids.size(); i++) if (!h->deleted.count(h->ids[i])) {
        ids.push_back(h->ids[i]);
        vecs.insert(vecs.end(), h->vecs.begin() + i * h->dim, h->vecs.begin() + (i + 1) * h->dim);
    }
    h->ids.swap(ids); h->vecs.swap(vecs); h->deleted.clear();
}
ORC_API int orc_flat_size(void* p) { return (int)((OFlat*)p)->ids.size(); }
ORC_API const float* orc_flat_vectors(void* p) { return ((OFlat*)p)->vecs.data(); }

// flatIndexSearch.searchSingleQuery flat_index_search.go:221-294.
// Returns the result count (>=0) or a negative ORC_ERR_*. Writes min(count, cap) rows.
ORC_API int orc_flat_search(void* p, const float* q, int k, float threshold, const uint32_t* filter, int n_filter,
                            uint32_t* out_ids, float* out_scores, int cap) {
    auto* h = (OFlat*)p;
    int n = (int)h->ids.size();
    k = sanitize_k(k, n);
    std::vector<float> pq(h->dim);
    int rc = orc_preprocess(h->metric, q, h->dim, pq.data());
    if (rc) return rc;
    Filter f(filter, n_filter);
    std::vector<Cand> res; res.reserve(n);
    for (int i = 0; i < n; i++) {
        uint32_t id = h->ids[i];
        if (h->deleted.count(id)) continue;
        if (f.skip(id)) continue;
        float dist = dist_calc(h->metric, pq.data(), &h->vecs[(size_t)i * h->dim], h->dim);
        if (threshold > 0 && dist > threshold) continue;
        res.push_back({id, dist});
    }
    return emit_topk(res, k, out_ids, out_scores, cap);
}

// ---------------------------------------------------------------------------------------------
// clustering.go
// ---------------------------------------------------------------------------------------------
// kmeansInternal clustering.go:119-243. vectors: n×d row-major. Returns effective k (k clamped to n),
// 0 for the reference's (nil,nil) cases. out_centroids must hold min(k,n)×d floats, out_assign n ints.
static int kmeans_internal(const float* vectors, int n, int d, int k, int metric, int max_iter,
                           float* centroids, int* assign) {
    if (n == 0) return 0;
    if (k <= 0) return 0;
    if (k > n) k = n;
    if (max_iter <= 0) max_iter = 20;  // DefaultMaxIter clustering.go:14
    int step = n / k; if (step == 0) step = 1;
    for (int c = 0; c < k; c++) {
        int vi = c * step; if (vi >= n) vi = n - 1;
        std::memcpy(centroids + (size_t)c * d, vectors + (size_t)vi * d, sizeof(float) * d);
    }
    for (int i = 0; i < n; i++) assign[i] = -1;  // UnassignedCluster
    std::vector<float> sums((size_t)k * d);
    std::vector<int> sizes(k);
    for (int it = 0; it < max_iter; it++) {
        bool changed = false;
        for (int vi = 0; vi < n; vi++) {
            float best = std::numeric_limits<float>::infinity();
            int bc = 0;
            for (int c = 0; c < k; c++) {
                float dist = dist_calc(metric, vectors + (size_t)vi * d, centroids + (size_t)c * d, d);
                if (dist < best) { best = dist; bc = c; }
            }
            if (assign[vi] != bc) { changed = true; assign[vi] = bc; }
        }
        if (!changed) break;
        std::fill(sums.begin(), sums.end(), 0.0f);
        std::fill(sizes.begin(), sizes.end(), 0);
        for (int vi = 0; vi < n; vi++) {
            int c = assign[vi];
            if (c != -1) {
                float* s = &sums[(size_t)c * d];
                const float* v = vectors + (size_t)vi * d;
                for (int j = 0; j < d; j++) s[j] = s[j] + v[j];
                sizes[c]++;
            }
        }
        for (int c = 0; c < k; c++) if (sizes[c] > 0) {
            float cnt = (float)sizes[c];
            for (int j = 0; j < d; j++) centroids[(size_t)c * d + j] = sums[(size_t)c * d + j] / cnt;
        }
    }
    return k;
}
// KMeans clustering.go:60 ; KMeansSubspace clustering.go:112 (metric = L2SQ)
ORC_API int orc_kmeans(const float* vectors, int n, int d, int k, int metric, int max_iter, float* centroids, int* assign) {
    return kmeans_internal(vectors, n, d, k, metric, max_iter, centroids, assign);
}
// FindNearestCentroidIndex clustering.go:259-272
static int nearest_centroid(const float* v, const float* centroids, int k, int d, int metric) {
    float best = std::numeric_limits<float>::infinity();
    int bi = 0;
    for (int i = 0; i < k; i++) {
        float dist = dist_calc(metric, v, centroids + (size_t)i * d, d);
        if (dist < best) { best = dist; bi = i; }
    }
    return bi;
}
ORC_API int orc_nearest_centroid(const float* v, const float* centroids, int k, int d, int metric) {
    return nearest_centroid(v, centroids, k, d, metric);
}

// ---------------------------------------------------------------------------------------------
// ivf_index.go / ivf_index_search.go
// ---------------------------------------------------------------------------------------------
struct OIVF {
    int dim, metric, nlist;
    bool trained = false;
    std::vector<float> centroids;               // nlist×dim
    std::vector<std::vector<uint32_t>> list_ids;  // per list, in Add order
    std::vector<std::vector<float>> list_vecs;    // per list, row-major
    std::unordered_set<uint32_t> deleted;
};
ORC_API void* orc_ivf_new(int dim, int metric, int nlist) {
    auto* h = new OIVF(); h->dim = dim; h->metric = metric; h->nlist = nlist;
    h->list_ids.resize(nlist); h->list_vecs.resize(nlist); return h;
}
ORC_API void orc_ivf_free(void* p) { delete (OIVF*)p; }
// IVFIndex.Train ivf_index.go:206-235 — raw (un-normalised) training vectors even for cosine.
ORC_API int orc_ivf_train(void* p, const float* vecs, int n) {
    auto* h = (OIVF*)p;
    if (n < h->nlist) return ORC_ERR_TRAIN_DATA;
    h->centroids.assign((size_t)h->nlist * h->dim, 0.0f);
    std::vector<int> assign(n);
    int k = kmeans_internal(vecs, n, h->dim, h->nlist, h->metric, 20, h->centroids.data(), assign.data());
    if (k == 0) return ORC_ERR_ARG;
    h->trained = true;
    return ORC_OK;
}
// IVFIndex.Add ivf_index.go:251-280
ORC_API int orc_ivf_add(void* p, uint32_t id, const float* v) {
    auto* h = (OIVF*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    std::vector<float> tmp(h->dim);
    int rc = orc_preprocess(h->metric, v, h->dim, tmp.data());
    if (rc) return rc;
    int li = nearest_centroid(tmp.data(), h->centroids.data(), h->nlist, h->dim, h->metric);
    h->list_ids[li].push_back(id);
    h->list_vecs[li].insert(h->list_vecs[li].end(), tmp.begin(), tmp.end());
    return ORC_OK;
}
ORC_API int orc_ivf_remove(void* p, uint32_t id) {  // ivf_index.go Remove: soft delete
    auto* h = (OIVF*)p;
    bool found = false;
    for (auto& l : h->list_ids) if (std::find(l.begin(), l.end(), id) != l.end()) { found = true; break; }
    if (!found) return ORC_ERR_NOT_FOUND;
    if (h->deleted.count(id)) return ORC_ERR_ALREADY_DELETED;
    h->deleted.insert(id); return ORC_OK;
}
ORC_API const float* orc_ivf_centroids(void* p) { return ((OIVF*)p)->centroids.data(); }
ORC_API int orc_ivf_list_size(void* p, int l) { return (int)((OIVF*)p)->list_ids[l].size(); }

struct CDist { int index; float dist; };
// coarse ranking shared by IVF and IVFPQ: ivf_index_search.go:246-261, ivfpq_index_search.go:257-272
static std::vector<CDist> rank_centroids(int metric, const float* pq, const float* centroids, int nlist, int d) {
    std::vector<CDist> cd(nlist);
    for (int i = 0; i < nlist; i++) cd[i] = {i, dist_calc(metric, pq, centroids + (size_t)i * d, d)};
    std::stable_sort(cd.begin(), cd.end(), [](const CDist& a, const CDist& b) { return a.dist < b.dist; });
    return cd;
}
// ivfIndexSearch.searchSingleQuery ivf_index_search.go:217-322
ORC_API int orc_ivf_search(void* p, const float* q, int k, int nprobes, float threshold, const uint32_t* filter,
                           int n_filter, uint32_t* out_ids, float* out_scores, int cap) {
    auto* h = (OIVF*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    if (nprobes <= 0 || nprobes > h->nlist) nprobes = h->nlist;
    std::vector<float> pq(h->dim);
    int rc = orc_preprocess(h->metric, q, h->dim, pq.data());
    if (rc) return rc;
    auto cd = rank_centroids(h->metric, pq.data(), h->centroids.data(), h->nlist, h->dim);
    Filter f(filter, n_filter);
    std::vector<Cand> res;
    for (int i = 0; i < nprobes; i++) {
        int li = cd[i].index;
        const auto& ids = h->list_ids[li];
        for (size_t j = 0; j < ids.size(); j++) {
            if (h->deleted.count(ids[j])) continue;
            if (f.skip(ids[j])) continue;
            float dist = dist_calc(h->metric, pq.data(), &h->list_vecs[li][j * h->dim], h->dim);
            if (threshold > 0 && dist > threshold) continue;
            res.push_back({ids[j], dist});
        }
    }
    return emit_topk(res, k, out_ids, out_scores, cap);
}

// ---------------------------------------------------------------------------------------------
// pq_index.go / pq_index_search.go
// ---------------------------------------------------------------------------------------------
// PQIndex.encode pq_index.go:439-471 == IVFPQIndex.encodeResidual ivfpq_index.go:467-500
static void pq_encode(const float* v, const float* codebooks, int M, int Ksub, int dsub, uint8_t* code) {
    for (int m = 0; m < M; m++) {
        const float* sub = v + m * dsub;
        float best = std::numeric_limits<float>::infinity();
        int bi = 0;
        for (int ks = 0; ks < Ksub; ks++) {
            const float* c = codebooks + ((size_t)m * Ksub + ks) * dsub;
            float dist = 0.0f;
            for (int i = 0; i < dsub; i++) { float diff = sub[i] - c[i]; float sq = diff * diff; dist = dist + sq; }
            if (dist < best) { best = dist; bi = ks; }
        }
        code[m] = (uint8_t)bi;  // uint8(minIdx): truncates when Nbits > 8, exactly like the reference
    }
}
// LUT: pq_index_search.go:243-264 == ivfpq_index_search.go:350-375
static void pq_lut(const float* q, const float* codebooks, int M, int Ksub, int dsub, float* lut) {
    for (int m = 0; m < M; m++) {
        const float* sub = q + m * dsub;
        for (int ks = 0; ks < Ksub; ks++) {
            const float* c = codebooks + ((size_t)m * Ksub + ks) * dsub;
            float dist = 0.0f;
            for (int i = 0; i < dsub; i++) { float diff = sub[i] - c[i]; float sq = diff * diff; dist = dist + sq; }
            lut[(size_t)m * Ksub + ks] = dist;
        }
    }
}
// ADC: pq_index_search.go:289-295 == ivfpq_index_search.go:384-390 — always sqrt(sum), any metric.
static inline float pq_adc(const float* lut, const uint8_t* code, int M, int Ksub) {
    float dist = 0.0f;
    for (int m = 0; m < M; m++) dist = dist + lut[(size_t)m * Ksub + code[m]];
    return go_sqrt32(dist);
}
// train M codebooks on (sub)vectors: pq_index.go:193-250 / ivfpq_index.go:232-256
static int pq_train_codebooks(const float* vecs, int n, int dim, int M, int Ksub, int dsub, std::vector<float>& codebooks) {
    codebooks.assign((size_t)M * Ksub * dsub, 0.0f);
    std::vector<float> sub((size_t)n * dsub);
    std::vector<float> cent((size_t)std::min(Ksub, n) * dsub);
    std::vector<int> assign(n);
    for (int m = 0; m < M; m++) {
        for (int i = 0; i < n; i++) std::memcpy(&sub[(size_t)i * dsub], vecs + (size_t)i * dim + m * dsub, sizeof(float) * dsub);
        int k = kmeans_internal(sub.data(), n, dsub, Ksub, ORC_L2SQ, 20, cent.data(), assign.data());
        if (k == 0) return ORC_ERR_ARG;
        if (k < Ksub) return ORC_ERR_TRAIN_DATA;  // reference would index out of range (panic) copying centroids[k]
        std::memcpy(&codebooks[(size_t)m * Ksub * dsub], cent.data(), sizeof(float) * (size_t)Ksub * dsub);
    }
    return ORC_OK;
}

struct OPQ {
    int dim, metric, M, nbits, Ksub, dsub;
    bool trained = false;
    std::vector<float> codebooks;  // M × Ksub × dsub
    std::vector<uint8_t> codes;    // n × M
    std::vector<uint32_t> ids;
    std::unordered_set<uint32_t> deleted;
};
ORC_API void* orc_pq_new(int dim, int metric, int M, int nbits) {
    if (dim <= 0 || M <= 0 || dim % M != 0 || nbits <= 0 || nbits > 16) return nullptr;  // pq_index.go:135-155
    auto* h = new OPQ(); h->dim = dim; h->metric = metric; h->M = M; h->nbits = nbits; h->Ksub = 1 << nbits; h->dsub = dim / M;
    return h;
}
ORC_API void orc_pq_free(void* p) { delete (OPQ*)p; }
// PQIndex.Train pq_index.go:193-250
ORC_API int orc_pq_train(void* p, const float* vecs, int n) {
    auto* h = (OPQ*)p;
    if (n < h->Ksub) return ORC_ERR_TRAIN_DATA;
    int rc = pq_train_codebooks(vecs, n, h->dim, h->M, h->Ksub, h->dsub, h->codebooks);
    if (rc) return rc;
    h->trained = true; return ORC_OK;
}
// PQIndex.Add pq_index.go:263-289
ORC_API int orc_pq_add(void* p, uint32_t id, const float* v) {
    auto* h = (OPQ*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    std::vector<float> tmp(h->dim);
    int rc = orc_preprocess(h->metric, v, h->dim, tmp.data());
    if (rc) return rc;
    size_t off = h->codes.size();
    h->codes.resize(off + h->M);
    pq_encode(tmp.data(), h->codebooks.data(), h->M, h->Ksub, h->dsub, &h->codes[off]);
    h->ids.push_back(id);
    return ORC_OK;
}
ORC_API int orc_pq_remove(void* p, uint32_t id) {
    auto* h = (OPQ*)p;
    if (std::find(h->ids.begin(), h->ids.end(), id) == h->ids.end()) return ORC_ERR_NOT_FOUND;
    if (h->deleted.count(id)) return ORC_ERR_ALREADY_DELETED;
    h->deleted.insert(id); return ORC_OK;
}
ORC_API const float* orc_pq_codebooks(void* p) { return ((OPQ*)p)->codebooks.data(); }
ORC_API const uint8_t* orc_pq_codes(void* p) { return ((OPQ*)p)->codes.data(); }
ORC_API int orc_pq_size(void* p) { return (int)((OPQ*)p)->ids.size(); }
// pqIndexSearch.searchSingleQuery pq_index_search.go:218-325
ORC_API int orc_pq_search(void* p, const float* q, int k, float threshold, const uint32_t* filter, int n_filter,
                          uint32_t* out_ids, float* out_scores, int cap) {
    auto* h = (OPQ*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    if (h->ids.empty()) return 0;
    std::vector<float> pq(h->dim);
    int rc = orc_preprocess(h->metric, q, h->dim, pq.data());
    if (rc) return rc;
    std::vector<float> lut((size_t)h->M * h->Ksub);
    pq_lut(pq.data(), h->codebooks.data(), h->M, h->Ksub, h->dsub, lut.data());
    Filter f(filter, n_filter);
    std::vector<Cand> res; res.reserve(h->ids.size());
    for (size_t i = 0; i < h->ids.size(); i++) {
        if (h->deleted.count(h->ids[i])) continue;
        if (f.skip(h->ids[i])) continue;
        float fd = pq_adc(lut.data(), &h->codes[i * h->M], h->M, h->Ksub);
        if (threshold > 0 && fd > threshold) continue;
        res.push_back({h->ids[i], fd});
    }
    return emit_topk(res, k, out_ids, out_scores, cap);
}

// ---------------------------------------------------------------------------------------------
// ivfpq_index.go / ivfpq_index_search.go
// ---------------------------------------------------------------------------------------------
struct OIVFPQ {
    int dim, metric, nlist, M, nbits, Ksub, dsub;
    bool trained = false;
    std::vector<float> centroids, codebooks;
    std::vector<std::vector<uint32_t>> list_ids;
    std::vector<std::vector<uint8_t>> list_codes;
    std::unordered_set<uint32_t> deleted;
};
ORC_API void* orc_ivfpq_new(int dim, int metric, int nlist, int M, int nbits) {
    if (dim <= 0 || nlist <= 0 || M <= 0 || dim % M != 0 || nbits <= 0 || nbits > 16) return nullptr;  // ivfpq_index.go:113-147
    auto* h = new OIVFPQ(); h->dim = dim; h->metric = metric; h->nlist = nlist; h->M = M; h->nbits = nbits;
    h->Ksub = 1 << nbits; h->dsub = dim / M; h->list_ids.resize(nlist); h->list_codes.resize(nlist);
    return h;
}
ORC_API void orc_ivfpq_free(void* p) { delete (OIVFPQ*)p; }
// IVFPQIndex.Train ivfpq_index.go:180-259
ORC_API int orc_ivfpq_train(void* p, const float* vecs, int n) {
    auto* h = (OIVFPQ*)p;
    if (n < h->nlist * 10) return ORC_ERR_TRAIN_DATA;
    h->centroids.assign((size_t)h->nlist * h->dim, 0.0f);
    std::vector<int> assign(n);
    if (kmeans_internal(vecs, n, h->dim, h->nlist, h->metric, 20, h->centroids.data(), assign.data()) == 0) return ORC_ERR_ARG;
    std::vector<float> resid((size_t)n * h->dim);
    for (int i = 0; i < n; i++) {
        int a = nearest_centroid(vecs + (size_t)i * h->dim, h->centroids.data(), h->nlist, h->dim, h->metric);
        const float* c = &h->centroids[(size_t)a * h->dim];
        for (int d = 0; d < h->dim; d++) resid[(size_t)i * h->dim + d] = vecs[(size_t)i * h->dim + d] - c[d];
    }
    int rc = pq_train_codebooks(resid.data(), n, h->dim, h->M, h->Ksub, h->dsub, h->codebooks);
    if (rc) return rc;
    h->trained = true; return ORC_OK;
}
// load externally trained quantizers (used to search a GPU-built index on the CPU baseline)
ORC_API int orc_ivfpq_set_quantizers(void* p, const float* centroids, const float* codebooks) {
    auto* h = (OIVFPQ*)p;
    h->centroids.assign(centroids, centroids + (size_t)h->nlist * h->dim);
    h->codebooks.assign(codebooks, codebooks + (size_t)h->M * h->Ksub * h->dsub);
    h->trained = true; return ORC_OK;
}
ORC_API int orc_ivfpq_append_encoded(void* p, int list, int n, const uint32_t* ids, const uint8_t* codes) {
    auto* h = (OIVFPQ*)p;
    h->list_ids[list].insert(h->list_ids[list].end(), ids, ids + n);
    h->list_codes[list].insert(h->list_codes[list].end(), codes, codes + (size_t)n * h->M);
    return ORC_OK;
}
// IVFPQIndex.Add ivfpq_index.go:279-319
ORC_API int orc_ivfpq_add(void* p, uint32_t id, const float* v) {
    auto* h = (OIVFPQ*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    std::vector<float> tmp(h->dim), resid(h->dim);
    int rc = orc_preprocess(h->metric, v, h->dim, tmp.data());
    if (rc) return rc;
    int li = nearest_centroid(tmp.data(), h->centroids.data(), h->nlist, h->dim, h->metric);
    const float* c = &h->centroids[(size_t)li * h->dim];
    for (int d = 0; d < h->dim; d++) resid[d] = tmp[d] - c[d];
    size_t off = h->list_codes[li].size();
    h->list_codes[li].resize(off + h->M);
    pq_encode(resid.data(), h->codebooks.data(), h->M, h->Ksub, h->dsub, &h->list_codes[li][off]);
    h->list_ids[li].push_back(id);
    return ORC_OK;
}
ORC_API int orc_ivfpq_remove(void* p, uint32_t id) {
    auto* h = (OIVFPQ*)p;
    bool found = false;
    for (auto& l : h->list_ids) if (std::find(l.begin(), l.end(), id) != l.end()) { found = true; break; }
    if (!found) return ORC_ERR_NOT_FOUND;
    if (h->deleted.count(id)) return ORC_ERR_ALREADY_DELETED;
    h->deleted.insert(id); return ORC_OK;
}
ORC_API const float* orc_ivfpq_centroids(void* p) { return ((OIVFPQ*)p)->centroids.data(); }
ORC_API const float* orc_ivfpq_codebooks(void* p) { return ((OIVFPQ*)p)->codebooks.data(); }
ORC_API int orc_ivfpq_list_size(void* p, int l) { return (int)((OIVFPQ*)p)->list_ids[l].size(); }
ORC_API const uint8_t* orc_ivfpq_list_codes(void* p, int l) { return ((OIVFPQ*)p)->list_codes[l].data(); }
ORC_API const uint32_t* orc_ivfpq_list_ids(void* p, int l) { return ((OIVFPQ*)p)->list_ids[l].data(); }
// ivfpqIndexSearch.searchSingleQuery ivfpq_index_search.go:231-341
ORC_API int orc_ivfpq_search(void* p, const float* q, int k, int nprobes, float threshold, const uint32_t* filter,
                             int n_filter, uint32_t* out_ids, float* out_scores, int cap) {
    auto* h = (OIVFPQ*)p;
    if (!h->trained) return ORC_ERR_NOT_TRAINED;
    if (nprobes <= 0 || nprobes > h->nlist) nprobes = h->nlist;
    std::vector<float> pq(h->dim), resid(h->dim);
    int rc = orc_preprocess(h->metric, q, h->dim, pq.data());
    if (rc) return rc;
    auto cd = rank_centroids(h->metric, pq.data(), h->centroids.data(), h->nlist, h->dim);
    Filter f(filter, n_filter);
    std::vector<float> lut((size_t)h->M * h->Ksub);
    std::vector<Cand> res;
    for (int i = 0; i < nprobes; i++) {
        int li = cd[i].index;
        const float* c = &h->centroids[(size_t)li * h->dim];
        for (int d = 0; d < h->dim; d++) resid[d] = pq[d] - c[d];
        pq_lut(resid.data(), h->codebooks.data(), h->M, h->Ksub, h->dsub, lut.data());
        const auto& ids = h->list_ids[li];
        for (size_t j = 0; j < ids.size(); j++) {
            if (h->deleted.count(ids[j])) continue;
            if (f.skip(ids[j])) continue;
            float dist = pq_adc(lut.data(), &h->list_codes[li][j * h->M], h->M, h->Ksub);
            if (threshold > 0 && dist > threshold) continue;
            res.push_back({ids[j], dist});
        }
    }
    return emit_topk(res, k, out_ids, out_scores, cap);
}

// ---------------------------------------------------------------------------------------------
// hnsw_index.go / hnsw_index_search.go
// ---------------------------------------------------------------------------------------------
// container/heap semantics (Go stdlib heap.go: up/down, Push = append+up, Pop = swap(0,n-1)+down+pop).
template <class Less> struct GoHeap {
    std::vector<Cand> a; Less less;
    int len() const { return (int)a.size(); }
    void up(int j) { for (;;) { int i = (j - 1) / 2; if (i == j || !less(a[j], a[i])) break; std::swap(a[i], a[j]); j = i; } }
    bool down(int i0, int n) {
        int i = i0;
        for (;;) {
            int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
            int j = j1; int j2 = j1 + 1;
            if (j2 < n && less(a[j2], a[j1])) j = j2;
            if (!less(a[j], a[i])) break;
            std::swap(a[i], a[j]); i = j;
        }
        return i > i0;
    }
    void push(Cand c) { a.push_back(c); up(len() - 1); }
    Cand pop() { int n = len() - 1; std::swap(a[0], a[n]); down(0, n); Cand c = a.back(); a.pop_back(); return c; }
};
struct MinLess { bool operator()(const Cand& x, const Cand& y) const { return x.dist < y.dist; } };  // hnsw_index_search.go:379
struct MaxLess { bool operator()(const Cand& x, const Cand& y) const { return x.dist > y.dist; } };  // :423

struct ONode { uint32_t id; int level; std::vector<float> vec; std::vector<std::vector<uint32_t>> edges; };
struct OHNSW {
    int dim, metric, M, efC, efS;
    int max_level = -1; uint32_t entry = 0; uint32_t next_id = 0;
    std::unordered_map<uint32_t, ONode*> nodes;
    std::vector<uint32_t> insertion_order;
    std::unordered_set<uint32_t> deleted;
    uint64_t rng;
    // instrumentation for the roofline byte model (SURVEY §8d HNSW)
    uint64_t n_dist_evals = 0, n_expansions = 0;
    ~OHNSW() { for (auto& kv : nodes) delete kv.second; }
};
static inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
ORC_API void* orc_hnsw_new(int dim, int metric, int m, int efc, int efs, uint64_t seed) {
    if (dim <= 0) return nullptr;
    if (m <= 0) m = 16; if (efc <= 0) efc = 200; if (efs <= 0) efs = efc;  // hnsw_index.go:179-187
    auto* h = new OHNSW(); h->dim = dim; h->metric = metric; h->M = m; h->efC = efc; h->efS = efs; h->rng = seed;
    return h;
}
ORC_API void orc_hnsw_free(void* p) { delete (OHNSW*)p; }
// randomLevel hnsw_index.go:474-484 — geometric(p = 1/M), cap 16. The reference draws from the
// unseeded global math/rand/v2; the oracle uses SplitMix64 (u = (next>>11)·2^-53) so graphs are
// reproducible. Levels can also be supplied explicitly (orc_hnsw_add_with_level).
static int random_level(OHNSW* h) {
    double prob = 1.0 / (double)h->M; int level = 0;
    while (level < 16) { double u = (double)(splitmix64(h->rng) >> 11) * (1.0 / 9007199254740992.0); if (!(u < prob)) break; level++; }
    return level;
}
static inline float hdist(OHNSW* h, const float* q, uint32_t id) { h->n_dist_evals++; return dist_calc(h->metric, q, h->nodes[id]->vec.data(), h->dim); }
// HNSWIndex.searchLayer hnsw_index.go:565-629
static std::vector<Cand> search_layer(OHNSW* h, const float* q, uint32_t entry, int ef, int layer) {
    std::unordered_set<uint32_t> visited;
    GoHeap<MinLess> cand; GoHeap<MaxLess> result;
    if (!h->deleted.count(entry)) {
        float d = hdist(h, q, entry);
        cand.push({entry, d}); result.push({entry, d});
    }
    visited.insert(entry);
    while (cand.len() > 0) {
        Cand cur = cand.pop();
        if (result.len() >= ef && cur.dist > result.a[0].dist) break;
        ONode* node = h->nodes[cur.id];
        h->n_expansions++;
        if (layer < (int)node->edges.size()) {
            for (uint32_t nb : node->edges[layer]) {
                if (h->deleted.count(nb)) continue;
                if (!visited.count(nb)) {
                    visited.insert(nb);
                    float d = hdist(h, q, nb);
                    if (result.len() < ef || d < result.a[0].dist) {
                        cand.push({nb, d}); result.push({nb, d});
                        if (result.len() > ef) result.pop();
                    }
                }
            }
        }
    }
    std::vector<Cand> fin(result.len());
    for (int i = result.len() - 1; i >= 0; i--) fin[i] = result.pop();
    return fin;
}
// selectNeighbors hnsw_index.go:637-656 (sort.Slice → canonical stable order)
static std::vector<uint32_t> select_neighbors(std::vector<Cand>& c, int M) {
    std::vector<uint32_t> r;
    if ((int)c.size() <= M) { for (auto& x : c) r.push_back(x.id); return r; }
    stable_sort_asc(c);
    for (int i = 0; i < M; i++) r.push_back(c[i].id);
    return r;
}
// pruneConnections hnsw_index.go:667-694 — NOTE: the node being inserted is not yet in idx.nodes
// (hnsw_index.go:281-282 inserts it after insertNode returns), so `idx.nodes[nid] == nil` drops the
// fresh back-edge whenever a neighbour is pruned. Restated faithfully.
static void prune_connections(OHNSW* h, uint32_t node_id, int layer, int M) {
    ONode* node = h->nodes[node_id];
    std::vector<Cand> cl;
    for (uint32_t nid : node->edges[layer]) {
        auto it = h->nodes.find(nid);
        if (it == h->nodes.end()) continue;
        float d = dist_calc(h->metric, node->vec.data(), it->second->vec.data(), h->dim);
        cl.push_back({nid, d});
    }
    stable_sort_asc(cl);
    int nn = std::min(M, (int)cl.size());
    node->edges[layer].resize(nn);
    for (int i = 0; i < nn; i++) node->edges[layer][i] = cl[i].id;
}
// insertNode hnsw_index.go:493-552
static void insert_node(OHNSW* h, ONode* node) {
    uint32_t curr = h->entry;
    float curr_dist = dist_calc(h->metric, node->vec.data(), h->nodes[curr]->vec.data(), h->dim);
    for (int lc = h->max_level; lc > node->level; lc--) {
        bool changed = true;
        while (changed) {
            changed = false;
            ONode* cn = h->nodes[curr];
            if (lc < (int)cn->edges.size()) {
                for (uint32_t nb : cn->edges[lc]) {
                    if (h->deleted.count(nb)) continue;
                    float d = dist_calc(h->metric, node->vec.data(), h->nodes[nb]->vec.data(), h->dim);
                    if (d < curr_dist) { curr_dist = d; curr = nb; changed = true; }
                }
            }
        }
    }
    for (int lc = node->level; lc >= 0; lc--) {
        auto cands = search_layer(h, node->vec.data(), curr, h->efC, lc);
        int M = h->M; if (lc == 0) M *= 2;
        auto nbs = select_neighbors(cands, M);
        for (uint32_t nb : nbs) {
            node->edges[lc].push_back(nb);
            ONode* nn = h->nodes[nb];
            if (lc <= nn->level) {
                nn->edges[lc].push_back(node->id);
                if ((int)nn->edges[lc].size() > M) prune_connections(h, nb, lc, M);
            }
        }
        if (!cands.empty()) curr = cands[0].id;
    }
}
// HNSWIndex.Add hnsw_index.go:228-288 (level < 0 → draw from the seeded RNG)
ORC_API int orc_hnsw_add_with_level(void* p, uint32_t id_in, const float* v, int level) {
    auto* h = (OHNSW*)p;
    std::vector<float> tmp(h->dim);
    int rc = orc_preprocess(h->metric, v, h->dim, tmp.data());
    if (rc) return rc;
    uint32_t id = id_in;
    if (level < 0) level = random_level(h);
    if (id == 0) { id = h->next_id; h->next_id++; }
    if (level > h->max_level) h->max_level = level;
    ONode* node = new ONode(); node->id = id; node->level = level; node->vec = tmp; node->edges.resize(level + 1);
    if (h->entry == 0 && h->nodes.empty()) { h->entry = id; h->nodes[id] = node; h->insertion_order.push_back(id); return ORC_OK; }
    insert_node(h, node);
    auto it = h->nodes.find(id);
    if (it != h->nodes.end()) delete it->second;
    h->nodes[id] = node;
    h->insertion_order.push_back(id);
    return ORC_OK;
}
ORC_API int orc_hnsw_add(void* p, uint32_t id, const float* v) { return orc_hnsw_add_with_level(p, id, v, -1); }
ORC_API int orc_hnsw_remove(void* p, uint32_t id) {
    auto* h = (OHNSW*)p;
    if (!h->nodes.count(id)) return ORC_ERR_NOT_FOUND;
    if (h->deleted.count(id)) return ORC_ERR_ALREADY_DELETED;
    h->deleted.insert(id); return ORC_OK;
}
ORC_API int orc_hnsw_size(void* p) { return (int)((OHNSW*)p)->nodes.size(); }
ORC_API int orc_hnsw_max_level(void* p) { return ((OHNSW*)p)->max_level; }
ORC_API uint32_t orc_hnsw_entry(void* p) { return ((OHNSW*)p)->entry; }
ORC_API void orc_hnsw_stats(void* p, uint64_t* evals, uint64_t* expansions, int reset) {
    auto* h = (OHNSW*)p; *evals = h->n_dist_evals; *expansions = h->n_expansions;
    if (reset) { h->n_dist_evals = 0; h->n_expansions = 0; }
}
// Export the graph in insertion order: ids[n], levels[n], vecs[n*dim] (preprocessed), and for each
// (node, layer<=level) an edge list. edge_offsets has (sum(level+1) + 1) entries. Two-call protocol:
// pass null arrays to get sizes.
ORC_API int orc_hnsw_export(void* p, uint32_t* ids, int* levels, float* vecs, int64_t* edge_offsets, uint32_t* edges,
                            int64_t* n_slots_out, int64_t* n_edges_out) {
    auto* h = (OHNSW*)p;
    int64_t slots = 0, ne = 0;
    // a re-added id appears once in nodes; de-duplicate insertion_order keeping the last occurrence
    std::vector<uint32_t> order; std::unordered_set<uint32_t> seen;
    for (auto it = h->insertion_order.rbegin(); it != h->insertion_order.rend(); ++it) if (!seen.count(*it)) { seen.insert(*it); order.push_back(*it); }
    std::reverse(order.begin(), order.end());
    for (uint32_t id : order) { ONode* n = h->nodes[id]; slots += n->level + 1; for (auto& e : n->edges) ne += (int64_t)e.size(); }
    if (n_slots_out) *n_slots_out = slots;
    if (n_edges_out) *n_edges_out = ne;
    if (!ids) return (int)order.size();
    int64_t s = 0, e = 0;
    for (size_t i = 0; i < order.size(); i++) {
        ONode* n = h->nodes[order[i]];
        ids[i] = n->id; levels[i] = n->level;
        std::memcpy(vecs + i * h->dim, n->vec.data(), sizeof(float) * h->dim);
        for (int l = 0; l <= n->level; l++) {
            edge_offsets[s++] = e;
            for (uint32_t x : n->edges[l]) edges[e++] = x;
        }
    }
    edge_offsets[s] = e;
    return (int)order.size();
}
// hnswIndexSearch.searchSingleQuery hnsw_index_search.go:248-354
ORC_API int orc_hnsw_search(void* p, const float* q, int k, int ef_search, float threshold, const uint32_t* filter,
                            int n_filter, uint32_t* out_ids, float* out_scores, int cap) {
    auto* h = (OHNSW*)p;
    if (h->nodes.empty() || h->max_level == -1) return 0;
    std::vector<float> pq(h->dim);
    int rc = orc_preprocess(h->metric, q, h->dim, pq.data());
    if (rc) return rc;
    uint32_t curr = h->entry;
    float curr_dist = hdist(h, pq.data(), curr);
    for (int lc = h->max_level; lc > 0; lc--) {
        bool changed = true;
        while (changed) {
            changed = false;
            ONode* node = h->nodes[curr];
            if (lc < (int)node->edges.size()) {
                for (uint32_t nb : node->edges[lc]) {
                    if (h->deleted.count(nb)) continue;
                    float d = hdist(h, pq.data(), nb);
                    if (d < curr_dist) { curr_dist = d; curr = nb; changed = true; }
                }
            }
        }
    }
    int ef = ef_search; if (ef <= 0) ef = h->efS;
    auto cands = search_layer(h, pq.data(), curr, ef, 0);
    Filter f(filter, n_filter);
    std::vector<Cand> res;
    for (auto& c : cands) {
        if (f.skip(c.id)) continue;
        if (threshold > 0 && c.dist > threshold) continue;
        res.push_back(c);
    }
    return emit_topk(res, k, out_ids, out_scores, cap);
}

// ---------------------------------------------------------------------------------------------
// bm25_index.go / bm25_index_search.go — scoring from token ids (tokenisation, NFKC and lower-casing
// live in third-party uax29 v2.2.0 / x/text v0.30.0 and are not restated; BM25 parity starts at
// token ids).
// ---------------------------------------------------------------------------------------------
// math.Log: Go's pure-Go implementation on amd64 (src/math/log.go) is the FreeBSD e_log.c algorithm;
// restated here so idf is bit-identical to the reference's rather than to glibc's log.
static double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (std::isnan(x) || std::isinf(x) && x > 0) return x;
    if (x < 0) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0) return -std::numeric_limits<double>::infinity();
    int ki; double f1 = std::frexp(x, &ki);
    if (f1 < 0.70710678118654752440 /* Sqrt2/2 */) { f1 *= 2; ki--; }
    double f = f1 - 1; double k = (double)ki;
    double s = f / (2 + f); double s2 = s * s; double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2; double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
ORC_API double orc_go_log(double x) { return go_log(x); }

struct OBM25 {
    std::map<uint32_t, std::vector<uint32_t>> postings;             // term -> sorted doc ids (roaring iterates ascending)
    std::unordered_map<uint32_t, std::unordered_map<uint32_t, int>> tf;  // term -> doc -> tf
    std::unordered_map<uint32_t, int> doc_len;
    std::unordered_map<uint32_t, std::vector<uint32_t>> doc_tokens;
    std::unordered_set<uint32_t> deleted;
    uint32_t num_docs = 0; long total_tokens = 0; double avg_doc_len = 0;
};
ORC_API void* orc_bm25_new() { return new OBM25(); }
ORC_API void orc_bm25_free(void* p) { delete (OBM25*)p; }
static void bm25_update_avg(OBM25* h) { h->avg_doc_len = h->num_docs == 0 ? 0 : (double)h->total_tokens / (double)h->num_docs; }
static void bm25_remove_internal(OBM25* h, uint32_t id) {  // bm25_index.go removeInternal
    auto it = h->doc_tokens.find(id); if (it == h->doc_tokens.end()) return;
    int dl = h->doc_len[id];
    for (uint32_t t : it->second) {
        auto pit = h->postings.find(t);
        if (pit != h->postings.end()) { auto& v = pit->second; auto f = std::lower_bound(v.begin(), v.end(), id); if (f != v.end() && *f == id) v.erase(f); if (v.empty()) h->postings.erase(pit); }
        auto tit = h->tf.find(t);
        if (tit != h->tf.end()) { tit->second.erase(id); if (tit->second.empty()) h->tf.erase(tit); }
    }
    h->doc_tokens.erase(it); h->doc_len.erase(id); h->num_docs--; h->total_tokens -= dl;
    if (h->num_docs > 0) bm25_update_avg(h); else { h->avg_doc_len = 0; h->total_tokens = 0; }
}
// BM25SearchIndex.Add bm25_index.go:168-201 (tokens already ids)
ORC_API int orc_bm25_add(void* p, uint32_t id, const uint32_t* tokens, int n) {
    auto* h = (OBM25*)p;
    if (h->doc_tokens.count(id)) bm25_remove_internal(h, id);
    h->doc_tokens[id].assign(tokens, tokens + n);
    h->doc_len[id] = n; h->num_docs++; h->total_tokens += n;
    for (int i = 0; i < n; i++) {
        auto& v = h->postings[tokens[i]];
        auto f = std::lower_bound(v.begin(), v.end(), id);
        if (f == v.end() || *f != id) v.insert(f, id);
        h->tf[tokens[i]][id]++;
    }
    bm25_update_avg(h);
    return ORC_OK;
}
ORC_API int orc_bm25_remove(void* p, uint32_t id) {  // soft delete bm25_index.go:203-222
    auto* h = (OBM25*)p;
    if (!h->doc_tokens.count(id)) return ORC_OK;
    h->deleted.insert(id); return ORC_OK;
}
// BM25SearchIndex.Flush bm25_index.go:374-400: hard-delete every soft-deleted document (roaring iterates ascending), which
// changes N, df and avgDocLen — scores after a flush differ from the soft-deleted state
ORC_API void orc_bm25_flush(void* p) {
    auto* h = (OBM25*)p;
    std::vector<uint32_t> d(h->deleted.begin(), h->deleted.end()); std::sort(d.begin(), d.end());
    for (uint32_t id : d) bm25_remove_internal(h, id);
    h->deleted.clear();
}
ORC_API uint32_t orc_bm25_num_docs(void* p) { return ((OBM25*)p)->num_docs; }
ORC_API double orc_bm25_avg_doc_len(void* p) { return ((OBM25*)p)->avg_doc_len; }
// bm25TextSearch.searchSingleQuery bm25_index_search.go:278-397. Scores accumulate in float64 in
// query-token order (duplicate query tokens counted again, :299); result cast to float32 (:392).
// Ordering: score descending; exact ties are undefined in the reference (map iteration order feeds a
// heap) — canonical here: ascending doc id among equal float64 scores.
ORC_API int orc_bm25_search(void* p, const uint32_t* qtokens, int nq, int k, const uint32_t* filter, int n_filter,
                            uint32_t* out_ids, float* out_scores, double* out_scores64, int cap) {
    auto* h = (OBM25*)p;
    if (nq == 0) return 0;
    std::map<uint32_t, double> scores;
    double N = (double)h->num_docs;
    if (N == 0) return 0;
    Filter f(filter, n_filter);
    const double K1 = 1.2, B = 0.75;
    const double K1p1 = 2.2;      // untyped-constant K1 + 1 folds exactly to 2.2 (bm25_index.go:75-80)
    const double omB = 0.25;      // 1 - B
    for (int qi = 0; qi < nq; qi++) {
        auto pit = h->postings.find(qtokens[qi]);
        if (pit == h->postings.end()) continue;
        double df = (double)pit->second.size();
        double idf = go_log((N - df + 0.5) / (df + 0.5) + 1.0);
        auto& tfm = h->tf[qtokens[qi]];
        for (uint32_t doc : pit->second) {
            if (h->deleted.count(doc)) continue;
            if (f.skip(doc)) continue;
            double tfv = (double)tfm[doc];
            double dl = (double)h->doc_len[doc];
            double r = dl / h->avg_doc_len;
            double inner = omB + B * r;
            double den = tfv + K1 * inner;
            double num = idf * (tfv * K1p1);
            double score = num / den;
            scores[doc] += score;
        }
    }
    struct DS { uint32_t id; double s; };
    std::vector<DS> all; all.reserve(scores.size());
    for (auto& kv : scores) all.push_back({kv.first, kv.second});
    std::stable_sort(all.begin(), all.end(), [](const DS& a, const DS& b) { return a.s > b.s; });
    int cnt = (k <= 0 || k >= (int)all.size()) ? (int)all.size() : k;
    int n = std::min(cnt, cap);
    for (int i = 0; i < n; i++) { out_ids[i] = all[i].id; out_scores[i] = (float)all[i].s; if (out_scores64) out_scores64[i] = all[i].s; }
    return cnt;
}

// ---------------------------------------------------------------------------------------------
// aggregation.go:107-255 — multi-query aggregation by node id (sum / max / mean), ascending sort.
// kind: 0 sum, 1 max, 2 mean. Canonical tie order: first appearance of the id.
// ---------------------------------------------------------------------------------------------
ORC_API int orc_aggregate(int kind, const uint32_t* ids, const float* scores, int n, uint32_t* out_ids, float* out_scores) {
    std::vector<uint32_t> order; std::unordered_map<uint32_t, std::vector<float>> m;
    for (int i = 0; i < n; i++) { if (!m.count(ids[i])) order.push_back(ids[i]); m[ids[i]].push_back(scores[i]); }
    std::vector<Cand> agg;
    for (uint32_t id : order) {
        auto& s = m[id]; float v = 0.0f;
        if (kind == 0) { for (float x : s) v = v + x; }
        else if (kind == 1) { v = s[0]; for (size_t i = 1; i < s.size(); i++) if (s[i] > v) v = s[i]; }
        else { for (float x : s) v = v + x; v = v / (float)s.size(); }
        agg.push_back({id, v});
    }
    stable_sort_asc(agg);
    for (size_t i = 0; i < agg.size(); i++) { out_ids[i] = agg[i].id; out_scores[i] = agg[i].dist; }
    return (int)agg.size();
}

// ---------------------------------------------------------------------------------------------
// fusion.go:174-243 — Reciprocal Rank Fusion. Inputs are (id, score) lists standing for the Go maps;
// ranks come from the reference's O(n²) exchange sort (vector ascending, text descending) applied to
// the list in the given order (Go map order is random; ties therefore undefined in the reference).
// ---------------------------------------------------------------------------------------------
static std::vector<std::pair<uint32_t, int>> score_ranks(const uint32_t* ids, const double* sc, int n, bool asc) {
    std::vector<std::pair<uint32_t, double>> s(n);
    for (int i = 0; i < n; i++) s[i] = {ids[i], sc[i]};
    for (int i = 0; i + 1 < n; i++) for (int j = i + 1; j < n; j++) {
        bool sw = asc ? s[i].second > s[j].second : s[i].second < s[j].second;
        if (sw) std::swap(s[i], s[j]);
    }
    std::vector<std::pair<uint32_t, int>> r(n);
    for (int i = 0; i < n; i++) r[i] = {s[i].first, i};
    return r;
}
ORC_API int orc_rrf(double K, const uint32_t* vids, const double* vsc, int nv, const uint32_t* tids, const double* tsc, int nt,
                    uint32_t* out_ids, double* out_scores) {
    std::map<uint32_t, double> comb;
    for (auto& pr : score_ranks(vids, vsc, nv, true)) comb[pr.first] = 1.0 / (K + (double)pr.second);
    for (auto& pr : score_ranks(tids, tsc, nt, false)) {
        double r = 1.0 / (K + (double)pr.second);
        auto it = comb.find(pr.first);
        if (it != comb.end()) it->second = it->second + r; else comb[pr.first] = r;
    }
    int i = 0;
    for (auto& kv : comb) { out_ids[i] = kv.first; out_scores[i] = kv.second; i++; }
    return i;
}

// ---------------------------------------------------------------------------------------------
// Synthetic data: SplitMix64 stream, u = (next>>40)·2^-24, value = 2u-1 in [-1,1).
// The GPU library generates the same stream on device (comet_synth_fill) — bit-identical.
// ---------------------------------------------------------------------------------------------
ORC_API void orc_synth_fill(uint64_t seed, uint64_t offset, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        uint64_t s = seed + (offset + i) * 0x9E3779B97F4A7C15ull;  // state before the (offset+i)-th draw
        uint64_t z = splitmix64(s);
        float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        out[i] = 2.0f * u - 1.0f;
    }
}

// clustered variant: centre[blob(r)][j] + sigma * u(seed, r*dim+j); see comet_synth_mixture_dev
static inline float synth_u(uint64_t seed, uint64_t counter) {
    uint64_t s = seed + counter * 0x9E3779B97F4A7C15ull; uint64_t z = splitmix64(s);
    return 2.0f * ((float)(z >> 40) * (1.0f / 16777216.0f)) - 1.0f;
}
ORC_API void orc_synth_mixture(uint64_t seed, int n_centers, float sigma, int n_sub, float sigma_noise, uint64_t row_base, uint64_t n_rows, int dim, float* out) {
    if (n_centers <= 0) { orc_synth_fill(seed, row_base * (uint64_t)dim, n_rows * (uint64_t)dim, out); return; }
    for (uint64_t i = 0; i < n_rows; i++) {
        const uint64_t r = row_base + i, b1 = ((r * 2654435761ull) >> 7) % (uint64_t)n_centers;
        const uint64_t b2 = n_sub > 0 ? ((r * 0x9E3779B1ull) >> 5) % (uint64_t)n_sub : 0;
        for (int j = 0; j < dim; j++) {
            float v = synth_u(seed ^ 0x5EEDull, b1 * (uint64_t)dim + j);
            const float nz = synth_u(seed, r * (uint64_t)dim + j);
            if (n_sub > 0) { const float s2 = synth_u(seed ^ 0x5EED2ull, b2 * (uint64_t)dim + j) * sigma; v = v + s2; const float t = nz * sigma_noise; v = v + t; }
            else { const float t = nz * sigma; v = v + t; }
            out[i * dim + j] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// On-disk formats: WriteTo / ReadFrom of the five vector indexes, restated from
//   flat_index.go:366-470 / :488-614      ("FLAT")     ivf_index.go:468-588 / :620-785     ("IVFX")
//   pq_index.go:509-640 / :672-846        ("PQIX")     ivfpq_index.go:544-700 / :745-936   ("IVPQ")
//   hnsw_index.go:734-880 / :898-1096     ("HNSW")
// binary.Write(w, binary.LittleEndian, x) of uint32/int32/float32/float64/uint8 = the value's little-endian bytes.
// Every WriteTo calls Flush() first, so deletedNodes is empty when serialised: roaring v1.9.4 (go.mod:6, not vendored)
// writes an empty bitmap in its portable format as the 8 bytes {uint32 12346 (serialCookieNoRunContainer),
// uint32 0 containers} (roaringArray.writeTo) — the only roaring bytes the reference can emit on this path.
// ReadFrom here accepts exactly that empty tail (anything else: ORC_ERR_ARG).
// HNSW iterates a Go map in WriteTo and Flush (undefined order): nodes are visited in ascending id order here.
// API: orc_*_write(h, buf, cap) -> bytes needed (writes only if cap suffices); orc_*_read(h, buf, len) ->
// bytes consumed, or a negative ORC_ERR_* (ORC_ERR_ARG for magic / version / parameter mismatch, ORC_ERR_DIM
// for a dimension mismatch, -8 for a short read).
// ---------------------------------------------------------------------------------------------
static const int ORC_ERR_EOF = -8;
struct OW {
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
    void u8(uint8_t v) { raw(&v, 1); } void u32(uint32_t v) { raw(&v, 4); } void i32(int32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); } void f64(double v) { raw(&v, 8); }
    void kind(int metric) { const char* s = metric == ORC_L2 ? "l2" : metric == ORC_L2SQ ? "l2_squared" : "cosine"; u32((uint32_t)std::strlen(s)); raw(s, std::strlen(s)); }
    void empty_bitmap() { u32(8); u32(12346u); u32(0u); }
    int64_t finish(uint8_t* out, int64_t cap) const { if (out && (int64_t)b.size() <= cap) std::memcpy(out, b.data(), b.size()); return (int64_t)b.size(); }
};
struct OR {
    const uint8_t* p; int64_t len, off = 0; bool eof = false;
    OR(const uint8_t* p_, int64_t l) : p(p_), len(l) {}
    bool raw(void* d, size_t n) { if (off + (int64_t)n > len) { eof = true; std::memset(d, 0, n); return false; } std::memcpy(d, p + off, n); off += (int64_t)n; return true; }
    uint8_t u8() { uint8_t v = 0; raw(&v, 1); return v; } uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    int32_t i32() { int32_t v = 0; raw(&v, 4); return v; } double f64() { double v = 0; raw(&v, 8); return v; }
    // magic + version + dim + distance kind; 0 or a negative error
    int header(const char* magic, int dim, int metric) {
        char m[4]; if (!raw(m, 4)) return ORC_ERR_EOF;
        if (std::memcmp(m, magic, 4) != 0) return ORC_ERR_ARG;
        if (u32() != 1u) return eof ? ORC_ERR_EOF : ORC_ERR_ARG;
        if ((int)u32() != dim) return eof ? ORC_ERR_EOF : ORC_ERR_DIM;
        uint32_t kl = u32(); if (eof) return ORC_ERR_EOF; if (kl > 64) return ORC_ERR_ARG;
        char k[65] = {0}; if (!raw(k, kl)) return ORC_ERR_EOF;
        const char* want = metric == ORC_L2 ? "l2" : metric == ORC_L2SQ ? "l2_squared" : "cosine";
        return std::strcmp(k, want) == 0 ? 0 : ORC_ERR_ARG;
    }
    int empty_bitmap() { uint32_t sz = u32(); if (eof) return ORC_ERR_EOF; if (sz != 8) return ORC_ERR_ARG; uint32_t a = u32(), b = u32(); if (eof) return ORC_ERR_EOF; return (a == 12346u && b == 0u) ? 0 : ORC_ERR_ARG; }
};

ORC_API int64_t orc_flat_write(void* p, uint8_t* out, int64_t cap) {
    auto* h = (OFlat*)p; orc_flat_flush(p);                       // flat_index.go:368
    OW w; w.raw("FLAT", 4); w.u32(1); w.u32((uint32_t)h->dim); w.kind(h->metric);
    w.u32((uint32_t)h->ids.size());
    for (size_t i = 0; i < h->ids.size(); i++) { w.u32(h->ids[i]); w.u32((uint32_t)h->dim); w.raw(&h->vecs[i * h->dim], (size_t)h->dim * 4); }
    w.empty_bitmap();
    return w.finish(out, cap);
}
ORC_API int64_t orc_flat_read(void* p, const uint8_t* in, int64_t len) {
    auto* h = (OFlat*)p; OR r(in, len);
    if (int rc = r.header("FLAT", h->dim, h->metric)) return rc;
    uint32_t n = r.u32(); if (r.eof) return ORC_ERR_EOF;
    std::vector<uint32_t> ids(n); std::vector<float> vecs((size_t)n * h->dim);
    for (uint32_t i = 0; i < n; i++) {
        ids[i] = r.u32(); uint32_t vd = r.u32(); if (r.eof) return ORC_ERR_EOF;
        if ((int)vd != h->dim) return ORC_ERR_DIM;                 // :573
        if (!r.raw(&vecs[(size_t)i * h->dim], (size_t)h->dim * 4)) return ORC_ERR_EOF;
    }
    if (int rc = r.empty_bitmap()) return rc;
    h->ids.swap(ids); h->vecs.swap(vecs); h->deleted.clear();
    return r.off;
}

// IVFIndex.Flush ivf_index.go:362-400
ORC_API void orc_ivf_flush(void* p) {
    auto* h = (OIVF*)p; if (h->deleted.empty()) return;
    for (int l = 0; l < h->nlist; l++) {
        std::vector<uint32_t> ids; std::vector<float> vecs;
        for (size_t j = 0; j < h->list_ids[l].size(); j++) if (!h->deleted.count(h->list_ids[l][j])) {
            ids.push_back(h->list_ids[l][j]); vecs.insert(vecs.end(), h->list_vecs[l].begin() + j * h->dim, h->list_vecs[l].begin() + (j + 1) * h->dim);
        }
        h->list_ids[l].swap(ids); h->list_vecs[l].swap(vecs);
    }
    h->deleted.clear();
}
ORC_API int64_t orc_ivf_write(void* p, uint8_t* out, int64_t cap) {
    auto* h = (OIVF*)p; orc_ivf_flush(p);
    OW w; w.raw("IVFX", 4); w.u32(1); w.u32((uint32_t)h->dim); w.kind(h->metric); w.u32((uint32_t)h->nlist); w.u8(h->trained ? 1 : 0);
    if (h->trained) for (int l = 0; l < h->nlist; l++) { w.u32((uint32_t)h->dim); w.raw(&h->centroids[(size_t)l * h->dim], (size_t)h->dim * 4); }
    w.u32((uint32_t)h->nlist);
    for (int l = 0; l < h->nlist; l++) {
        w.u32((uint32_t)h->list_ids[l].size());
        for (size_t j = 0; j < h->list_ids[l].size(); j++) { w.u32(h->list_ids[l][j]); w.raw(&h->list_vecs[l][j * h->dim], (size_t)h->dim * 4); }
    }
    w.empty_bitmap();
    return w.finish(out, cap);
}
ORC_API int64_t orc_ivf_read(void* p, const uint8_t* in, int64_t len) {
    auto* h = (OIVF*)p; OR r(in, len);
    if (int rc = r.header("IVFX", h->dim, h->metric)) return rc;
    if ((int)r.u32() != h->nlist) return r.eof ? ORC_ERR_EOF : ORC_ERR_ARG;
    bool tr = r.u8() == 1; if (r.eof) return ORC_ERR_EOF;
    std::vector<float> cen;
    if (tr) { cen.resize((size_t)h->nlist * h->dim); for (int l = 0; l < h->nlist; l++) { uint32_t cs = r.u32(); if (r.eof) return ORC_ERR_EOF; if ((int)cs != h->dim) return ORC_ERR_DIM; if (!r.raw(&cen[(size_t)l * h->dim], (size_t)h->dim * 4)) return ORC_ERR_EOF; } }
    uint32_t lc = r.u32(); if (r.eof) return ORC_ERR_EOF; if ((int)lc != h->nlist) return ORC_ERR_ARG;
    std::vector<std::vector<uint32_t>> li(lc); std::vector<std::vector<float>> lv(lc);
    for (uint32_t l = 0; l < lc; l++) {
        uint32_t n = r.u32(); if (r.eof) return ORC_ERR_EOF;
        li[l].resize(n); lv[l].resize((size_t)n * h->dim);
        for (uint32_t j = 0; j < n; j++) { li[l][j] = r.u32(); if (!r.raw(&lv[l][(size_t)j * h->dim], (size_t)h->dim * 4)) return ORC_ERR_EOF; }
    }
    if (int rc = r.empty_bitmap()) return rc;
    h->trained = tr; if (tr) h->centroids.swap(cen); h->list_ids.swap(li); h->list_vecs.swap(lv); h->deleted.clear();
    return r.off;
}

// PQIndex.Flush pq_index.go:340-380
ORC_API void orc_pq_flush(void* p) {
    auto* h = (OPQ*)p; if (h->deleted.empty()) return;
    std::vector<uint32_t> ids; std::vector<uint8_t> codes;
    for (size_t i = 0; i < h->ids.size(); i++) if (!h->deleted.count(h->ids[i])) { ids.push_back(h->ids[i]); codes.insert(codes.end(), h->codes.begin() + i * h->M, h->codes.begin() + (i + 1) * h->M); }
    h->ids.swap(ids); h->codes.swap(codes); h->deleted.clear();
}
ORC_API int64_t orc_pq_write(void* p, uint8_t* out, int64_t cap) {
    auto* h = (OPQ*)p; orc_pq_flush(p);
    OW w; w.raw("PQIX", 4); w.u32(1); w.u32((uint32_t)h->dim); w.kind(h->metric);
    w.u32((uint32_t)h->M); w.u32((uint32_t)h->nbits); w.u32((uint32_t)h->Ksub); w.u32((uint32_t)h->dsub); w.u8(h->trained ? 1 : 0);
    const size_t cbs = (size_t)h->Ksub * h->dsub;
    if (h->trained) for (int m = 0; m < h->M; m++) { w.u32((uint32_t)cbs); w.raw(&h->codebooks[m * cbs], cbs * 4); }
    w.u32((uint32_t)h->ids.size());
    for (size_t i = 0; i < h->ids.size(); i++) { w.u32(h->ids[i]); w.raw(&h->codes[i * h->M], (size_t)h->M); }
    w.empty_bitmap();
    return w.finish(out, cap);
}
ORC_API int64_t orc_pq_read(void* p, const uint8_t* in, int64_t len) {
    auto* h = (OPQ*)p; OR r(in, len);
    if (int rc = r.header("PQIX", h->dim, h->metric)) return rc;
    uint32_t M = r.u32(), nb = r.u32(), K = r.u32(), ds = r.u32(); if (r.eof) return ORC_ERR_EOF;
    if ((int)M != h->M || (int)nb != h->nbits || (int)K != h->Ksub || (int)ds != h->dsub) return ORC_ERR_ARG;   // pq_index.go:737-748
    bool tr = r.u8() == 1; if (r.eof) return ORC_ERR_EOF;
    const size_t cbs = (size_t)h->Ksub * h->dsub;
    std::vector<float> cb;
    if (tr) { cb.resize((size_t)h->M * cbs); for (int m = 0; m < h->M; m++) { uint32_t sz = r.u32(); if (r.eof) return ORC_ERR_EOF; if (sz != cbs) return ORC_ERR_ARG; if (!r.raw(&cb[m * cbs], cbs * 4)) return ORC_ERR_EOF; } }
    uint32_t n = r.u32(); if (r.eof) return ORC_ERR_EOF;
    std::vector<uint32_t> ids(n); std::vector<uint8_t> codes((size_t)n * h->M);
    for (uint32_t i = 0; i < n; i++) { ids[i] = r.u32(); if (!r.raw(&codes[(size_t)i * h->M], (size_t)h->M)) return ORC_ERR_EOF; }
    if (int rc = r.empty_bitmap()) return rc;
    h->trained = tr; if (tr) h->codebooks.swap(cb); h->ids.swap(ids); h->codes.swap(codes); h->deleted.clear();
    return r.off;
}

// IVFPQIndex.Flush ivfpq_index.go:401-439
ORC_API void orc_ivfpq_flush(void* p) {
    auto* h = (OIVFPQ*)p; if (h->deleted.empty()) return;
    for (int l = 0; l < h->nlist; l++) {
        std::vector<uint32_t> ids; std::vector<uint8_t> codes;
        for (size_t j = 0; j < h->list_ids[l].size(); j++) if (!h->deleted.count(h->list_ids[l][j])) {
            ids.push_back(h->list_ids[l][j]); codes.insert(codes.end(), h->list_codes[l].begin() + j * h->M, h->list_codes[l].begin() + (j + 1) * h->M);
        }
        h->list_ids[l].swap(ids); h->list_codes[l].swap(codes);
    }
    h->deleted.clear();
}
ORC_API int64_t orc_ivfpq_write(void* p, uint8_t* out, int64_t cap) {
    auto* h = (OIVFPQ*)p; orc_ivfpq_flush(p);
    OW w; w.raw("IVPQ", 4); w.u32(1); w.u32((uint32_t)h->dim); w.kind(h->metric); w.u32((uint32_t)h->nlist);
    w.u32((uint32_t)h->M); w.u32((uint32_t)h->nbits); w.u32((uint32_t)h->Ksub); w.u32((uint32_t)h->dsub); w.u8(h->trained ? 1 : 0);
    const size_t cbs = (size_t)h->Ksub * h->dsub;
    if (h->trained) {
        for (int l = 0; l < h->nlist; l++) { w.u32((uint32_t)h->dim); w.raw(&h->centroids[(size_t)l * h->dim], (size_t)h->dim * 4); }
        for (int m = 0; m < h->M; m++) { w.u32((uint32_t)cbs); w.raw(&h->codebooks[m * cbs], cbs * 4); }
    }
    w.u32((uint32_t)h->nlist);
    for (int l = 0; l < h->nlist; l++) {
        w.u32((uint32_t)h->list_ids[l].size());
        for (size_t j = 0; j < h->list_ids[l].size(); j++) { w.u32(h->list_ids[l][j]); w.raw(&h->list_codes[l][j * h->M], (size_t)h->M); }
    }
    w.empty_bitmap();
    return w.finish(out, cap);
}
ORC_API int64_t orc_ivfpq_read(void* p, const uint8_t* in, int64_t len) {
    auto* h = (OIVFPQ*)p; OR r(in, len);
    if (int rc = r.header("IVPQ", h->dim, h->metric)) return rc;
    uint32_t nl = r.u32(), M = r.u32(), nb = r.u32(), K = r.u32(), ds = r.u32(); if (r.eof) return ORC_ERR_EOF;
    if ((int)nl != h->nlist || (int)M != h->M || (int)nb != h->nbits || (int)K != h->Ksub || (int)ds != h->dsub) return ORC_ERR_ARG;
    bool tr = r.u8() == 1; if (r.eof) return ORC_ERR_EOF;
    const size_t cbs = (size_t)h->Ksub * h->dsub;
    std::vector<float> cen, cb;
    if (tr) {
        cen.resize((size_t)h->nlist * h->dim);
        for (int l = 0; l < h->nlist; l++) { uint32_t cs = r.u32(); if (r.eof) return ORC_ERR_EOF; if ((int)cs != h->dim) return ORC_ERR_DIM; if (!r.raw(&cen[(size_t)l * h->dim], (size_t)h->dim * 4)) return ORC_ERR_EOF; }
        cb.resize((size_t)h->M * cbs);
        for (int m = 0; m < h->M; m++) { uint32_t sz = r.u32(); if (r.eof) return ORC_ERR_EOF; if (sz != cbs) return ORC_ERR_ARG; if (!r.raw(&cb[m * cbs], cbs * 4)) return ORC_ERR_EOF; }
    }
    uint32_t lc = r.u32(); if (r.eof) return ORC_ERR_EOF; if ((int)lc != h->nlist) return ORC_ERR_ARG;
    std::vector<std::vector<uint32_t>> li(lc); std::vector<std::vector<uint8_t>> lcodes(lc);
    for (uint32_t l = 0; l < lc; l++) {
        uint32_t n = r.u32(); if (r.eof) return ORC_ERR_EOF;
        li[l].resize(n); lcodes[l].resize((size_t)n * h->M);
        for (uint32_t j = 0; j < n; j++) { li[l][j] = r.u32(); if (!r.raw(&lcodes[l][(size_t)j * h->M], (size_t)h->M)) return ORC_ERR_EOF; }
    }
    if (int rc = r.empty_bitmap()) return rc;
    h->trained = tr; if (tr) { h->centroids.swap(cen); h->codebooks.swap(cb); } h->list_ids.swap(li); h->list_codes.swap(lcodes); h->deleted.clear();
    return r.off;
}

static std::vector<uint32_t> hnsw_sorted_ids(OHNSW* h) { std::vector<uint32_t> v; for (auto& kv : h->nodes) v.push_back(kv.first); std::sort(v.begin(), v.end()); return v; }
// HNSWIndex.Flush hnsw_index.go:348-431 (map iteration replaced by ascending id order)
ORC_API void orc_hnsw_flush(void* p) {
    auto* h = (OHNSW*)p; if (h->deleted.empty()) return;
    const auto order = hnsw_sorted_ids(h);
    for (uint32_t id : order) {                                                  // PHASE 1
        if (h->deleted.count(id)) continue;
        ONode* n = h->nodes[id];
        for (auto& e : n->edges) { std::vector<uint32_t> f; for (uint32_t x : e) if (!h->deleted.count(x)) f.push_back(x); e.swap(f); }
    }
    if (h->deleted.count(h->entry)) {                                            // PHASE 2
        bool found = false;
        for (uint32_t id : order) { ONode* n = h->nodes[id]; if (!h->deleted.count(id) && n->level == h->max_level) { h->entry = id; found = true; break; } }
        if (!found) {
            int best = -1;
            for (uint32_t id : order) { ONode* n = h->nodes[id]; if (!h->deleted.count(id) && n->level > best) { best = n->level; h->entry = id; } }
            if (best >= 0) h->max_level = best; else { h->entry = 0; h->max_level = -1; }
        }
    }
    for (uint32_t id : h->deleted) { auto it = h->nodes.find(id); if (it != h->nodes.end()) { delete it->second; h->nodes.erase(it); } }   // PHASE 3
    std::vector<uint32_t> io; for (uint32_t id : h->insertion_order) if (h->nodes.count(id)) io.push_back(id);
    h->insertion_order.swap(io);
    h->deleted.clear();                                                          // PHASE 4
}
ORC_API int64_t orc_hnsw_write(void* p, uint8_t* out, int64_t cap) {
    auto* h = (OHNSW*)p; orc_hnsw_flush(p);
    OW w; w.raw("HNSW", 4); w.u32(1); w.u32((uint32_t)h->dim); w.kind(h->metric);
    w.u32((uint32_t)h->M); w.u32((uint32_t)h->efC); w.u32((uint32_t)h->efS);
    w.f64(1.0 / go_log((double)h->M));                                           // levelMult hnsw_index.go:206
    w.i32(h->max_level); w.u32(h->entry); w.u32((uint32_t)h->nodes.size());
    for (uint32_t id : hnsw_sorted_ids(h)) {
        ONode* n = h->nodes[id];
        w.u32(id); w.i32(n->level); w.u32((uint32_t)n->vec.size()); w.raw(n->vec.data(), n->vec.size() * 4);
        w.u32((uint32_t)n->edges.size());
        for (auto& e : n->edges) { w.u32((uint32_t)e.size()); if (!e.empty()) w.raw(e.data(), e.size() * 4); }
    }
    w.empty_bitmap();
    return w.finish(out, cap);
}
ORC_API int64_t orc_hnsw_read(void* p, const uint8_t* in, int64_t len) {
    auto* h = (OHNSW*)p; OR r(in, len);
    if (int rc = r.header("HNSW", h->dim, h->metric)) return rc;
    uint32_t M = r.u32(), efc = r.u32(), efs = r.u32(); if (r.eof) return ORC_ERR_EOF;
    if ((int)M != h->M || (int)efc != h->efC || (int)efs != h->efS) return ORC_ERR_ARG;   // :975-986
    (void)r.f64(); int32_t maxl = r.i32(); uint32_t ep = r.u32(); uint32_t n = r.u32(); if (r.eof) return ORC_ERR_EOF;
    std::unordered_map<uint32_t, ONode*> nodes; std::vector<uint32_t> order;
    auto fail = [&](int rc) { for (auto& kv : nodes) delete kv.second; return (int64_t)rc; };
    for (uint32_t i = 0; i < n; i++) {
        ONode* nd = new ONode(); nd->id = r.u32(); nd->level = r.i32(); uint32_t vd = r.u32();
        if (r.eof || vd > (1u << 20)) { delete nd; return fail(r.eof ? ORC_ERR_EOF : ORC_ERR_ARG); }
        nd->vec.resize(vd); r.raw(nd->vec.data(), (size_t)vd * 4);
        uint32_t layers = r.u32(); if (r.eof || layers > 64) { delete nd; return fail(r.eof ? ORC_ERR_EOF : ORC_ERR_ARG); }
        nd->edges.resize(layers);
        for (uint32_t l = 0; l < layers; l++) { uint32_t ec = r.u32(); if (r.eof || ec > (1u << 24)) { delete nd; return fail(r.eof ? ORC_ERR_EOF : ORC_ERR_ARG); } nd->edges[l].resize(ec); r.raw(nd->edges[l].data(), (size_t)ec * 4); }
        if (r.eof) { delete nd; return fail(ORC_ERR_EOF); }
        auto it = nodes.find(nd->id); if (it != nodes.end()) delete it->second;
        nodes[nd->id] = nd; order.push_back(nd->id);
    }
    if (int rc = r.empty_bitmap()) return fail(rc);
    for (auto& kv : h->nodes) delete kv.second;
    h->nodes.swap(nodes); h->insertion_order.swap(order); h->max_level = maxl; h->entry = ep; h->deleted.clear();
    return r.off;
}
