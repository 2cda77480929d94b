/* abi_harness.c — drives libcomet_hip.so through include/comet_gpu.h exactly the way the cgo shim (go/cometgpu) does:
 * plain C, no Python, caller-allocated buffers, callbacks for io.Writer / io.Reader, pthreads for concurrent Execute calls.
 * Built by __graft_entry__.build() (gcc -pthread); run by tests/test_abi_harness.py (GPU) — exit code 0 = every check passed,
 * 77 = no gfx950 device (what the CPU-only container gets), anything else = a failed check (message on stderr).
 * Expected values are computed here with the reference's own arithmetic (sequential float32 sums, distance.go:158-165). */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "comet_gpu.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "abi_harness: %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [last error: %s]\n", comet_last_error()); exit(1); } } while (0)
#define OK(call) do { int _rc = (call); CHECK(_rc == COMET_OK, "%s -> %d", #call, _rc); } while (0)

enum { DIM = 24, N = 5000, NQ = 16, K = 10, THREADS = 8 };
static float X[N][DIM], Q[NQ][DIM];
static uint32_t want_ids[NQ][K];
static float want_sc[NQ][K];

static uint64_t rng_state = 0x1234;
static float frand(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }

static float l2sq(const float* a, const float* b) {   /* distance.go:158-165: sequential float32, product rounded before the add */
    volatile float sum = 0.0f;
    for (int i = 0; i < DIM; i++) { volatile float d = a[i] - b[i]; volatile float sq = d * d; sum = sum + sq; }
    return sum;
}
static void reference_topk(const float* q, int n, uint32_t* ids, float* sc) {   /* flat_index_search.go:221-294 with the canonical tie order */
    for (int k = 0; k < K; k++) { ids[k] = 0; sc[k] = INFINITY; }
    for (int i = 0; i < n; i++) {
        float d = l2sq(q, X[i]);
        int pos = K;
        while (pos > 0 && d < sc[pos - 1]) pos--;
        if (pos < K) { for (int j = K - 1; j > pos; j--) { sc[j] = sc[j - 1]; ids[j] = ids[j - 1]; } sc[pos] = d; ids[pos] = (uint32_t)(i + 1); }
    }
}

struct membuf { uint8_t* p; size_t len, cap, off; };
static int mem_write(void* u, const void* d, size_t n) { struct membuf* b = u; if (b->len + n > b->cap) { b->cap = (b->len + n) * 2; b->p = realloc(b->p, b->cap); } memcpy(b->p + b->len, d, n); b->len += n; return 0; }
static int mem_read(void* u, void* d, size_t n) { struct membuf* b = u; if (b->off + n > b->len) return 1; memcpy(d, b->p + b->off, n); b->off += n; return 0; }

struct worker { comet_index* idx; int tid; int failures; };
static void* search_worker(void* arg) {
    struct worker* w = arg;
    comet_search_params p; memset(&p, 0, sizeof(p)); p.k = K;
    for (int rep = 0; rep < 20; rep++) {
        int q = (w->tid * 7 + rep) % NQ;
        uint32_t ids[K]; float sc[K]; int32_t cnt = 0;
        if (comet_index_search(w->idx, Q[q], 1, &p, ids, sc, &cnt, K) != COMET_OK) { w->failures++; continue; }   /* one query per call = the Go Execute() shape */
        if (cnt != K || memcmp(ids, want_ids[q], sizeof(ids)) != 0 || memcmp(sc, want_sc[q], sizeof(sc)) != 0) w->failures++;
    }
    return NULL;
}
static void* add_worker(void* arg) {          /* vectors far outside the data: never in anybody's top-K */
    struct worker* w = arg;
    for (int i = 0; i < 40; i++) {
        float v[DIM]; for (int j = 0; j < DIM; j++) v[j] = 100.0f + (float)i;
        uint32_t id = 1000000u + (uint32_t)i; int64_t added = 0;
        if (comet_index_add(w->idx, &id, v, 1, &added, NULL) != COMET_OK || added != 1) w->failures++;
    }
    return NULL;
}

int main(void) {
    int ndev = 0;
    comet_device_count(&ndev);
    comet_ctx* ctx = NULL;
    int rc = comet_ctx_create(0, &ctx);
    if (rc == COMET_ERR_NO_DEVICE) { fprintf(stderr, "abi_harness: no gfx950 device (%s)\n", comet_last_error()); return 77; }
    CHECK(rc == COMET_OK, "comet_ctx_create -> %d", rc);
    for (int i = 0; i < N; i++) for (int j = 0; j < DIM; j++) X[i][j] = frand();
    for (int i = 0; i < NQ; i++) for (int j = 0; j < DIM; j++) Q[i][j] = frand();

    /* NewFlatIndex(24, L2Squared); Add one vector per call like idx.Add(*NewVectorNodeWithID(id, v)) for the first 100, then a batch */
    comet_index* idx = NULL;
    OK(comet_flat_create(ctx, DIM, COMET_L2SQ, &idx));
    for (int i = 0; i < 100; i++) { uint32_t id = (uint32_t)(i + 1); int64_t added = 0; OK(comet_index_add(idx, &id, X[i], 1, &added, NULL)); CHECK(added == 1, "added %lld", (long long)added); }
    { static uint32_t ids[N]; for (int i = 100; i < N; i++) ids[i] = (uint32_t)(i + 1); int64_t added = 0; OK(comet_index_add(idx, ids + 100, X[100], N - 100, &added, NULL)); CHECK(added == N - 100, "batch added %lld", (long long)added); }
    CHECK(comet_index_size(idx) == N && comet_index_kind(idx) == COMET_KIND_FLAT && comet_index_dim(idx) == DIM && comet_index_trained(idx) == 1, "introspection");

    /* batched search == NQ reference searches, bit for bit */
    for (int q = 0; q < NQ; q++) reference_topk(Q[q], N, want_ids[q], want_sc[q]);
    comet_search_params p; memset(&p, 0, sizeof(p)); p.k = K;
    static uint32_t ids[NQ * K]; static float sc[NQ * K]; int32_t cnt[NQ];
    for (int mode = 0; mode <= 1; mode++) {
        p.mode = mode;
        OK(comet_index_search(idx, &Q[0][0], NQ, &p, ids, sc, cnt, K));
        for (int q = 0; q < NQ; q++) CHECK(cnt[q] == K && memcmp(ids + q * K, want_ids[q], K * 4) == 0 && memcmp(sc + q * K, want_sc[q], K * 4) == 0, "query %d differs from the reference arithmetic (mode %d)", q, mode);
    }
    p.mode = 0;
    /* WithThreshold / WithDocumentIDs / k <= 0 */
    p.threshold = want_sc[0][4]; OK(comet_index_search(idx, Q[0], 1, &p, ids, sc, cnt, K)); CHECK(cnt[0] == 5, "threshold kept %d", cnt[0]); p.threshold = 0;
    { uint32_t flt[3] = {want_ids[0][7], want_ids[0][2], 4999999u}; p.filter_ids = flt; p.n_filter = 3; OK(comet_index_search(idx, Q[0], 1, &p, ids, sc, cnt, K));
      CHECK(cnt[0] == 2 && ids[0] == want_ids[0][2] && ids[1] == want_ids[0][7], "filter"); p.filter_ids = NULL; p.n_filter = 0; }
    /* errors mirror the reference: soft delete twice, unknown id, cosine zero vector */
    OK(comet_index_remove(idx, want_ids[0][0]));
    CHECK(comet_index_remove(idx, want_ids[0][0]) == COMET_ERR_ALREADY_DELETED && strstr(comet_last_error(), "already deleted"), "double remove");
    CHECK(comet_index_remove(idx, 77777777u) == COMET_ERR_NOT_FOUND && strstr(comet_last_error(), "not found"), "remove unknown");
    OK(comet_index_search(idx, Q[0], 1, &p, ids, sc, cnt, K)); CHECK(ids[0] == want_ids[0][1], "soft-deleted row still returned");
    { comet_index* c = NULL; OK(comet_flat_create(ctx, 4, COMET_COSINE, &c)); float z[4] = {0, 0, 0, 0}; uint32_t id = 1; int64_t added = 0;
      CHECK(comet_index_add(c, &id, z, 1, &added, NULL) == COMET_ERR_ZERO_VECTOR && added == 0, "zero vector accepted");
      float v[4] = {3, 4, 0, 0}, n[4]; OK(comet_index_add(c, &id, v, 1, &added, n)); CHECK(fabsf(n[0] - 0.6f) < 1e-6f && fabsf(n[1] - 0.8f) < 1e-6f, "normalised write-back (flat_index.go:182)");
      OK(comet_index_destroy(c)); }
    { comet_index* bad = NULL; CHECK(comet_flat_create(ctx, 0, COMET_L2, &bad) == COMET_ERR_INVALID_ARG && bad == NULL, "dimension 0 accepted");
      CHECK(comet_flat_create(ctx, 4, 9, &bad) == COMET_ERR_UNKNOWN_METRIC, "unknown metric accepted"); }

    /* io.WriterTo / io.ReaderFrom through callbacks: flushes (the removed id disappears), round-trips, rejects a wrong magic */
    struct membuf wb = {0};
    int64_t nbytes = 0;
    OK(comet_index_write_to(idx, mem_write, &wb, &nbytes));
    CHECK((size_t)nbytes == wb.len && memcmp(wb.p, "FLAT", 4) == 0 && comet_index_size(idx) == N - 1, "write_to: %lld bytes", (long long)nbytes);
    comet_index* idx2 = NULL;
    OK(comet_flat_create(ctx, DIM, COMET_L2SQ, &idx2));
    OK(comet_index_read_from(idx2, mem_read, &wb, &nbytes));
    CHECK((size_t)nbytes == wb.len && comet_index_size(idx2) == N - 1, "read_from consumed %lld of %zu", (long long)nbytes, wb.len);
    OK(comet_index_search(idx2, Q[0], 1, &p, ids, sc, cnt, K)); CHECK(ids[0] == want_ids[0][1] && sc[0] == want_sc[0][1], "reloaded index answers differently");
    { comet_index* iv = NULL; OK(comet_ivf_create(ctx, DIM, COMET_L2SQ, 4, &iv)); wb.off = 0;
      CHECK(comet_index_read_from(iv, mem_read, &wb, &nbytes) == COMET_ERR_FORMAT && strstr(comet_last_error(), "invalid magic number: expected 'IVFX', got 'FLAT'"), "wrong magic accepted");
      OK(comet_index_destroy(iv)); }
    OK(comet_index_destroy(idx2));

    /* concurrent Execute()s racing Adds (flat_index_search_test.go:392-465): restore the removed row's absence in the expectation */
    for (int q = 0; q < NQ; q++) {   /* expectations for the flushed index: recompute without the removed id */
        uint32_t removed = want_ids[0][0];
        uint32_t t_ids[K + 1]; float t_sc[K + 1]; int m = 0;
        float best_d[K + 1]; (void)best_d;
        /* simple re-evaluation: top-(K) over rows != removed */
        for (int k = 0; k < K; k++) { t_ids[k] = 0; t_sc[k] = INFINITY; }
        for (int i = 0; i < N; i++) {
            if ((uint32_t)(i + 1) == removed) continue;
            float d = l2sq(Q[q], X[i]); int pos = K;
            while (pos > 0 && d < t_sc[pos - 1]) pos--;
            if (pos < K) { for (int j = K - 1; j > pos; j--) { t_sc[j] = t_sc[j - 1]; t_ids[j] = t_ids[j - 1]; } t_sc[pos] = d; t_ids[pos] = (uint32_t)(i + 1); }
        }
        (void)m;
        memcpy(want_ids[q], t_ids, sizeof(want_ids[q])); memcpy(want_sc[q], t_sc, sizeof(want_sc[q]));
    }
    pthread_t th[THREADS + 1]; struct worker ws[THREADS + 1];
    for (int t = 0; t <= THREADS; t++) { ws[t].idx = idx; ws[t].tid = t; ws[t].failures = 0; pthread_create(&th[t], NULL, t < THREADS ? search_worker : add_worker, &ws[t]); }
    int failures = 0;
    for (int t = 0; t <= THREADS; t++) { pthread_join(th[t], NULL); failures += ws[t].failures; }
    CHECK(failures == 0, "%d concurrent calls failed or returned different rows", failures);
    CHECK(comet_index_size(idx) == N - 1 + 40, "size after concurrent adds: %lld", (long long)comet_index_size(idx));

    OK(comet_index_destroy(idx));
    /* HNSW: built on the GPU with given levels, serialised with the two-call protocol, read back into a fresh index: the copy
     * answers the same (hnsw_index.go:228-288 Add, :701-1096 WriteTo / ReadFrom). */
    {
        enum { HN = 600 };
        comet_index *h = NULL, *h2 = NULL;
        OK(comet_hnsw_create(ctx, DIM, COMET_L2SQ, 8, 40, 40, &h));
        static uint32_t hid[HN]; static int32_t hlv[HN];
        for (int i = 0; i < HN; i++) { hid[i] = (uint32_t)(i + 1); hlv[i] = (i % 9 == 0) ? 1 : ((i % 50 == 0) ? 2 : 0); }
        int64_t added = 0;
        OK(comet_hnsw_add_with_levels(h, hid, &X[0][0], hlv, HN, &added)); CHECK(added == HN && comet_index_size(h) == HN, "hnsw add_with_levels: %lld", (long long)added);
        CHECK(comet_index_add(h, hid, &X[0][0], 1, &added, NULL) != COMET_OK, "re-adding a live HNSW id accepted");
        comet_search_params hp; memset(&hp, 0, sizeof(hp)); hp.k = 3; hp.ef_search = 64;
        uint32_t a_ids[3], b_ids[3]; float a_sc[3], b_sc[3]; int32_t a_cnt = 0, b_cnt = 0;
        OK(comet_index_search(h, X[17], 1, &hp, a_ids, a_sc, &a_cnt, 3));
        CHECK(a_cnt == 3 && a_sc[0] <= a_sc[1] && a_sc[1] <= a_sc[2] && a_ids[0] >= 1 && a_ids[0] <= HN, "hnsw query: %d results, first id %u", a_cnt, a_ids[0]);   /* (the reference's graphs have poor recall: no claim about WHICH ids) */
        size_t need = 0, used = 0;
        OK(comet_index_serialize(h, NULL, 0, &need)); CHECK(need > 4, "hnsw image size");
        uint8_t* img = malloc(need);
        OK(comet_index_serialize(h, img, need, &used)); CHECK(used == need && memcmp(img, "HNSW", 4) == 0, "hnsw image: %zu of %zu bytes", used, need);
        OK(comet_hnsw_create(ctx, DIM, COMET_L2SQ, 8, 40, 40, &h2));
        OK(comet_index_deserialize(h2, img, need, &used)); CHECK(used == need && comet_index_size(h2) == HN, "hnsw image consumed %zu of %zu", used, need);
        OK(comet_index_search(h2, X[17], 1, &hp, b_ids, b_sc, &b_cnt, 3));
        CHECK(b_cnt == a_cnt && memcmp(a_ids, b_ids, sizeof(a_ids)) == 0 && memcmp(a_sc, b_sc, sizeof(a_sc)) == 0, "hnsw copy answers differently");
        CHECK(comet_index_deserialize(h2, img, need / 2, &used) == COMET_ERR_FORMAT || comet_index_deserialize(h2, img, need / 2, &used) == COMET_ERR_IO, "truncated hnsw image accepted");
        free(img);
        OK(comet_index_destroy(h)); OK(comet_index_destroy(h2));
    }
    /* IVFPQ (ivfpq_index.go:113 New, :180 Train, :279 Add, :544 WriteTo; ivfpq_index_search.go:231 search): trained and filled on the GPU;
     * the pruned (default) search and the every-candidate search (mode 1) return the same rows; the IVPQ image round-trips. */
    {
        enum { PN = 3000, PK = 8 };
        comet_index* pq = NULL;
        OK(comet_ivfpq_create(ctx, DIM, COMET_L2SQ, 8, 4, 8, &pq));
        CHECK(comet_index_trained(pq) == 0, "untrained index reports trained");
        { comet_search_params sp0; memset(&sp0, 0, sizeof(sp0)); sp0.k = 3; uint32_t i3[3]; float s3[3]; int32_t c3;
          CHECK(comet_index_search(pq, Q[0], 1, &sp0, i3, s3, &c3, 3) == COMET_ERR_NOT_TRAINED, "search before Train accepted"); }
        OK(comet_index_train(pq, &X[0][0], 2000));
        static uint32_t pid[PN]; for (int i = 0; i < PN; i++) pid[i] = (uint32_t)(i + 1);
        int64_t added = 0; OK(comet_index_add(pq, pid, &X[0][0], PN, &added, NULL)); CHECK(added == PN && comet_index_size(pq) == PN, "ivfpq add: %lld", (long long)added);
        comet_search_params sp; memset(&sp, 0, sizeof(sp)); sp.k = PK; sp.nprobes = 4;
        static uint32_t a_ids[NQ * PK], b_ids[NQ * PK]; static float a_sc[NQ * PK], b_sc[NQ * PK]; int32_t a_cn[NQ], b_cn[NQ];
        OK(comet_index_search(pq, &Q[0][0], NQ, &sp, a_ids, a_sc, a_cn, PK));
        sp.mode = 1; OK(comet_index_search(pq, &Q[0][0], NQ, &sp, b_ids, b_sc, b_cn, PK)); sp.mode = 0;
        for (int q = 0; q < NQ; q++) {
            CHECK(a_cn[q] == PK && b_cn[q] == PK, "ivfpq counts %d / %d", a_cn[q], b_cn[q]);
            CHECK(memcmp(a_ids + q * PK, b_ids + q * PK, PK * 4) == 0 && memcmp(a_sc + q * PK, b_sc + q * PK, PK * 4) == 0, "ivfpq: pruned and every-candidate search differ (query %d)", q);
            for (int i = 1; i < PK; i++) CHECK(a_sc[q * PK + i - 1] <= a_sc[q * PK + i], "ivfpq scores not ascending");
        }
        size_t need = 0, used = 0;
        OK(comet_index_serialize(pq, NULL, 0, &need));
        unsigned char* img = malloc(need);
        OK(comet_index_serialize(pq, img, need, &used)); CHECK(used == need && memcmp(img, "IVPQ", 4) == 0, "ivfpq image");
        comet_index* pq2 = NULL; OK(comet_ivfpq_create(ctx, DIM, COMET_L2SQ, 8, 4, 8, &pq2));
        OK(comet_index_deserialize(pq2, img, need, &used)); CHECK(used == need && comet_index_size(pq2) == PN && comet_index_trained(pq2) == 1, "ivfpq image consumed %zu of %zu", used, need);
        OK(comet_index_search(pq2, &Q[0][0], NQ, &sp, b_ids, b_sc, b_cn, PK));
        CHECK(memcmp(a_ids, b_ids, sizeof(a_ids)) == 0 && memcmp(a_sc, b_sc, sizeof(a_sc)) == 0, "reloaded ivfpq index answers differently");
        free(img);
        OK(comet_index_destroy(pq)); OK(comet_index_destroy(pq2));
    }
    /* BM25 (bm25_index.go:143 New, :188 Add, :253 Remove; bm25_index_search.go:143 Execute): token ids in, scores descending, a removed
     * document disappears, a document holding the query token twice outranks one holding it once (same length). */
    {
        comet_text_index* tx = NULL;
        OK(comet_bm25_create(ctx, &tx));
        const uint32_t d1[4] = {7, 7, 3, 4}, d2[4] = {7, 5, 3, 4}, d3[4] = {9, 5, 3, 4}, d4[2] = {1, 2};
        OK(comet_bm25_add(tx, 11, d1, 4)); OK(comet_bm25_add(tx, 12, d2, 4)); OK(comet_bm25_add(tx, 13, d3, 4)); OK(comet_bm25_add(tx, 14, d4, 2));
        CHECK(comet_bm25_num_docs(tx) == 4 && fabs(comet_bm25_avg_doc_len(tx) - 3.5) < 1e-12, "bm25 stats");
        const uint32_t qt[1] = {7}; const int32_t qo[2] = {0, 1};
        uint32_t t_ids[4]; float t_sc[4]; double t_sc64[4]; int32_t t_cn = -1;
        OK(comet_bm25_search(tx, qt, qo, 1, 10, NULL, 0, t_ids, t_sc, t_sc64, &t_cn, 4));
        CHECK(t_cn == 2 && t_ids[0] == 11 && t_ids[1] == 12 && t_sc64[0] > t_sc64[1] && t_sc64[1] > 0.0, "bm25 ranking: %d results, first %u", t_cn, t_ids[0]);
        OK(comet_bm25_remove(tx, 11));
        OK(comet_bm25_search(tx, qt, qo, 1, 10, NULL, 0, t_ids, t_sc, t_sc64, &t_cn, 4));
        CHECK(t_cn == 1 && t_ids[0] == 12, "bm25 after remove: %d results", t_cn);
        OK(comet_bm25_destroy(tx));
    }
    /* segment layer (storage.go:489-626, vector leg): three Flat segments; id 5 re-written in the newest one. The fused call returns,
     * per query, the highest score per id over the per-segment top-k's, scores DESCENDING, cut to k (storage_merge.go:13-54). */
    {
        enum { SK = 6 };
        comet_index* sg[3]; const int lo[4] = {0, 700, 1500, N};
        for (int s = 0; s < 3; s++) {
            OK(comet_flat_create(ctx, DIM, COMET_L2SQ, &sg[s]));
            static uint32_t sid[N]; for (int i = lo[s]; i < lo[s + 1]; i++) sid[i] = (uint32_t)(i + 1);
            if (s == 2) sid[lo[2]] = 5u;                                   /* the row at lo[2] carries document id 5 again */
            int64_t added = 0; OK(comet_index_add(sg[s], sid + lo[s], X[lo[s]], lo[s + 1] - lo[s], &added, NULL)); CHECK(added == lo[s + 1] - lo[s], "segment add");
        }
        comet_search_params sp; memset(&sp, 0, sizeof(sp)); sp.k = SK;
        uint32_t g_ids[SK]; float g_sc[SK]; int32_t g_cnt = -1;
        OK(comet_segments_search((comet_index* const*)sg, 3, Q[1], 1, &sp, g_ids, g_sc, &g_cnt, SK));
        /* expectation from the per-segment searches through the single-index entry point */
        uint32_t u_ids[3 * SK]; float u_sc[3 * SK]; int un = 0;
        for (int s = 0; s < 3; s++) {
            uint32_t t_ids[SK]; float t_sc[SK]; int32_t t_cnt = 0;
            OK(comet_index_search(sg[s], Q[1], 1, &sp, t_ids, t_sc, &t_cnt, SK));
            for (int i = 0; i < t_cnt; i++) {
                int at = -1; for (int j = 0; j < un; j++) if (u_ids[j] == t_ids[i]) at = j;
                if (at < 0) { u_ids[un] = t_ids[i]; u_sc[un] = t_sc[i]; un++; } else if (t_sc[i] > u_sc[at]) u_sc[at] = t_sc[i];
            }
        }
        for (int i = 0; i < un; i++) for (int j = i + 1; j < un; j++)      /* score descending, equal scores: ascending id */
            if (u_sc[j] > u_sc[i] || (u_sc[j] == u_sc[i] && u_ids[j] < u_ids[i])) { uint32_t ti = u_ids[i]; u_ids[i] = u_ids[j]; u_ids[j] = ti; float ts = u_sc[i]; u_sc[i] = u_sc[j]; u_sc[j] = ts; }
        CHECK(g_cnt == SK && un >= SK && memcmp(g_ids, u_ids, sizeof(g_ids)) == 0 && memcmp(g_sc, u_sc, sizeof(g_sc)) == 0, "segments search differs from the merged per-segment searches (count %d)", g_cnt);
        sp.k = -1; CHECK(comet_segments_search((comet_index* const*)sg, 3, Q[1], 1, &sp, g_ids, g_sc, &g_cnt, SK) == COMET_ERR_INVALID_ARG, "negative k accepted");
        for (int s = 0; s < 3; s++) OK(comet_index_destroy(sg[s]));
    }
    OK(comet_ctx_destroy(ctx));
    free(wb.p);
    printf("abi_harness OK: create/add/search/filter/threshold/remove/errors/write_to/read_from/%d-thread concurrency/hnsw build+serialize/ivfpq train+search+serialize/bm25/segment fan-out through the C ABI\n", THREADS);
    return 0;
}
