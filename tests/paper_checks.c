/* Known-answer checks on the GPU through the C ABI, without Python (starts in a second; tests/test_paper_checks_gpu.py runs it under pytest, -m gpu):
 *   hnsw     the hand-traced HNSW case of tests/golden/paper_kats.json (hnsw_dim2_m2_efc3): nine nodes inserted by hnsw_insert_kernel with the given levels, the exported
 *            graph (14 edge lists, order included), entry point, maxLevel and six searches against the values written out on paper, L2^2 and Euclidean;
 *   filters  the reference's document-filter test tables (tests/golden/reference_kats_r06.json: flat / ivf / pq / ivfpq / hnsw / bm25 _index_document_filter_test.go);
 *   searches the behaviours of hnsw_index_search_test.go:123-330,646-852 held in the same file;
 *   lifecycle training preconditions, Add / search before Train, zero vectors under cosine, soft delete / Flush bookkeeping and searches over deleted rows for the five
 *            vector kinds, HNSW's Flush of the entry point and of every node (the *_index_test.go files) — status codes and the reference's messages.
 * The expected values are copied from those fixtures. Exit 0 and one "... OK" line per check, 1 and the first difference, 77 when there is no gfx950 device.
 *   gcc -O1 -std=c11 -I include tests/paper_checks.c -o tests/paper_checks -L comet_amd -lcomet_hip -Wl,-rpath,$PWD/comet_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "comet_gpu.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { printf("FAIL %s:%d %s -> %d: %s\n", __func__, __LINE__, #x, rc_, comet_last_error()); return 1; } } while (0)
#define EXPECT(c, ...) do { if (!(c)) { printf("FAIL %s:%d ", __func__, __LINE__); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)

static comet_ctx* ctx;

/* ------------------------------------------------------------------------------------------------ the hand-traced HNSW case */
static const float HV[9][2] = {{5, 8}, {2, 6}, {9, 8}, {0, 1}, {3, 4}, {1, 1}, {0, 5}, {6, 1}, {6, 7}};
static const int32_t HLV[9] = {2, 0, 1, 0, 1, 0, 1, 0, 0};
static const int HG[][6] = {                            /* (node, layer) in node-major order, layer 0 first; -1 ends a list */
    {2, 3, 5, 4, -1}, {3, 5, -1}, {-1}, {5, 1, 4, 3, -1}, {1, 2, 4, 9, -1}, {1, 5, -1}, {5, 2, 1, 3, -1},
    {2, 6, 4, 1, -1}, {1, 3, -1}, {4, 5, 2, 8, -1}, {2, 5, 4, -1}, {5, 1, -1}, {5, 6, 4, -1}, {1, 3, 2, -1}};
struct HQ { float q[2]; int k, ef, n; uint32_t ids[8]; float d2[8]; };
static const struct HQ HQS[6] = {
    {{1, 2}, 3, 0, 3, {6, 4, 5}, {1, 2, 8}}, {{8, 8}, 2, 0, 2, {3, 9}, {1, 5}}, {{8, 8}, 0, 8, 8, {3, 9, 1, 2, 5, 8, 6, 4}, {1, 5, 9, 40, 41, 53, 98, 113}},
    {{1, 5}, 4, 0, 3, {2, 5, 6}, {2, 5, 16}}, {{1, 2}, 1, 1, 1, {6}, {1}}, {{1, 5}, 0, 9, 8, {2, 5, 6, 4, 1, 9, 8, 3}, {2, 5, 16, 17, 25, 29, 41, 73}}};

static int hnsw_paper(int metric) {
    comet_index* h = NULL;
    CK(comet_hnsw_create(ctx, 2, metric, 2, 3, 3, &h));
    uint32_t ids[9]; for (int i = 0; i < 9; i++) ids[i] = (uint32_t)(i + 1);
    int64_t added = 0;
    CK(comet_hnsw_add_with_levels(h, ids, &HV[0][0], HLV, 5, &added));             /* two batches: state carries over between launches */
    CK(comet_hnsw_add_with_levels(h, ids + 5, &HV[5][0], HLV + 5, 4, &added));
    int64_t n = 0, slots = 0, ne = 0; uint32_t entry = 0; int32_t maxl = -9;
    CK(comet_hnsw_export_graph(h, &n, &slots, &ne, NULL, NULL, NULL, NULL, NULL, &entry, &maxl));
    EXPECT(n == 9 && slots == 14 && entry == 1 && maxl == 2, "shape: n %ld slots %ld edges %ld entry %u maxLevel %d", (long)n, (long)slots, (long)ne, entry, maxl);
    uint32_t gi[9]; int32_t gl[9]; float gv[18]; int64_t off[15]; uint32_t ed[64];
    EXPECT(ne < 64, "%ld edges", (long)ne);
    CK(comet_hnsw_export_graph(h, &n, &slots, &ne, gi, gl, gv, off, ed, &entry, &maxl));
    int s = 0;
    for (int i = 0; i < 9; i++) {
        EXPECT(gi[i] == ids[i] && gl[i] == HLV[i], "node %d: id %u level %d", i, gi[i], gl[i]);
        for (int l = 0; l <= HLV[i]; l++, s++) {
            int len = 0; while (HG[s][len] >= 0) len++;
            EXPECT(off[s + 1] - off[s] == len, "node %u layer %d: %ld edges, the paper says %d", gi[i], l, (long)(off[s + 1] - off[s]), len);
            for (int e = 0; e < len; e++) EXPECT((int)ed[off[s] + e] == HG[s][e], "node %u layer %d edge %d: %u, the paper says %d", gi[i], l, e, ed[off[s] + e], HG[s][e]);
        }
    }
    for (int qi = 0; qi < 6; qi++) {
        const struct HQ* q = &HQS[qi];
        comet_search_params p; memset(&p, 0, sizeof(p)); p.k = q->k; p.ef_search = q->ef;
        uint32_t oi[16]; float os[16]; int32_t cnt = -1;
        CK(comet_index_search(h, q->q, 1, &p, oi, os, &cnt, 16));
        EXPECT(cnt == q->n, "query %d: %d results, the paper says %d", qi, cnt, q->n);
        for (int j = 0; j < q->n; j++) {
            const float want = metric == COMET_L2SQ ? q->d2[j] : (float)sqrt((double)q->d2[j]);
            EXPECT(oi[j] == q->ids[j] && memcmp(&os[j], &want, 4) == 0, "query %d result %d: id %u score %.9g, the paper says %u / %.9g", qi, j, oi[j], os[j], q->ids[j], want);
        }
    }
    CK(comet_index_destroy(h));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ helpers for the reference's tables */
static const float U3[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}};
static void ramp3(int i, float* v) { v[0] = v[1] = v[2] = 0; v[i % 3] = (float)i; }                  /* the tests' vec[i%3] = float32(i) */

/* one search of one query; the ids returned, sorted ascending, into got[]; returns the count (or -1000 - rc) */
static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }
static int ids_of(comet_index* h, const float* q, int k, float thr, int nprobes, int ef, const uint32_t* flt, int nf, uint32_t* got, float* scores) {
    comet_search_params p; memset(&p, 0, sizeof(p)); p.k = k; p.threshold = thr; p.nprobes = nprobes; p.ef_search = ef; p.filter_ids = nf ? flt : NULL; p.n_filter = nf;
    float sc[32]; int32_t cnt = -1;
    int rc = comet_index_search(h, q, 1, &p, got, sc, &cnt, 32);
    if (rc != 0) return -1000 - rc;
    if (scores) memcpy(scores, sc, sizeof(float) * (cnt > 0 ? (cnt < 32 ? cnt : 32) : 0));
    for (int i = 1; i < cnt; i++) if (sc[i] < sc[i - 1]) return -2000;                               /* ascending distances */
    if (cnt > 0 && !scores) qsort(got, (size_t)cnt, 4, cmp_u32);
    return cnt;
}
static int subset(const uint32_t* got, int n, const uint32_t* want, int nw) {
    for (int i = 0; i < n; i++) { int f = 0; for (int j = 0; j < nw; j++) f |= got[i] == want[j]; if (!f) return 0; }
    return 1;
}
struct Case { int nf; uint32_t flt[4]; int nw; uint32_t want[6]; };
static const struct Case T5[5] = {{0, {0}, 6, {1, 2, 3, 4, 5, 6}}, {3, {1, 3, 5}, 3, {1, 3, 5}}, {1, {2}, 1, {2}}, {2, {100, 200}, 0, {0}}, {0, {0}, 6, {1, 2, 3, 4, 5, 6}}};
static const uint32_t ID6[6] = {1, 2, 3, 4, 5, 6};

static int add_rows(comet_index* h, const uint32_t* ids, const float* v, int n) { int64_t a = 0; CK(comet_index_add(h, ids, v, n, &a, NULL)); EXPECT(a == n, "%ld of %d rows added", (long)a, n); return 0; }

static int filters(void) {
    uint32_t got[32]; float sc[32]; comet_index* h = NULL;
    const float q3[3] = {1, 0, 0}; const float q8[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    float U8[6][8]; memset(U8, 0, sizeof(U8)); for (int i = 0; i < 6; i++) memcpy(U8[i], U3[i], 12);
    float slope[20][8]; for (int i = 0; i < 20; i++) for (int j = 0; j < 8; j++) slope[i][j] = (float)(i + j);
    /* flat_index_document_filter_test.go:10-91: exact id sets */
    CK(comet_flat_create(ctx, 3, COMET_L2, &h)); if (add_rows(h, ID6, &U3[0][0], 6)) return 1;
    for (int c = 0; c < 5; c++) {
        int n = ids_of(h, q3, 10, 0, 0, 0, T5[c].flt, T5[c].nf, got, NULL);
        EXPECT(n == T5[c].nw && memcmp(got, T5[c].want, 4 * (size_t)n) == 0, "flat case %d: %d results", c, n);
    }
    CK(comet_index_destroy(h));
    /* :94-131 ten ramp rows, two queries, only even ids */
    { CK(comet_flat_create(ctx, 3, COMET_L2, &h)); uint32_t ids[10]; float v[10][3]; for (int i = 0; i < 10; i++) { ids[i] = (uint32_t)(i + 1); ramp3(i + 1, v[i]); }
      if (add_rows(h, ids, &v[0][0], 10)) return 1;
      const uint32_t even[4] = {2, 4, 6, 8}; const float qs[2][3] = {{1, 0, 0}, {0, 1, 0}};
      for (int b = 0; b < 2; b++) { int n = ids_of(h, qs[b], 10, 0, 0, 0, even, 4, got, NULL); EXPECT(n == 4 && subset(got, n, even, 4), "flat multi-query %d: %d results", b, n); }
      CK(comet_index_destroy(h)); }
    /* :134-181 filter {1,2,3} and threshold 1.5 -> ids 1, 2 at distances 0, 1 */
    { CK(comet_flat_create(ctx, 3, COMET_L2, &h)); const uint32_t ids[4] = {1, 2, 3, 4}; const float v[4][3] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {10, 0, 0}};
      if (add_rows(h, ids, &v[0][0], 4)) return 1;
      const uint32_t f[3] = {1, 2, 3}; int n = ids_of(h, q3, 10, 1.5f, 0, 0, f, 3, got, sc);
      EXPECT(n == 2 && got[0] == 1 && got[1] == 2 && sc[0] == 0.0f && sc[1] == 1.0f, "flat threshold: %d results", n);
      CK(comet_index_destroy(h)); }
    /* ivf_index_document_filter_test.go:8-112 */
    { CK(comet_ivf_create(ctx, 3, COMET_L2, 2, &h)); const float tr[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}};
      CK(comet_index_train(h, &tr[0][0], 4)); if (add_rows(h, ID6, &U3[0][0], 6)) return 1;
      for (int c = 0; c < 4; c++) { int n = ids_of(h, q3, 10, 0, 2, 0, T5[c].flt, T5[c].nf, got, NULL); EXPECT(n == T5[c].nw && subset(got, n, T5[c].want, T5[c].nw), "ivf case %d: %d results", c, n); }
      CK(comet_index_destroy(h)); }
    /* :115-196 nlist 4, twelve ramp training rows, twenty ramp rows, filter {1..5}, nprobes 1 / 2 / 4 */
    { CK(comet_ivf_create(ctx, 3, COMET_L2, 4, &h)); float tr[12][3]; for (int i = 0; i < 12; i++) ramp3(i, tr[i]);
      CK(comet_index_train(h, &tr[0][0], 12)); uint32_t ids[20]; float v[20][3]; for (int i = 0; i < 20; i++) { ids[i] = (uint32_t)(i + 1); ramp3(i + 1, v[i]); }
      if (add_rows(h, ids, &v[0][0], 20)) return 1;
      const uint32_t f[5] = {1, 2, 3, 4, 5}; const int nps[3] = {1, 2, 4}; int last = 0;
      for (int t = 0; t < 3; t++) { int n = ids_of(h, q3, 10, 0, nps[t], 0, f, 5, got, NULL); EXPECT(n >= last && n <= 5 && subset(got, n, f, 5), "ivf nprobes %d: %d results", nps[t], n); last = n; }
      EXPECT(last == 5, "ivf nprobes 4 (every list): %d of the five filtered rows", last);
      CK(comet_index_destroy(h)); }
    /* pq_index_document_filter_test.go:8-110 and ivfpq_index_document_filter_test.go:8-116: dim 8, M 2, nbits 4, twenty slope training rows */
    for (int ivf = 0; ivf < 2; ivf++) {
        if (ivf) CK(comet_ivfpq_create(ctx, 8, COMET_L2, 2, 2, 4, &h)); else CK(comet_pq_create(ctx, 8, COMET_L2, 2, 4, &h));
        CK(comet_index_train(h, &slope[0][0], 20)); if (add_rows(h, ID6, &U8[0][0], 6)) return 1;
        for (int c = 0; c < 4; c++) { int n = ids_of(h, q8, 10, 0, ivf ? 2 : 0, 0, T5[c].flt, T5[c].nf, got, NULL); EXPECT(n == T5[c].nw && subset(got, n, T5[c].want, T5[c].nw), "%s case %d: %d results", ivf ? "ivfpq" : "pq", c, n); }
        CK(comet_index_destroy(h));
    }
    /* hnsw_index_document_filter_test.go:11-85 (exact sets), :88-124 (efSearch 50), :127-178 (a soft-deleted node named by the filter) */
    { CK(comet_hnsw_create(ctx, 3, COMET_L2, 16, 200, 100, &h)); CK(comet_hnsw_set_level_seed(h, 12345)); if (add_rows(h, ID6, &U3[0][0], 6)) return 1;
      for (int c = 0; c < 4; c++) { int n = ids_of(h, q3, 10, 0, 0, 0, T5[c].flt, T5[c].nf, got, NULL); EXPECT(n == T5[c].nw && memcmp(got, T5[c].want, 4 * (size_t)n) == 0, "hnsw case %d: %d results", c, n); }
      CK(comet_index_destroy(h));
      CK(comet_hnsw_create(ctx, 3, COMET_L2, 16, 200, 100, &h)); CK(comet_hnsw_set_level_seed(h, 7)); uint32_t ids[20]; float v[20][3]; for (int i = 0; i < 20; i++) { ids[i] = (uint32_t)(i + 1); ramp3(i + 1, v[i]); }
      if (add_rows(h, ids, &v[0][0], 20)) return 1;
      const uint32_t f[6] = {2, 4, 6, 8, 10, 12}; int n = ids_of(h, q3, 5, 0, 0, 50, f, 6, got, NULL); EXPECT(n == 5 && subset(got, n, f, 6), "hnsw efSearch 50: %d results", n);
      CK(comet_index_destroy(h));
      CK(comet_hnsw_create(ctx, 3, COMET_L2, 16, 200, 100, &h)); CK(comet_hnsw_set_level_seed(h, 7)); if (add_rows(h, ID6, &U3[0][0], 4)) return 1;
      CK(comet_index_remove(h, 2)); const uint32_t f2[3] = {1, 2, 3}; n = ids_of(h, q3, 10, 0, 0, 0, f2, 3, got, NULL); EXPECT(n == 2 && got[0] == 1 && got[1] == 3, "hnsw after deletion: %d results", n);
      CK(comet_index_destroy(h)); }
    /* bm25_index_document_filter_test.go:8-122: five documents, token ids per distinct word */
    { comet_text_index* t = NULL; CK(comet_bm25_create(ctx, &t));
      /* the 1, quick 2, brown 3, fox 4, jumps 5, over 6, lazy 7, dog 8, cat 9, sleeps 10, all 11, day 12, movements 13, of 14, barks 15, at 16, strangers 17, a 18, in 19, forest 20 */
      const uint32_t d1[] = {1, 2, 3, 4, 5, 6, 1, 7, 8}, d2[] = {1, 7, 9, 10, 11, 12}, d3[] = {2, 13, 14, 1, 4}, d4[] = {1, 8, 15, 16, 17}, d5[] = {18, 4, 19, 1, 20};
      CK(comet_bm25_add(t, 1, d1, 9)); CK(comet_bm25_add(t, 2, d2, 6)); CK(comet_bm25_add(t, 3, d3, 5)); CK(comet_bm25_add(t, 4, d4, 5)); CK(comet_bm25_add(t, 5, d5, 5));
      struct { uint32_t tok; int nf; uint32_t flt[2]; int nw; uint32_t want[3]; } cs[5] = {{4, 0, {0}, 3, {1, 3, 5}}, {4, 2, {1, 3}, 2, {1, 3}}, {4, 1, {5}, 1, {5}}, {4, 2, {2, 4}, 0, {0}}, {7, 0, {0}, 2, {1, 2}}};
      for (int c = 0; c < 5; c++) {
          const int32_t qo[2] = {0, 1}; uint32_t oi[16]; float os[16]; double os64[16]; int32_t cnt = -1;
          CK(comet_bm25_search(t, &cs[c].tok, qo, 1, 10, cs[c].nf ? cs[c].flt : NULL, cs[c].nf, oi, os, os64, &cnt, 16));
          EXPECT(cnt == cs[c].nw, "bm25 case %d: %d results", c, cnt);
          for (int i = 1; i < cnt; i++) EXPECT(os64[i] <= os64[i - 1] && os64[i] > 0, "bm25 case %d: scores not descending", c);
          qsort(oi, (size_t)cnt, 4, cmp_u32); EXPECT(memcmp(oi, cs[c].want, 4 * (size_t)cnt) == 0, "bm25 case %d: ids", c);
      }
      CK(comet_bm25_destroy(t)); }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ hnsw_index_search_test.go:123-330,646-852 */
static int hnsw_with(int metric, const float (*rows)[3], int n, comet_index** out) {
    CK(comet_hnsw_create(ctx, 3, metric, 16, 200, 200, out)); CK(comet_hnsw_set_level_seed(*out, 12345));
    uint32_t ids[8]; for (int i = 0; i < n; i++) ids[i] = (uint32_t)(i + 1);
    return n ? add_rows(*out, ids, &rows[0][0], n) : 0;
}
static int searches(void) {
    uint32_t got[32]; float sc[32]; comet_index* h = NULL; int n;
    const float q100[3] = {1, 0, 0};
    { const float r[5][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {2, 0, 0}}; if (hnsw_with(COMET_L2, r, 5, &h)) return 1;
      n = ids_of(h, q100, 2, 0, 0, 0, NULL, 0, got, sc); EXPECT(n == 2 && got[0] == 1 && sc[0] == 0.0f, "Simple: %d results, first %u", n, got[0]); CK(comet_index_destroy(h)); }
    { const float r[3][3] = {{1, 2, 3}, {4, 5, 6}, {7, 8, 9}}; const float q[3] = {4, 5, 6}; if (hnsw_with(COMET_L2, r, 3, &h)) return 1;
      n = ids_of(h, q, 1, 0, 0, 0, NULL, 0, got, sc); EXPECT(n == 1 && got[0] == 2 && sc[0] == 0.0f, "ExactMatch: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; if (hnsw_with(COMET_L2, r, 3, &h)) return 1;
      n = ids_of(h, q100, 10, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n == 3, "KGreaterThanSize: %d results", n);
      n = ids_of(h, q100, 2, 0, 0, -1, NULL, 0, got, sc); EXPECT(n == 2 && got[0] == 1, "efSearch -1: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[4][3] = {{1, 0, 0}, {2, 0, 0}, {4, 0, 0}, {10, 0, 0}}; if (hnsw_with(COMET_L2, r, 4, &h)) return 1;
      n = ids_of(h, q100, 10, 2.0f, 0, 0, NULL, 0, got, sc); EXPECT(n == 2 && got[0] == 1 && got[1] == 2 && sc[1] == 1.0f, "WithThreshold: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[3][3] = {{10, 10, 10}, {20, 20, 20}, {30, 30, 30}}; const float q[3] = {0, 0, 0}; if (hnsw_with(COMET_L2, r, 3, &h)) return 1;
      n = ids_of(h, q, 10, 1.0f, 0, 0, NULL, 0, got, NULL); EXPECT(n == 0, "ThresholdStrictFiltering: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[1][3] = {{1, 2, 3}}; const float z[3] = {0, 0, 0}; if (hnsw_with(COMET_COSINE, r, 1, &h)) return 1;
      n = ids_of(h, z, 1, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n < 0, "ZeroVectorCosine: a zero query under cosine must fail, got %d", n); CK(comet_index_destroy(h)); }
    { const float q[3] = {1, 2, 3}; if (hnsw_with(COMET_L2, NULL, 0, &h)) return 1;
      n = ids_of(h, q, 10, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n == 0, "Empty: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[2][3] = {{1, 0, 0}, {2, 0, 0}}; if (hnsw_with(COMET_L2, r, 2, &h)) return 1;
      CK(comet_index_remove(h, 1)); CK(comet_index_remove(h, 2)); CK(comet_index_flush(h));
      n = ids_of(h, q100, 10, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n == 0, "AfterAllDeleted: %d results", n); CK(comet_index_destroy(h)); }
    { const float r[1][3] = {{1, 2, 3}}; const float q[3] = {1, 2, 3}; if (hnsw_with(COMET_L2, r, 1, &h)) return 1;
      n = ids_of(h, q, 1, 0, 0, 0, NULL, 0, got, sc); EXPECT(n == 1 && got[0] == 1, "SingleNode: %d results", n); CK(comet_index_destroy(h)); }
    for (int m = COMET_L2; m <= COMET_COSINE; m++) {
        const float r[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}}; if (hnsw_with(m, r, 4, &h)) return 1;
        n = ids_of(h, q100, 2, 0, 0, 0, NULL, 0, got, sc); EXPECT(n == 2 && got[0] == 1, "DifferentMetrics %d: %d results", m, n); CK(comet_index_destroy(h));
    }
    { const float r[3][3] = {{1, 0, 0}, {1, 1, 0}, {0, 1, 0}}; const float q[3] = {2, 0, 0}; if (hnsw_with(COMET_COSINE, r, 3, &h)) return 1;
      n = ids_of(h, q, 1, 0, 0, 0, NULL, 0, got, sc); EXPECT(n == 1 && got[0] == 1 && sc[0] == 0.0f, "Cosine: %d results, first %u at %.9g", n, got[0], sc[0]); CK(comet_index_destroy(h)); }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ lifecycle behaviours of the reference's *_index_test.go files
 * (flat_index_test.go:188-313,343-435; ivf_index_test.go:97-132,204-378; pq_index_test.go:159-227,319-531; ivfpq_index_test.go:150-204,297-506; hnsw_index_test.go:443-672):
 * status codes AND the reference's messages (ivf_index.go:212,257; pq_index.go:199,269; ivfpq_index.go:186,285; *_index_search.go "must be trained before searching" /
 * "index not trained"; distance.go ErrZeroVector; flat_index.go:233,236) */
#define FAILS(call, code, text) do { int rc_ = (call); EXPECT(rc_ == (code) && strstr(comet_last_error(), (text)) != NULL, "%s -> %d \"%s\", expected %d \"%s\"", #call, rc_, comet_last_error(), (int)(code), (text)); } while (0)
static float R8[100][8], S8[100][8];
static int make_kind(int kind, int metric, comet_index** h) {
    const float off = metric == COMET_COSINE ? 1.0f : 0.0f;                          /* (no zero row in a cosine training set) */
    for (int i = 0; i < 100; i++) for (int j = 0; j < 8; j++) { R8[i][j] = (float)(i * 8 + j) + off; S8[i][j] = (float)((i * 8 + j) % 10) + off; }
    switch (kind) {
    case 0: CK(comet_flat_create(ctx, 8, metric, h)); break;
    case 1: CK(comet_ivf_create(ctx, 8, metric, 2, h)); CK(comet_index_train(*h, &R8[0][0], 100)); break;
    case 2: CK(comet_pq_create(ctx, 8, metric, 4, 6, h)); CK(comet_index_train(*h, &S8[0][0], 100)); break;
    case 3: CK(comet_ivfpq_create(ctx, 8, metric, 2, 4, 4, h)); CK(comet_index_train(*h, &R8[0][0], 100)); break;
    default: CK(comet_hnsw_create(ctx, 8, metric, 16, 200, 200, h)); CK(comet_hnsw_set_level_seed(*h, 99)); break;
    }
    return 0;
}
static int lifecycle(void) {
    static const char* NAME[5] = {"flat", "ivf", "pq", "ivfpq", "hnsw"};
    uint32_t got[32]; comet_index* h = NULL; int64_t added = -1; int n;
    const float e8[8] = {1, 0, 0, 0, 0, 0, 0, 0}, z8[8] = {0}; const uint32_t one = 1, two = 2;
    /* training preconditions, Add / search before Train */
    { float five3[5][3]; for (int i = 0; i < 5; i++) { five3[i][0] = (float)i; five3[i][1] = five3[i][2] = 0; }
      CK(comet_ivf_create(ctx, 3, COMET_L2, 10, &h)); FAILS(comet_index_train(h, &five3[0][0], 5), COMET_ERR_TRAIN_DATA, "need at least 10 training vectors for 10 clusters (got 5)"); CK(comet_index_destroy(h));
      float five8[5][8]; memset(five8, 0, sizeof(five8)); for (int i = 0; i < 5; i++) five8[i][0] = (float)i;
      CK(comet_ivfpq_create(ctx, 8, COMET_L2, 10, 4, 4, &h)); FAILS(comet_index_train(h, &five8[0][0], 5), COMET_ERR_TRAIN_DATA, "need at least 100 vectors for training"); CK(comet_index_destroy(h));
      float same[100][8]; for (int i = 0; i < 100; i++) for (int j = 0; j < 8; j++) same[i][j] = (float)i;
      CK(comet_pq_create(ctx, 8, COMET_L2, 4, 8, &h)); FAILS(comet_index_train(h, &same[0][0], 100), COMET_ERR_TRAIN_DATA, "need at least 256 vectors for training"); CK(comet_index_destroy(h));
      const float q3[3] = {1, 0, 0};
      CK(comet_ivf_create(ctx, 3, COMET_L2, 2, &h)); FAILS(comet_index_add(h, &one, q3, 1, &added, NULL), COMET_ERR_NOT_TRAINED, "index must be trained before adding vectors");
      EXPECT(ids_of(h, q3, 5, 0, 1, 0, NULL, 0, got, NULL) == -1000 - COMET_ERR_NOT_TRAINED && strstr(comet_last_error(), "index must be trained before searching"), "ivf search before train: %s", comet_last_error()); CK(comet_index_destroy(h));
      CK(comet_pq_create(ctx, 8, COMET_L2, 4, 6, &h)); FAILS(comet_index_add(h, &one, e8, 1, &added, NULL), COMET_ERR_NOT_TRAINED, "index must be trained before adding");
      EXPECT(ids_of(h, e8, 5, 0, 0, 0, NULL, 0, got, NULL) == -1000 - COMET_ERR_NOT_TRAINED && strstr(comet_last_error(), "index not trained"), "pq search before train: %s", comet_last_error()); CK(comet_index_destroy(h));
      CK(comet_ivfpq_create(ctx, 8, COMET_L2, 2, 4, 4, &h)); FAILS(comet_index_add(h, &one, e8, 1, &added, NULL), COMET_ERR_NOT_TRAINED, "index must be trained before adding");
      EXPECT(ids_of(h, e8, 5, 0, 1, 0, NULL, 0, got, NULL) == -1000 - COMET_ERR_NOT_TRAINED && strstr(comet_last_error(), "index must be trained before searching"), "ivfpq search before train: %s", comet_last_error()); CK(comet_index_destroy(h)); }
    for (int kind = 0; kind < 5; kind++) {
        /* a zero vector under cosine stops the batch at that row (ErrZeroVector); Euclidean takes it */
        if (make_kind(kind, COMET_COSINE, &h)) return 1;
        FAILS(comet_index_add(h, &one, z8, 1, &added, NULL), COMET_ERR_ZERO_VECTOR, "zero vector not allowed for this metric"); EXPECT(added == 0, "%s: %ld rows added before the zero vector", NAME[kind], (long)added);
        CK(comet_index_add(h, &two, e8, 1, &added, NULL)); EXPECT(added == 1 && comet_index_size(h) == 1, "%s: size %ld after one good row", NAME[kind], (long)comet_index_size(h)); CK(comet_index_destroy(h));
        /* soft delete, Flush, searches over deleted rows */
        if (make_kind(kind, COMET_L2, &h)) return 1;
        const uint32_t ids[4] = {11, 12, 13, 14}; float rows[4][8]; memset(rows, 0, sizeof(rows)); for (int i = 0; i < 4; i++) rows[i][0] = (float)(i + 1);
        if (add_rows(h, ids, &rows[0][0], 4)) return 1;
        const float q[8] = {1.5f, 0, 0, 0, 0, 0, 0, 0}; const int np = (kind == 1 || kind == 3) ? 2 : 0;
        n = ids_of(h, q, 10, 0, np, 0, NULL, 0, got, NULL); EXPECT(n == 4 && got[0] == 11 && got[3] == 14, "%s: %d results before any delete", NAME[kind], n);
        CK(comet_index_remove(h, 12)); CK(comet_index_remove(h, 13)); EXPECT(comet_index_size(h) == 4, "%s: size %ld after two soft deletes", NAME[kind], (long)comet_index_size(h));
        FAILS(comet_index_remove(h, 12), COMET_ERR_ALREADY_DELETED, "12 already deleted"); FAILS(comet_index_remove(h, 9999), COMET_ERR_NOT_FOUND, "9999 not found");
        n = ids_of(h, q, 10, 0, np, 0, NULL, 0, got, NULL); EXPECT(n == 2 && got[0] == 11 && got[1] == 14, "%s: %d results after two soft deletes", NAME[kind], n);
        const uint32_t f[3] = {11, 12, 13}; n = ids_of(h, q, 10, 0, np, 0, f, 3, got, NULL); EXPECT(n == 1 && got[0] == 11, "%s: %d results with a filter naming deleted rows", NAME[kind], n);
        CK(comet_index_flush(h)); EXPECT(comet_index_size(h) == 2, "%s: size %ld after Flush", NAME[kind], (long)comet_index_size(h));
        n = ids_of(h, q, 10, 0, np, 0, NULL, 0, got, NULL); EXPECT(n == 2 && got[0] == 11 && got[1] == 14, "%s: %d results after Flush", NAME[kind], n);
        FAILS(comet_index_remove(h, 12), COMET_ERR_NOT_FOUND, "12 not found"); CK(comet_index_flush(h)); EXPECT(comet_index_size(h) == 2, "%s: size after a second Flush", NAME[kind]);
        CK(comet_index_remove(h, 11)); CK(comet_index_remove(h, 14)); CK(comet_index_flush(h)); EXPECT(comet_index_size(h) == 0, "%s: size %ld after flushing everything", NAME[kind], (long)comet_index_size(h));
        n = ids_of(h, q, 10, 0, np, 0, NULL, 0, got, NULL); EXPECT(n == 0, "%s: %d results from the emptied index", NAME[kind], n);
        CK(comet_index_destroy(h));
    }
    /* hnsw_index_test.go:586-629: the entry point removed and flushed -> another node takes over; :631-672: everything flushed -> entry 0, maxLevel -1 */
    { CK(comet_hnsw_create(ctx, 3, COMET_L2, 16, 200, 200, &h)); CK(comet_hnsw_set_level_seed(h, 3));
      uint32_t ids[5]; float v[5][3]; memset(v, 0, sizeof(v)); for (int i = 0; i < 5; i++) { ids[i] = (uint32_t)(i + 1); v[i][0] = (float)i; }
      if (add_rows(h, ids, &v[0][0], 5)) return 1;
      int64_t nn = 0, slots = 0, ne = 0; uint32_t entry = 99; int32_t maxl = -9;
      CK(comet_hnsw_export_graph(h, &nn, &slots, &ne, NULL, NULL, NULL, NULL, NULL, &entry, &maxl)); EXPECT(entry == 1 && nn == 5, "entry %u of %ld nodes", entry, (long)nn);
      CK(comet_index_remove(h, 1)); CK(comet_index_flush(h));
      CK(comet_hnsw_export_graph(h, &nn, &slots, &ne, NULL, NULL, NULL, NULL, NULL, &entry, &maxl)); EXPECT(entry != 1 && entry != 0 && nn == 4, "after flushing the entry point: entry %u, %ld nodes", entry, (long)nn);
      const float q0[3] = {0, 0, 0}; n = ids_of(h, q0, 10, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n == 4 && got[0] == 2 && got[3] == 5, "after flushing the entry point: %d results", n);
      for (uint32_t i = 2; i <= 5; i++) CK(comet_index_remove(h, i));
      CK(comet_index_flush(h));
      CK(comet_hnsw_export_graph(h, &nn, &slots, &ne, NULL, NULL, NULL, NULL, NULL, &entry, &maxl)); EXPECT(nn == 0 && entry == 0 && maxl == -1, "after flushing everything: %ld nodes, entry %u, maxLevel %d", (long)nn, entry, maxl);
      n = ids_of(h, q0, 10, 0, 0, 0, NULL, 0, got, NULL); EXPECT(n == 0, "emptied HNSW: %d results", n); CK(comet_index_destroy(h)); }
    return 0;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    if (comet_ctx_create(0, &ctx) != 0) { fprintf(stderr, "paper_checks: %s\n", comet_last_error()); return 77; }
    int all = !strcmp(what, "all");
    if (all || !strcmp(what, "hnsw")) { if (hnsw_paper(COMET_L2SQ) || hnsw_paper(COMET_L2)) return 1; printf("paper HNSW case OK (graph, entry point, maxLevel, six searches; L2^2 and Euclidean)\n"); }
    if (all || !strcmp(what, "filters")) { if (filters()) return 1; printf("document-filter tables OK (flat, ivf, pq, ivfpq, hnsw, bm25)\n"); }
    if (all || !strcmp(what, "searches")) { if (searches()) return 1; printf("HNSW search behaviours OK\n"); }
    if (all || !strcmp(what, "lifecycle")) { if (lifecycle()) return 1; printf("lifecycle behaviours OK (training preconditions, zero vectors, soft delete / Flush for five kinds, HNSW entry-point flush; codes and messages)\n"); }
    comet_ctx_destroy(ctx);
    return 0;
}
