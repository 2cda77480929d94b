/* The headline step driven from C through the ABI, the way a cgo caller would (no Python, no torch): Flat cosine 1M x 768 (rows and queries generated on the device with
 * the bench's own seeds), batch 256, K 100, device-resident queries and results, `depth` searches in flight through comet_index_search_dev_async / _wait; regions of `steps`
 * steps bracketed by comet_ctx_sync, wall clock. An independent cross-check of bench.py's `value` / `single_stream` figures (same library, same kernels, another host).
 *   gcc -O2 -std=c11 -I include tools/c_bench.c -o tools/c_bench -L comet_amd -lcomet_hip -Wl,-rpath,$PWD/comet_amd -lm && tools/c_bench [rows [steps [regions [max in flight]]]] */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "comet_gpu.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "FAIL %s -> %d: %s\n", #x, rc_, comet_last_error()); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int cmpd(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

enum { DIM = 768, B = 256, K = 100, NQB = 8, CHUNK = 65536 };

static double region(comet_index* idx, float** q, uint32_t** oi, float** os, int32_t** oc, int nbuf, int depth, int steps, comet_ctx* ctx, int* qi) {
    comet_search_params p; memset(&p, 0, sizeof(p)); p.k = K;
    uint64_t fly[8]; int nf = 0;
    CK(comet_ctx_sync(ctx));
    const double t0 = now();
    for (int s = 0; s < steps; s++) {
        const int w = *qi % nbuf; float* qq = q[*qi % NQB]; (*qi)++;
        uint64_t t = 0; CK(comet_index_search_dev_async(idx, qq, B, &p, oi[w], os[w], oc[w], K, &t));
        fly[nf++] = t;
        if (nf >= depth) { CK(comet_index_search_wait(idx, fly[0])); memmove(fly, fly + 1, sizeof(uint64_t) * (size_t)(--nf)); }
    }
    for (int j = 0; j < nf; j++) CK(comet_index_search_wait(idx, fly[j]));
    CK(comet_ctx_sync(ctx));
    return now() - t0;
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 1000000; const int steps = argc > 2 ? atoi(argv[2]) : 20; const int regions = argc > 3 ? atoi(argv[3]) : 7;
    comet_ctx* ctx = NULL; if (comet_ctx_create(0, &ctx) != 0) { fprintf(stderr, "c_bench: %s\n", comet_last_error()); return 77; }
    comet_index* idx = NULL; CK(comet_flat_create(ctx, DIM, COMET_COSINE, &idx));
    void *buf = NULL, *idbuf = NULL; CK(comet_dev_alloc(ctx, (size_t)CHUNK * DIM * 4, &buf)); CK(comet_dev_alloc(ctx, (size_t)CHUNK * 4, &idbuf));
    uint32_t* ids = (uint32_t*)malloc((size_t)CHUNK * 4);
    const double tb = now();
    for (long lo = 0; lo < rows; lo += CHUNK) {
        const long m = rows - lo < CHUNK ? rows - lo : CHUNK;
        CK(comet_synth_fill_dev(ctx, 0xC0FFEE + 2, (uint64_t)lo * DIM, (uint64_t)m * DIM, (float*)buf));            /* bench.py: CORPUS_SEED */
        for (long i = 0; i < m; i++) ids[i] = (uint32_t)(lo + i + 1);
        CK(comet_memcpy_h2d(ctx, idbuf, ids, (size_t)m * 4));
        int64_t added = 0; CK(comet_index_add_dev(idx, (const uint32_t*)idbuf, (const float*)buf, m, &added));
        if (added != m) { fprintf(stderr, "added %ld of %ld\n", (long)added, m); return 1; }
    }
    CK(comet_ctx_sync(ctx));
    const double build_s = now() - tb;
    float* q[NQB]; void* qbase = NULL; CK(comet_dev_alloc(ctx, (size_t)NQB * B * DIM * 4, &qbase));
    for (int i = 0; i < NQB; i++) { q[i] = (float*)qbase + (size_t)i * B * DIM; CK(comet_synth_fill_dev(ctx, 0xBEEF + 2, (uint64_t)i * B * DIM, (uint64_t)B * DIM, q[i])); }   /* QUERY_SEED */
    CK(comet_ctx_sync(ctx));
    uint32_t* oi[5]; float* os[5]; int32_t* oc[5];
    for (int w = 0; w < 5; w++) { void* p = NULL; CK(comet_dev_alloc(ctx, (size_t)B * K * 4, &p)); oi[w] = p; CK(comet_dev_alloc(ctx, (size_t)B * K * 4, &p)); os[w] = p; CK(comet_dev_alloc(ctx, (size_t)B * 4, &p)); oc[w] = p; }
    int qi = 0;
    (void)region(idx, q, oi, os, oc, 3, 2, 25, ctx, &qi);                                                              /* warm-up: shadows, scratch, clocks */
    printf("{\"harness\": \"tools/c_bench.c (C over the ABI, no Python)\", \"workload\": \"Flat cosine %ldx%d, batch %d, K %d\", \"build_s\": %.2f, \"steps\": %d, \"regions\": %d", rows, DIM, B, K, build_s, steps, regions);
    const int maxdepth = argc > 4 ? atoi(argv[4]) : 2;                    /* searches in flight: 2 = bench.py's Flat leg; up to 4 here (a Flat index rotates through two lanes) */
    for (int depth = maxdepth > 4 ? 4 : maxdepth; depth >= 1; depth--) {
        double t[32]; const int R = regions > 32 ? 32 : regions;
        for (int r = 0; r < R; r++) t[r] = region(idx, q, oi, os, oc, depth + 1, depth, steps, ctx, &qi);
        qsort(t, (size_t)R, sizeof(double), cmpd);
        const double med = t[R / 2];
        printf(", \"in_flight_%d\": {\"qps\": %.0f, \"ms_per_step\": %.4f, \"best_region_qps\": %.0f}", depth, (double)B * steps / med, med / steps * 1e3, (double)B * steps / t[0]);
    }
    const double t0 = now(); int n = 0; while (now() - t0 < 2.0) { (void)region(idx, q, oi, os, oc, 3, 2, 200, ctx, &qi); n += 200; }
    printf(", \"sustained_two_in_flight_qps\": %.0f", (double)B * n / (now() - t0));
    int32_t cnt[B]; CK(comet_memcpy_d2h(ctx, cnt, oc[0], sizeof(cnt))); long tot = 0; for (int i = 0; i < B; i++) tot += cnt[i];
    printf(", \"results_per_query\": %.1f}\n", (double)tot / B);
    comet_index_destroy(idx); comet_ctx_destroy(ctx);
    return 0;
}
