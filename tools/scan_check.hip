// scan_check.hip — stand-alone check + timing of the Flat fast-path scan kernels (kernels_fast.hip is compiled into this program).
// For every COMET_SCAN_VARIANT it runs the scan on random data and compares the emitted unit keys (two smallest approximate
// distances + the bound of every (query, 128-row unit)) with a float64 host computation on the same fp16-rounded operands.
// usage: scan_check [rows] [dim] [queries] [iters]        build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DQR_TRACE] tools/scan_check.hip -o tools/scan_check
#define COMET_SCAN_QR_RUNTIME_SWITCH 1
#include "../comet_amd/csrc/kernels_fast.hip"
#include "../comet_amd/csrc/kernels_scanq.hip"
#include "../comet_amd/csrc/kernels_scanq_l2.hip"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <random>

using namespace comet;

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 5000;
    const int dim = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 256, iters = argc > 4 ? atoi(argv[4]) : 0;
    const int ld = padded_dim(dim), ldh = (int)round_up(dim, 64);
    Ctx c; c.device = 0;
    HIP_CHECK(hipSetDevice(0)); HIP_CHECK(hipGetDeviceProperties(&c.prop, 0)); HIP_CHECK(hipStreamCreate(&c.stream));
    std::mt19937 rng(7); std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> X((size_t)n * ld, 0.f), Q((size_t)256 * ld, 0.f);
    for (long i = 0; i < n; i++) { double s = 0; for (int j = 0; j < dim; j++) { X[i * ld + j] = U(rng); s += X[i * ld + j] * X[i * ld + j]; } for (int j = 0; j < dim; j++) X[i * ld + j] /= (float)std::sqrt(s); }
    for (int i = 0; i < B; i++) { double s = 0; for (int j = 0; j < dim; j++) { Q[i * ld + j] = U(rng); s += Q[i * ld + j] * Q[i * ld + j]; } for (int j = 0; j < dim; j++) Q[i * ld + j] /= (float)std::sqrt(s); }
    if (argc > 5) { const float sc = (float)atof(argv[5]); for (auto& v : X) v *= sc; }      // argv[5] = 0: zero corpus (clock / power experiment)
    float *dX, *dQ, *rn, *qn, *err; void *Xh, *Qh; uint32_t* stats; int32_t* st4;
    const int unit = (B <= 64 || (getenv("SCAN_UNIT") && atoi(getenv("SCAN_UNIT")) == 64)) ? 64 : 128;          // the narrow tiles (<= 64 queries) emit per 64-row unit; SCAN_UNIT=64: the wide tile with 64-row units (variant 0 only)
    const long tiles = ceil_div(n, 256), units = tiles * (256 / unit), ldS = round_up(2 * units, 16), ldB = round_up(units, 16);
    HIP_CHECK(hipMalloc(&dX, X.size() * 4)); HIP_CHECK(hipMalloc(&dQ, Q.size() * 4)); HIP_CHECK(hipMalloc(&rn, n * 4)); HIP_CHECK(hipMalloc(&qn, 1024)); HIP_CHECK(hipMalloc(&err, 1024));
    HIP_CHECK(hipMalloc(&Xh, (size_t)tiles * 256 * ldh * 2)); HIP_CHECK(hipMalloc(&Qh, (size_t)256 * ldh * 2 * 2)); HIP_CHECK(hipMalloc(&stats, 8)); HIP_CHECK(hipMalloc(&st4, 16));
    HIP_CHECK(hipMemset(Xh, 0, (size_t)tiles * 256 * ldh * 2)); HIP_CHECK(hipMemset(stats, 0, 8));
    HIP_CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    launch_to_half_rows(&c, dX, n, ld, Xh, ldh, 0, rn, stats);
    launch_prep_queries_fast(&c, dQ, B, ld, dim, Qh, ldh, qn, err, 0, 1.0002f, st4);
    float *S0, *bound;
    HIP_CHECK(hipMalloc(&S0, (size_t)256 * ldS * 4)); HIP_CHECK(hipMalloc(&bound, (size_t)256 * ldB * 4));
    // host reference on fp16-rounded operands
    std::vector<float> Xr(X.size()), Qr(Q.size());
    for (size_t i = 0; i < X.size(); i++) Xr[i] = (float)(_Float16)X[i];
    for (size_t i = 0; i < Q.size(); i++) Qr[i] = (float)(_Float16)Q[i];
    int bad_total = 0;
    unsigned long long* dtrace = nullptr;
    if (getenv("SCAN_TRACE")) { HIP_CHECK(hipMalloc(&dtrace, (8 * 32 * 4 + 64) * 8)); HIP_CHECK(hipMemset(dtrace, 0, (8 * 32 * 4 + 64) * 8)); g_scan_trace = dtrace; }
    // variant 8: the int8 shadow on the wide tile (reference = float64 dot product of the DEQUANTISED operands read back from the device)
    const int ld8 = (int)round_up(dim, 256);
    void *X8 = nullptr, *Q8F = nullptr, *Q8R = nullptr; float *sxd = nullptr, *sqd = nullptr, *qn8 = nullptr, *err8 = nullptr; uint32_t* st8 = nullptr;
    std::vector<float> X8r, Q8r;          // dequantised rows / queries, [row][ld8]
    {
        HIP_CHECK(hipMalloc(&Q8R, (size_t)256 * ld8));
        HIP_CHECK(hipMalloc(&X8, (size_t)tiles * 256 * ld8)); HIP_CHECK(hipMemset(X8, 0, (size_t)tiles * 256 * ld8));
        HIP_CHECK(hipMalloc(&Q8F, (size_t)256 * ld8)); HIP_CHECK(hipMalloc(&sxd, tiles * 4)); HIP_CHECK(hipMalloc(&sqd, 1024)); HIP_CHECK(hipMalloc(&qn8, 1024)); HIP_CHECK(hipMalloc(&err8, 1024));
        HIP_CHECK(hipMalloc(&st8, 16)); HIP_CHECK(hipMemset(st8, 0, 16));
        launch_to_i8_tiles(&c, dX, n, ld, X8, ld8, 0, sxd, st8);
        uint32_t hst[4]; HIP_CHECK(hipMemcpy(hst, st8, 16, hipMemcpyDeviceToHost)); float dx2; memcpy(&dx2, &hst[2], 4);
        launch_prep_queries_i8(&c, 0, nullptr, B, dim, dQ, ld, nullptr, Q8F, Q8R, ld8, sqd, qn8, err8, 0, 1.0002f, std::sqrt(dx2), st4);
        HIP_CHECK(hipStreamSynchronize(c.stream));
        if (n <= 20000) {
        std::vector<signed char> hx((size_t)tiles * 256 * ld8), hq((size_t)256 * ld8); std::vector<float> hsx(tiles), hsq(256), herr(256);
        HIP_CHECK(hipMemcpy(hx.data(), X8, hx.size(), hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(hq.data(), Q8F, hq.size(), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(hsx.data(), sxd, tiles * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(hsq.data(), sqd, 1024, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(herr.data(), err8, 1024, hipMemcpyDeviceToHost));
        X8r.assign((size_t)n * ld8, 0.f); Q8r.assign((size_t)256 * ld8, 0.f);
        const int nk8 = ld8 >> 7;
        double worst = 0, worst_in = 0;
        for (long r = 0; r < n; r++) for (int k = 0; k < ld8; k++) {
            const long off = (((r >> 8) * (ld8 >> 6) + (k >> 6)) * 256 + (r & 255)) * 64 + (k & 63);
            X8r[r * ld8 + k] = hsx[r >> 8] * (float)hx[off];
        }
        for (int q = 0; q < 256; q++) for (int k = 0; k < ld8; k++) {
            const long off = ((((long)(q >> 5) * nk8 + (k >> 7)) * 4 + ((k >> 5) & 3)) * 64 + ((k >> 4) & 1) * 32 + (q & 31)) * 16 + (k & 15);
            Q8r[(size_t)q * ld8 + k] = hsq[q] * (float)hq[off];
        }
        // the bound: |x.q - xhat.qhat| <= E for every (row, query) checked
        for (int q = 0; q < B; q += 5) for (long r = 0; r < n; r += 97) {
            double ex = 0, ap = 0; for (int j = 0; j < dim; j++) { ex += (double)X[r * ld + j] * Q[q * ld + j]; ap += (double)X8r[r * ld8 + j] * Q8r[(size_t)q * ld8 + j]; }
            worst = std::max(worst, std::fabs(ex - ap) / herr[q]); worst_in = std::max(worst_in, std::fabs(ex - ap));
        }
        printf("int8 shadow: max ||delta|| %.3e, E(query 0) %.3e, worst |x.q - xhat.qhat| %.3e = %.3f of E\n", std::sqrt(dx2), herr[0], worst_in, worst);
        if (worst > 1.0) { printf("int8 bound VIOLATED\n"); bad_total++; }
        }
    }
    // variant 9: the register-stationary int8 tile (flat_scan_qr_kernel; wide batches, ld8 in {256, 512, 768}); variant 8: the query-stationary one
    for (int variant : {0, 1, 8, 9}) {
        if (variant == 1 && unit == 64 && B > 64) continue;
        if (getenv("SCAN_ONLY") && atoi(getenv("SCAN_ONLY")) != variant) continue;
        if (variant == 9 && flat_scan_qr_steps(ld8) == 0) continue;
        setenv("COMET_SCAN_VARIANT_RT", variant == 1 ? "1" : "0", 1);
        setenv("COMET_SCAN_QR_RT", variant == 9 ? "1" : "0", 1);
        HIP_CHECK(hipMemset(S0, 0xFF, (size_t)256 * ldS * 4)); HIP_CHECK(hipMemset(bound, 0xFF, (size_t)256 * ldB * 4));
        auto run = [&]() {
            if (variant >= 8) launch_flat_scan_i8(&c, 0, X8, n, ld8, Q8F, Q8R, B, rn, qn8, sxd, sqd, nullptr, S0, ldS, bound, ldB, unit);
            else launch_flat_scan_f16(&c, 0, Xh, n, ldh, Qh, B, rn, qn, nullptr, S0, ldS, bound, ldB, unit);
        };
        run();
        HIP_CHECK(hipStreamSynchronize(c.stream));
        if (dtrace) { run(); run(); HIP_CHECK(hipStreamSynchronize(c.stream)); }      // the trace of a warm launch
        std::vector<float> hS((size_t)256 * ldS), hB((size_t)256 * ldB);
        HIP_CHECK(hipMemcpy(hS.data(), S0, hS.size() * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(hB.data(), bound, hB.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; double maxerr = 0;
        if (variant >= 8 && X8r.empty()) printf("variant %d: keys not checked at this size\n", variant);
        else for (int q = 0; q < B && bad < 10; q += 7) for (long u = 0; u < units; u++) {
            std::vector<std::pair<double, int>> d;
            for (int r = 0; r < unit; r++) {
                const long row = u * unit + r; if (row >= n) break; double s = 0;
                if (variant >= 8) for (int j = 0; j < dim; j++) s += (double)X8r[row * ld8 + j] * Q8r[(size_t)q * ld8 + j];
                else for (int j = 0; j < dim; j++) s += (double)Xr[row * ld + j] * Qr[q * ld + j];
                d.push_back({std::max(0.0, 1.0 - s), r});
            }
            std::sort(d.begin(), d.end());
            for (int e = 0; e < 2 && e < (int)d.size(); e++) {
                const float key = hS[(size_t)q * ldS + 2 * u + e]; uint32_t kb; memcpy(&kb, &key, 4);
                const int row = kb & (unit - 1); uint32_t vb = kb & 0xFFFFFF00u; float v; memcpy(&v, &vb, 4);
                const double want = d[e].first; maxerr = std::max(maxerr, std::fabs(v - want));
                if (std::fabs(v - want) > 2e-3 || (row != d[e].second && std::fabs(d[e].first - (e + 1 < (int)d.size() ? d[e + 1].first : 9)) > 1e-3 && std::fabs(v - want) > 1e-4)) { if (bad++ < 5) printf("variant %d q %d unit %ld e %d: got %.6f row %d, want %.6f row %d\n", variant, q, u, e, v, row, want, d[e].second); }
            }
        }
        printf("variant %d: %s (max |key - float64| = %.2e)\n", variant, bad ? "MISMATCH" : "ok", maxerr);
        if (dtrace && variant == 9) {
            std::vector<unsigned long long> t(8 * 32 * 4 + 64);
            HIP_CHECK(hipMemcpy(t.data(), dtrace, t.size() * 8, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemset(dtrace, 0, (8 * 32 * 4 + 64) * 8));
            printf("TRACE of variant 9 (register-stationary tile; one pass = 64 rows x %d bytes, %d MFMAs per wave)\n", ld8, ld8 / 32 * 4);
            for (int w = 0; w < 4; w++)
                for (int p = 0; p < 23; p++) {
                    const unsigned long long* q = &t[(w * 24 + p) * 4]; const unsigned long long nxt = t[(w * 24 + p + 1) * 4];
                    if (!q[0] || !nxt) break;
                    printf("TRACE wave %d pass %2d: mfma+select to the wait %6llu  vmcnt wait %5llu  barrier %5llu  tail to next pass %5llu  total %6llu\n", w, p, q[1] - q[0], q[2] - q[1], q[3] - q[2], nxt - q[3], nxt - q[0]);
                }
        }
        if (dtrace && (variant == 0 || variant == 8) && B > 64) {
            std::vector<unsigned long long> t(8 * 32 * 4 + 64);
            HIP_CHECK(hipMemcpy(t.data(), dtrace, t.size() * 8, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemset(dtrace, 0, (8 * 32 * 4 + 64) * 8));
            const int nk = variant == 8 ? ld8 / 128 : ldh / 64;
            printf("TRACE of variant %d (%d K steps per tile)\n", variant, nk);
            for (int w : {0, 4}) {
                for (int g = 0; g < 2 * nk && g < 31; g++) {
                    const unsigned long long* p = &t[(w * 32 + g) * 4]; const unsigned long long nxt = t[(w * 32 + g + 1) * 4];
                    printf("TRACE wave %d step %2d: wait+barrier %5llu  mfma+issue %5llu  rest %5llu  total %5llu%s\n", w, g, p[1] - p[0], p[2] - p[1], nxt - p[2], nxt - p[0],
                           (g % nk) == nk - 1 ? "   <- includes the tile's epilogue" : "");
                }
                const unsigned long long* e = &t[8 * 32 * 4 + w * 8];
                printf("TRACE wave %d epilogue: unit0 network %llu, unit0 merge+store %llu, unit1 network %llu, end at +%llu after mfma stamp\n", w, e[1] - e[0], e[2] - e[1], e[3] - e[2],
                       t[(w * 32 + 2 * nk - 1) * 4 + 3] - t[(w * 32 + 2 * nk - 1) * 4 + 2]);
            }
        }
        bad_total += bad;
        if (iters > 0) {
            hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
            for (int i = 0; i < 3; i++) run();
            HIP_CHECK(hipEventRecord(a, c.stream));
            for (int i = 0; i < iters; i++) run();
            HIP_CHECK(hipEventRecord(b, c.stream)); HIP_CHECK(hipEventSynchronize(b));
            float ms; HIP_CHECK(hipEventElapsedTime(&ms, a, b));
            printf("variant %d: %.4f ms per launch (%ld rows x %d, %d queries)\n", variant, ms / iters, n, dim, B);
        }
    }
    return bad_total ? 1 : 0;
}
