// kernels_scanq.hip — register-stationary scan tiles of the Flat fast path's int8 shadow (kernels_scanq.inc.hpp): the cosine instantiations and the
// dispatcher. The L2-family instantiations live in kernels_scanq_l2.hip.
#include "kernels_scanq.inc.hpp"

namespace comet {
template void launch_flat_scan_qr_mode<0>(Ctx*, int, const void*, int64_t, const void*, int, const float*, const float*, const float*, const float*, const uint8_t*, float*, int64_t, float*, int64_t, int);
extern template void launch_flat_scan_qr_mode<1>(Ctx*, int, const void*, int64_t, const void*, int, const float*, const float*, const float*, const float*, const uint8_t*, float*, int64_t, float*, int64_t, int);

int flat_scan_qr_steps(int ld8) { return (ld8 == 256 || ld8 == 512 || ld8 == 768) ? ld8 / 128 : 0; }    // K steps per row the register-stationary tiles are built for (0: not these)

// Launch the register-stationary tile for this scan if there is one (int8 shadow rows of 256 / 512 / 768 bytes; COMET_SCAN_QR=0 turns them off — read once.
// tools/scan_check.hip, which compiles this file into its own program with -DCOMET_SCAN_QR_RUNTIME_SWITCH, switches per call through COMET_SCAN_QR_RT: the
// library itself never calls getenv on the search path). Returns false if the caller must use the older tiles.
bool launch_flat_scan_qr(Ctx* c, int mode, const void* X8, int64_t n, int ld8, const void* Q8F, int nq_used, const float* rn, const float* qn,
                         const float* sx, const float* sq, const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows) {
    static const int qr_env = [] { const char* e = getenv("COMET_SCAN_QR"); return e ? atoi(e) : 1; }();
#ifdef COMET_SCAN_QR_RUNTIME_SWITCH
    const char* qr_rt = getenv("COMET_SCAN_QR_RT");
    const int nks = ((qr_rt ? atoi(qr_rt) : qr_env) != 0) ? flat_scan_qr_steps(ld8) : 0;
#else
    const int nks = qr_env != 0 ? flat_scan_qr_steps(ld8) : 0;
#endif
    if (!nks) return false;
    if (mode == 0) launch_flat_scan_qr_mode<0>(c, nks, X8, n, Q8F, nq_used, rn, qn, sx, sq, elig, S0, ldS, bound, ldB, unit_rows);
    else launch_flat_scan_qr_mode<1>(c, nks, X8, n, Q8F, nq_used, rn, qn, sx, sq, elig, S0, ldS, bound, ldB, unit_rows);
    return true;
}
}  // namespace comet
