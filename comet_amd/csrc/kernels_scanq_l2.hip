// kernels_scanq_l2.hip — the L2-family instantiations of the register-stationary scan tiles (kernels_scanq.inc.hpp)
#include "kernels_scanq.inc.hpp"

namespace comet {
template void launch_flat_scan_qr_mode<1>(Ctx*, int, const void*, int64_t, const void*, int, const float*, const float*, const float*, const float*, const uint8_t*, float*, int64_t, float*, int64_t, int);
}  // namespace comet
