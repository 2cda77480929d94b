// index_quant.hip — IVF / PQ / IVFPQ indexes.
#include "index.hpp"
namespace comet {
comet_index* make_ivf(Ctx*, int, int, int) { COMET_FAIL(COMET_ERR_UNSUPPORTED, "IVF not built yet"); }
comet_index* make_pq(Ctx*, int, int, int, int) { COMET_FAIL(COMET_ERR_UNSUPPORTED, "PQ not built yet"); }
comet_index* make_ivfpq(Ctx*, int, int, int, int, int) { COMET_FAIL(COMET_ERR_UNSUPPORTED, "IVFPQ not built yet"); }
}
extern "C" {
int comet_kmeans(comet_ctx*, const float*, int64_t, int, int, int, int, float*, int32_t*, int*) { return comet::set_error(COMET_ERR_UNSUPPORTED, "kmeans not built yet"); }
int comet_nearest_centroid(comet_ctx*, const float*, int64_t, int, const float*, int, int, int32_t*) { return comet::set_error(COMET_ERR_UNSUPPORTED, "not built yet"); }
int comet_merge_topk_dev(comet_ctx*, const uint32_t*, const float*, const int32_t*, int32_t, int32_t, int32_t, int32_t, uint32_t*, float*, int32_t*) { return comet::set_error(COMET_ERR_UNSUPPORTED, "not built yet"); }
}
