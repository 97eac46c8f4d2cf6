// Placeholder until the MFMA fast path lands (next commit).
#include "fn2_common.hpp"
namespace fn2 {
struct CorrGeom;
bool corr_fwd_mfma_supported(const CorrGeom&) { return false; }
int corr_fwd_mfma_launch(const CorrGeom&, const float*, const float*, float*, hipStream_t) { return FN2_ERR_UNSUPPORTED; }
}  // namespace fn2
