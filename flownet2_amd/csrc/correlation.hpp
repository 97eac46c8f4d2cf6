// Shared between correlation.hip (generic kernels, C ABI) and correlation_mfma.hip (fast path).
#pragma once
#include "fn2_common.hpp"

namespace fn2 {

struct CorrGeom {
  int N, C, H, W;
  int pad, K, md, s1, s2, kr;
  int topC, topH, topW, ngr, ngw;
  int type;
  // fused epilogue (fn2_correlation_forward_fused): the top blob may be a channel slice [top_c0, top_c0 + topC) of a blob with
  // top_ctot channels, and the in-place ReLU that follows the layer in the FlowNetC graph can be applied on the way out
  int top_ctot, top_c0, relu;
  float slope;
};

bool corr_fwd_mfma_supported(const CorrGeom& g);
int corr_fwd_mfma_launch(const CorrGeom& g, const float* b0, const float* b1, float* top, hipStream_t st);
// third-generation forward of the FlowNetC instance (correlation_units.hip): unit lists, loader wave; corr_fwd_pair serves what it does not take
bool corr_fwd_units_supported(const CorrGeom& g, const float* b0, const float* b1, const float* top);
int corr_fwd_units_launch(const CorrGeom& g, const float* b0, const float* b1, float* top, hipStream_t st);
int corr_fwd_units_plan_words(int N, int H, int W, int policy, unsigned* out, int max_words);
bool corr_bwd_mfma_supported(const CorrGeom& g);
int corr_bwd_mfma_launch(const CorrGeom& g, int which, const float* other, const float* top_diff, float* out, hipStream_t st);
// both bottom diffs in ONE launch (round 6); FN2_ERR_UNSUPPORTED (no error text) where it does not apply
int corr_bwd_mfma_launch_both(const CorrGeom& g, const float* b0, const float* b1, const float* top_diff, float* d0, float* d1, hipStream_t st);

}  // namespace fn2
