// Host-side helpers shared by the augmentation layers: AugmentationCoeff as a flat record (caffe.proto:436-486, declaration order =
// the array layout of coeff_to_array / array_to_coeff, augmentation_layer_base.cpp:352-380) and tTransMat (cpp:14-68).
#pragma once
#include <cmath>

#include "fn2_common.hpp"

namespace fn2 {

enum AugField {
  A_MIRROR, A_DX, A_DY, A_ANGLE, A_ZOOM_X, A_ZOOM_Y,
  A_GAMMA, A_BRIGHTNESS, A_CONTRAST, A_COLOR1, A_COLOR2, A_COLOR3,
  A_POW_NOMEAN0, A_POW_NOMEAN1, A_POW_NOMEAN2, A_ADD_NOMEAN0, A_ADD_NOMEAN1, A_ADD_NOMEAN2,
  A_MULT_NOMEAN0, A_MULT_NOMEAN1, A_MULT_NOMEAN2, A_POW_WITHMEAN0, A_POW_WITHMEAN1, A_POW_WITHMEAN2,
  A_ADD_WITHMEAN0, A_ADD_WITHMEAN1, A_ADD_WITHMEAN2, A_MULT_WITHMEAN0, A_MULT_WITHMEAN1, A_MULT_WITHMEAN2,
  A_LMULT_POW, A_LMULT_ADD, A_LMULT_MULT, A_COL_ANGLE,
  A_FOG_AMOUNT, A_FOG_SIZE, A_MOTION_BLUR_ANGLE, A_MOTION_BLUR_SIZE, A_SHADOW_ANGLE, A_SHADOW_DISTANCE, A_SHADOW_STRENGTH, A_NOISE,
  A_COUNT
};
static_assert(A_COUNT == FN2_AUG_NUM_PARAMS, "AugmentationCoeff has 42 fields");

// proto defaults, in declaration order
constexpr float kAugDefault[A_COUNT] = {0, 0, 0, 0, 1, 1,  1, 0, 1, 1, 1, 1,  1, 1, 1, 0, 0, 0,  1, 1, 1, 1, 1, 1,
                                        0, 0, 0, 1, 1, 1,  1, 0, 1, 0,  0, 0, 0, 0, 0, 0, 0, 0};

struct AugCoeff {
  float v[A_COUNT];
  bool has[A_COUNT];
  // array_to_coeff, cpp:368-380: every field is SET (has-bit on); fields with a non-zero default come back through exp.
  // `exp(in[fn])` with a float argument resolves to ::exp(double) in that translation unit: computed in double, stored as float.
  void from_array(const float* in) {
    for (int f = 0; f < A_COUNT; ++f) {
      v[f] = std::fabs(kAugDefault[f]) < 1e-3f ? in[f] : (float)std::exp((double)in[f]);
      has[f] = true;
    }
  }
  // clear_defaults, cpp:340-350: a field within 1e-3 of its default is cleared (value = default, has-bit off)
  void clear_defaults() {
    for (int f = 0; f < A_COUNT; ++f)
      if (std::fabs(kAugDefault[f] - v[f]) < 1e-3) { v[f] = kAugDefault[f]; has[f] = false; }
  }
};

// tTransMat, include/caffe/layers/augmentation_layer_base.hpp:20-35:  | t0 t2 t4 |
//                                                                     | t1 t3 t5 |
struct TransMat {
  float t0, t1, t2, t3, t4, t5;
  void identity() { t0 = 1; t2 = 0; t4 = 0; t1 = 0; t3 = 1; t5 = 0; }                        // cpp:15-19
  void left_multiply(float u0, float u1, float u2, float u3, float u4, float u5) {           // cpp:22-35
    const float a0 = t0, a2 = t2, a4 = t4, a1 = t1, a3 = t3, a5 = t5;
    t0 = a0 * u0 + a1 * u2;
    t1 = a0 * u1 + a1 * u3;
    t2 = a2 * u0 + a3 * u2;
    t3 = a2 * u1 + a3 * u3;
    t4 = a4 * u0 + a5 * u2 + u4;
    t5 = a4 * u1 + a5 * u3 + u5;
  }
  // fromCoeff, cpp:38-49.  leftMultiply takes floats; its call sites compute the arguments in double (.5 * float, cos(double),
  // 1.0 / float) and convert.
  void from_coeff(const AugCoeff& c, int width, int height, int bottomwidth, int bottomheight) {
    if (c.v[A_MIRROR]) left_multiply(-1, 0, 0, 1, (float)(.5 * (double)(float)width), (float)(-.5 * (double)(float)height));
    else left_multiply(1, 0, 0, 1, (float)(-.5 * (double)(float)width), (float)(-.5 * (double)(float)height));
    const double ang = (double)c.v[A_ANGLE];
    if (c.has[A_ANGLE]) left_multiply((float)std::cos(ang), (float)std::sin(ang), (float)-std::sin(ang), (float)std::cos(ang), 0, 0);
    if (c.has[A_DX] || c.has[A_DY]) left_multiply(1, 0, 0, 1, c.v[A_DX] * (float)width, c.v[A_DY] * (float)height);
    if (c.has[A_ZOOM_X] || c.has[A_ZOOM_Y]) left_multiply((float)(1.0 / (double)c.v[A_ZOOM_X]), 0, 0, (float)(1.0 / (double)c.v[A_ZOOM_Y]), 0, 0);
    left_multiply(1, 0, 0, 1, (float)(.5 * (double)(float)bottomwidth), (float)(.5 * (double)(float)bottomheight));
  }
  TransMat inverse() const {                                                                 // cpp:52-68
    const float a = t0, c = t2, e = t4, b = t1, d = t3, f = t5;
    const float denom = a * d - b * c;
    TransMat r;
    r.t0 = d / denom;
    r.t1 = -b / denom;
    r.t2 = -c / denom;
    r.t3 = a / denom;
    r.t4 = (c * f - d * e) / denom;
    r.t5 = (b * e - a * f) / denom;
    return r;
  }
};

}  // namespace fn2
