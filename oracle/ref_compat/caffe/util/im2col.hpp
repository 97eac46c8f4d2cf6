// oracle/_ref build: the reference's own im2col declarations (pure prototypes; src/caffe/util/im2col.{cpp,cu} are
// compiled in place).  The adapter's stand-in of this header is empty, so reach past it by path.
#pragma once
#define FN2_STR2(x) #x
#define FN2_STR(x) FN2_STR2(x)
#include FN2_STR(FN2_REF_INC/caffe/util/im2col.hpp)
