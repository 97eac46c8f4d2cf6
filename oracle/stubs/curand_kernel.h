// stand-in: data_augmentation_layer.cu includes the cuRAND device header but uses none of it (noise comes from caffe_gpu_rng_gaussian);
// the CUDA header is also what brings FLT_MAX into that file
#pragma once
#include <cfloat>
