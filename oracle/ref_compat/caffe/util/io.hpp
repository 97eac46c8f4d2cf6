// Stand-in for include/caffe/util/io.hpp in the pin build: the one function the data layer names.
#pragma once
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"
namespace caffe {
inline void ReadProtoFromBinaryFileOrDie(const char* filename, BlobProto*) { LOG(FATAL) << "mean_file (" << filename << ") is not available in the pin harness"; }
// DataAugmentationLayer::LayerSetUp dumps its LayerParameter to a hard-coded path on the authors' cluster (data_augmentation_layer.cpp:68)
template <typename M> inline void WriteProtoToTextFile(const M&, const char*) {}
}  // namespace caffe
