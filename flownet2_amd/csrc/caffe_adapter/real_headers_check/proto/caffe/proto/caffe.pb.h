// the generated caffe.pb.h does not exist without protoc: the hand-written message stand-in of compat/ (plain structs with
// protoc accessor names) is the only Caffe-side header this check does NOT take from the reference tree
#include "../../../../compat/caffe/proto/caffe.pb.h"
