// third-party stand-in (compile-only check): caffe/common.hpp only needs the header guard name
#pragma once
#define GFLAGS_GFLAGS_H_
namespace gflags {}
