#!/bin/bash
# Compile-only check of the Caffe adapter against the REFERENCE's real headers (include/caffe/layer.hpp, blob.hpp, common.hpp,
# layer_factory.hpp, syncedmem.hpp, util/math_functions.hpp, util/device_alternate.hpp, util/mkl_alternate.hpp ...): the
# reference's include/ comes FIRST on the include path; only third-party headers (boost, glog, gflags, CUDA, cuBLAS, cuRAND,
# CBLAS) and the protoc-generated caffe.pb.h are stand-ins.  A signature drift between compat/ (which the runnable test
# harness uses) and the real Layer<Dtype> / Blob<Dtype> / LayerRegistry interfaces fails here.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${FN2_REFERENCE_ROOT:-/root/reference}
[ -f "$REF/include/caffe/layer.hpp" ] || { echo "reference tree not found at $REF: check skipped" >&2; exit 0; }
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${1:-/tmp/fn2_real_headers_check.o}
$HIPCC --offload-arch=gfx950 -O0 -std=c++17 -fPIC -w -DUSE_CUDNN=0 -UUSE_CUDNN \
  -I"$HERE/stubs" -I"$REF/include" -I"$HERE/proto" -I"$HERE/../../../../include" \
  -x hip -c "$HERE/../fn2_caffe_layers.cpp" -o "$OUT"
echo "fn2_caffe_layers.cpp compiles against $REF/include"
