#!/bin/bash
# Builds oracle/_ref/libfn2_ref.so from the reference's OWN layer sources, compiled where they lie under
# /root/reference (nothing is copied), as HIP for gfx950, against the stand-in Caffe headers in
# flownet2_amd/csrc/caffe_adapter/compat/ and empty third-party stubs in oracle/stubs/.
# Unbuildable parts of the reference (stated in DESIGN.md / oracle/README.md): its Makefile/CMake build
# (boost, glog, gflags, protobuf, BLAS, HDF5, LMDB, OpenCV, nvcc).  L1LossLayer is built together with the stock
# Eltwise / Power / Convolution layers it is composed from; the cuBLAS / CBLAS calls underneath those are the only
# part replaced (oracle/ref_compat).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${FN2_REFERENCE_ROOT:-/root/reference}
OUT="$HERE/_ref"
COMPAT="$HERE/../flownet2_amd/csrc/caffe_adapter/compat"
[ -d "$REF/src/caffe/layers" ] || { echo "reference tree not found at $REF; keeping any prebuilt $OUT" >&2; exit 0; }
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fvisibility=hidden -w -I$HERE/ref_compat -I$COMPAT -I$HERE/stubs -I$REF/include -I$REF/src -DFN2_REF_INC=$REF/include -fopenmp"
# the five custom layers (+ Correlation1D, the horizontal variant), then L1LossLayer and the stock layers it is composed from (l1loss_layer.cpp:19-62), plus the stock
# Deconvolution and ReLU layers (pins of the stock-layer fast paths: stem, flow heads, GEMM route, bias + ReLU); cuBLAS/CBLAS
# are replaced by the plain stand-ins of oracle/ref_compat/caffe/util/math_functions.hpp.  custom_data_layer (the LMDB data layer whose
# DecodeData defines the sample format) is built against an in-memory stand-in for liblmdb (oracle/stubs/lmdb.h)
LAYERS="correlation_layer correlation_layer1d flow_warp_layer resample_layer channel_norm_layer downsample_layer l1loss_layer eltwise_layer power_layer conv_layer deconv_layer relu_layer custom_data_layer flow_augmentation_layer data_augmentation_layer"
EXTRA="layers/augmentation_layer_base.cpp layers/base_conv_layer.cpp layers/loss_layer.cpp layers/neuron_layer.cpp util/im2col.cpp util/im2col.cu"
# newest stand-in header: an object older than it is rebuilt (the stand-ins define Blob / Layer layouts)
NEWEST_HDR=$(find "$COMPAT" "$HERE/ref_compat" "$HERE/stubs" -type f -printf '%T@ %p\n' | sort -n | tail -1 | cut -d' ' -f2-)
OBJS=""
for l in $LAYERS; do
  for ext in cpp cu; do
    src="$REF/src/caffe/layers/$l.$ext"
    obj="$OUT/$l.$ext.o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$0" -nt "$obj" ] || [ "$NEWEST_HDR" -nt "$obj" ]; then
      $HIPCC $FLAGS -x hip -c "$src" -o "$obj"
    fi
    OBJS="$OBJS $obj"
  done
done
for f in $EXTRA; do
  src="$REF/src/caffe/$f"
  obj="$OUT/$(echo $f | tr / _).o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$0" -nt "$obj" ] || [ "$NEWEST_HDR" -nt "$obj" ]; then
    $HIPCC $FLAGS -x hip -c "$src" -o "$obj"
  fi
  OBJS="$OBJS $obj"
done
$HIPCC $FLAGS -DFN2_SHIM_L1LOSS -DFN2_SHIM_STOCK -DFN2_SHIM_DATA -x hip -c "$HERE/ref_shim.cpp" -o "$OUT/ref_shim.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -fopenmp -o "$OUT/libfn2_ref.so" $OBJS "$OUT/ref_shim.o"
echo "built $OUT/libfn2_ref.so from $REF"
