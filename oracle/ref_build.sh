#!/bin/bash
# Builds oracle/_ref/libfn2_ref.so from the reference's OWN layer sources, compiled where they lie under
# /root/reference (nothing is copied), as HIP for gfx950, against the stand-in Caffe headers in
# flownet2_amd/csrc/caffe_adapter/compat/ and empty third-party stubs in oracle/stubs/.
# Unbuildable parts of the reference (stated in DESIGN.md / oracle/README.md): its Makefile/CMake build
# (boost, glog, gflags, protobuf, BLAS, HDF5, LMDB, OpenCV, nvcc) and L1LossLayer (composed from stock Eltwise /
# Power / Convolution layers that need the BLAS/im2col stack).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${FN2_REFERENCE_ROOT:-/root/reference}
OUT="$HERE/_ref"
COMPAT="$HERE/../flownet2_amd/csrc/caffe_adapter/compat"
[ -d "$REF/src/caffe/layers" ] || { echo "reference tree not found at $REF; keeping any prebuilt $OUT" >&2; exit 0; }
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fvisibility=hidden -w -I$COMPAT -I$HERE/stubs -I$REF/include -I$REF/src -fopenmp"
LAYERS="correlation_layer flow_warp_layer resample_layer channel_norm_layer downsample_layer"
OBJS=""
for l in $LAYERS; do
  for ext in cpp cu; do
    src="$REF/src/caffe/layers/$l.$ext"
    obj="$OUT/$l.$ext.o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$0" -nt "$obj" ]; then
      $HIPCC $FLAGS -x hip -c "$src" -o "$obj"
    fi
    OBJS="$OBJS $obj"
  done
done
$HIPCC $FLAGS -x hip -c "$HERE/ref_shim.cpp" -o "$OUT/ref_shim.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -fopenmp -o "$OUT/libfn2_ref.so" $OBJS "$OUT/ref_shim.o"
echo "built $OUT/libfn2_ref.so from $REF"
