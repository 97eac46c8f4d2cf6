#!/bin/bash
# Compiles the Caffe adapter against the stand-in headers (compat/) and links it with libflownet2_hip.so and the
# layer-driving C shim into _build/libfn2_caffe_adapter_test.so -- the "does the plug-in compile and run" check
# for a box without a Caffe tree (tests/test_caffe_adapter.py).  In a real Caffe tree only fn2_caffe_layers.cpp is
# used.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../../.."
OUT="$HERE/_build"
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-parameter -Wno-unused-function -I$HERE/compat -I$ROOT/include"
$HIPCC $FLAGS -x hip -c "$HERE/fn2_caffe_layers.cpp" -o "$OUT/fn2_caffe_layers.o"
# The FlowNetC `caffe time` driver (fn2ref_flownetc_time, oracle/ref_shim.cpp) chains the plug-ins with the REFERENCE's own in-place ReLU and
# Concat layers: compiled where they lie, only where the reference tree exists (here; the GPU box uses the prebuilt library).
REF=${FN2_REFERENCE_ROOT:-/root/reference}
STOCK_OBJS=""
NETDEF=""
if [ -d "$REF/src/caffe/layers" ]; then
  RFLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -fvisibility=hidden -w -I$ROOT/oracle/ref_compat -I$HERE/compat -I$ROOT/oracle/stubs -I$REF/include -I$REF/src"
  for f in relu_layer.cpp relu_layer.cu neuron_layer.cpp concat_layer.cpp concat_layer.cu; do
    $HIPCC $RFLAGS -x hip -c "$REF/src/caffe/layers/$f" -o "$OUT/ref_$f.o"
    STOCK_OBJS="$STOCK_OBJS $OUT/ref_$f.o"
  done
  NETDEF="-DFN2_SHIM_NET=1 -I$REF/include"
fi
$HIPCC $FLAGS -DFN2_SHIM_L1LOSS=1 -DFN2_SHIM_CONV_REGISTRY=1 $NETDEF -x hip -c "$ROOT/oracle/ref_shim.cpp" -o "$OUT/shim.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfn2_caffe_adapter_test.so" "$OUT/fn2_caffe_layers.o" "$OUT/shim.o" $STOCK_OBJS \
  -L"$ROOT/flownet2_amd" -lflownet2_hip -Wl,-rpath,'$ORIGIN/../../..'
echo "built $OUT/libfn2_caffe_adapter_test.so"
