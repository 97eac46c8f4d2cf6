#!/bin/bash
# FlowWarp backward tile-kernel variants, rebuilt ON the GPU box (hipcc is in the image) and timed one after the other:
#   bash scripts/probes/warp_tile_variants.sh <tag>      -> gpurun_out/<tag>/warp_variants.txt
# Each variant = sed over the constants of csrc/flow_warp.hip, recompile that one object, relink libflownet2_hip.so.
set -u
TAG=${1:-warpvar}
R=gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
SRC=flownet2_amd/csrc/flow_warp.hip
cp $SRC /tmp/flow_warp.orig.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
build() {
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $SRC -o flownet2_amd/build/flow_warp.hip.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flownet2_amd/libflownet2_hip.so flownet2_amd/build/*.o -lz
}
run() {
  echo "== $1" >> $R/warp_variants.txt
  FN2_MB_ONLY="FlowWarp bwd" python scripts/layer_microbench.py 2>/dev/null | grep "FlowWarp bwd" >> $R/warp_variants.txt
}
: > $R/warp_variants.txt
build; run "as committed"
# kernel split of the committed form
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -o w -- env FN2_MB_ONLY="FlowWarp bwd [4,3" python scripts/layer_microbench.py > /dev/null 2>&1
f=$(find $R/trace -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -d, -f1-6 >> $R/warp_variants.txt
sed -i 's/constexpr int kWarpBatch = 8;/constexpr int kWarpBatch = 4;/' $SRC; build; run "batch 4"
sed -i 's/constexpr int kWarpNear = 32;/constexpr int kWarpNear = 16;/' $SRC; build; run "batch 4, near 16"
sed -i 's/constexpr int kWarpNear = 16;/constexpr int kWarpNear = 8;/' $SRC; build; run "batch 4, near 8"
cp /tmp/flow_warp.orig.hip $SRC
python - <<'P' >> $R/warp_variants.txt
# the three-kernel form of rounds 1-5 on this box, for reference
import os, sys
sys.path.insert(0, ".")
P
build
python -c "
import sys; sys.path.insert(0,'.')
from flownet2_amd import ops
ops.debug_set_flow_warp_impl(True)
import runpy, os
os.environ['FN2_MB_ONLY']='FlowWarp bwd'
runpy.run_path('scripts/layer_microbench.py', run_name='__main__')
" 2>/dev/null | grep "FlowWarp bwd" | sed 's/^/global lists: /' >> $R/warp_variants.txt
cat $R/warp_variants.txt
