#!/bin/bash
# Correlation backward (corr_bwd_g3) ablation builds, rebuilt ON the GPU box and timed one after the other at config A:
#   bash scripts/probes/corr_bwd_variants.sh <tag>   -> gpurun_out/<tag>/corr_bwd_variants.txt
set -u
TAG=${1:-corrbwd}
R=gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
SRC=flownet2_amd/csrc/correlation_bwd_mfma.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wall -Wno-unused-function"
OUT=$R/corr_bwd_variants.txt
: > $OUT
build() {
  /opt/rocm/bin/hipcc $FLAGS $1 -x hip -c $SRC -o flownet2_amd/build/correlation_bwd_mfma.hip.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o flownet2_amd/libflownet2_hip.so flownet2_amd/build/*.o -lz
}
run() {
  echo "== $1" >> $OUT
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/t_$2 -o c -- python scripts/corr_microbench.py --iters 1500 --backward --impl ${FN2_CORR_IMPL:-0} > /dev/null 2>&1
  f=$(find $R/t_$2 -name "*_kernel_stats.csv" | head -1)
  [ -n "$f" ] && python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'corr_bwd' in r['Name']: print('  %-22s calls %s  avg %.2f us' % (r['Name'].split('(')[0].replace('void fn2::bwd::g3::',''), r['Calls'], float(r['AverageNs'])/1e3))
" >> $OUT
  rm -rf $R/t_$2
}
if [ "${FN2_GEN:-4}" = "3" ]; then
  for v in 0 1 2 3 4 7; do build "-DFN2_G3_ABL=$v"; run "generation 3, ablation $v (1 no G DMA, 2 no other-map DMA, 4 no MFMA)" $v; done
else
  for v in 0 1 2 3 4 7 8 15; do build "-DFN2_G4_ABL=$v"; run "generation 4 (both bottoms, one launch), ablation $v (1 no G DMA, 2 no other-map DMA, 4 no MFMA, 8 no stores)" $v; done
fi
build ""
cat $OUT
