#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun):  bash scripts/profile_round.sh r01
# Writes rocprofv3 CSVs under gpurun_out/<tag>/; scripts/summarize_profiles.py and scripts/summarize_layer_microbench.py turn them into profiles/<tag>_*.
# Counters are collected in their own passes (never combined with trace domains other than the kernel trace);
# every pass runs under its own timeout (a counter pass that aborts can otherwise hang until the box limit).
set -u
TAG=${1:-r01}
R=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $R
# the convolution kernels pick their tile variant by timing every candidate on first use: record the picks once so that the
# profiled runs below launch no candidates
export FN2_AUTOTUNE_CACHE=$PWD/$R/autotune.txt
rm -f $FN2_AUTOTUNE_CACHE
# the unprofiled bench line first (what the driver runs, on a box that has done nothing else yet); it also records the autotune picks
python bench.py > $R/bench.json 2> $R/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/bench_profiled.json 2>/dev/null
# 6000 forward launches (0.25 s): the chip needs ~25 ms of load (the first ~500 launches) to reach its steady clocks -- a 200-launch run
# (rounds 1-2) sat inside that ramp and read 47 us for a kernel that runs at 41.3 us from launch 500 on; the average below includes the ramp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/corr -o corr -- python scripts/corr_microbench.py --iters 6000 --backward > $R/corr_stdout.txt 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -o corr -- python scripts/corr_microbench.py --iters 10 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -o corr -- python scripts/corr_microbench.py --iters 10 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc_sq -o corr -- python scripts/corr_microbench.py --iters 10 > /dev/null 2>&1
# the correlation BACKWARD kernel (round 6: both bottoms in one launch): matrix-pipe and LDS counters, a pass of their own
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc_sq_bwd -o corr -- python scripts/corr_microbench.py --iters 10 --backward > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_bench -o bench -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --corr-iters 4 > /dev/null 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE on streaming kernels of known byte count (dword per lane, like the staging loads)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/cal_fetch -o cal -- python scripts/hbm_calibrate.py > $R/cal_stdout.txt 2>/dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/cal_write -o cal -- python scripts/hbm_calibrate.py > /dev/null 2>&1
# custom-layer micro-benchmarks (own kernel trace), their host baselines, and the training input pipeline (FN2_PROFILE_LIGHT=1 skips them
# and the per-family counter passes at the end: a round that did not touch those kernels keeps the previous round's files)
if [ "${FN2_PROFILE_LIGHT:-0}" != "1" ]; then
python scripts/layer_microbench.py > gpurun_out/layer_microbench.txt 2>/dev/null
python tests/cpu_baselines.py >> gpurun_out/layer_microbench.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/layers -o lay -- python scripts/layer_microbench.py > /dev/null 2>&1
timeout 300 python scripts/train_pipeline.py --iters 8 2>/dev/null | tail -11 > gpurun_out/train_pipeline.txt
fi
# the other configurations of BASELINE.json (FlowNet2 at 768x384 batch 4 and 1024x448 batch 1, FlowNetC training step) and the
# per-variant convolution timings
python bench.py --net 2 --batch 4 --height 384 --width 768 --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $R/bench_flownet2.json 2>/dev/null
python bench.py --net 2 --batch 1 --height 448 --width 1024 --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $R/bench_flownet2_1024.json 2>/dev/null
python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $R/bench_train.json 2>/dev/null
timeout 300 python scripts/conv_bench.py --net C --layers conv2,conv3,conv3_1,conv4_1 > $R/conv_bench_C.txt 2>&1
# the small-map kernels (csrc/conv_plane.hip): every variant of the convolutions and of the (opt-in) deconvolutions next to the GEMM routes
timeout 300 python scripts/conv_bench.py --net C --layers conv4,conv5,conv5_1,conv6,conv6_1 --only-plane > $R/conv_plane_bench_C.txt 2>&1
timeout 300 python scripts/deconv_bench.py --net C > $R/deconv_bench_C.txt 2>&1
# kernel table of the FlowNet2 step (768x384, batch 4)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench2 -o bench2 -- python bench.py --net 2 --batch 4 --height 384 --width 768 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/bench2b1 -o bench2b1 -- python bench.py --net 2 --batch 1 --height 448 --width 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(find $R/bench2b1 -name "*_kernel_stats.csv" | head -1); [ -n "$f" ] && python scripts/nonfn2_kernels.py $f > $R/bench2b1_nonfn2.txt 2>&1
# matrix-pipe counters of the convolution kernels alone (a counter pass of its own)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc_conv -o conv -- env FN2_AUTOTUNE=0 python scripts/conv_bench.py --net C --layers conv2,conv3_1 --iters 3 > /dev/null 2>&1
# round 3: kernel table of the FlowNetC TRAINING step, the weight-gradient / transposed-convolution micro-benchmarks, SQ counters of the weight-gradient kernel
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/train -o t -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/train_profiled.json 2>/dev/null
python scripts/summarize_train_trace.py $R/train/t_kernel_stats.csv > $R/train_kernels.txt 2>&1
if [ "${FN2_PROFILE_LIGHT:-0}" != "1" ]; then
timeout 300 python scripts/wgrad_bench.py > $R/wgrad_bench.txt 2>&1
timeout 300 python scripts/tconv_bench.py > $R/tconv_bench.txt 2>&1
bash scripts/wgrad_pmc.sh $TAG/wgrad_pmc conv2,conv3_1 > $R/wgrad_counters.txt 2>&1
fi
# library-dispatch audit: the kernels of each traced configuration that are not fn2:: ones
for t in bench bench2 train; do
  f=$(find $R/$t -name "*_kernel_stats.csv" | head -1)
  [ -n "$f" ] && python scripts/nonfn2_kernels.py $f > $R/${t}_nonfn2.txt 2>&1
done
tail -c 300 $R/bench.json
