# interleaved A/B of the forward step: own GEMM / 1x1 kernels against the library routes (same process order twice)
one() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_p10_p50_p90'])"; }
for r in 1 2; do
echo "own gemm + own 1x1:"; one
echo "lib gemm + lib 1x1:"; FN2_DECONV_GEMM=lib FN2_CONV_1X1=0 one
done
echo "FlowNet2 own:"; one --net 2 --batch 4 --height 384 --width 768 --steps 30
echo "FlowNet2 lib:"; FN2_DECONV_GEMM=lib FN2_CONV_1X1=0 one --net 2 --batch 4 --height 384 --width 768 --steps 30
