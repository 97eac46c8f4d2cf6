# FlowNet2 / FlowNetC step time under the routing toggles (own kernels vs library for the deconvolution GEMMs and the small 3x3 layers)
for cfg in "--net 2 --batch 1 --height 448 --width 1024" "--net 2 --batch 4 --height 384 --width 768" ""; do
  for env in "FN2_X=0" "FN2_DECONV_GEMM=lib" "FN2_CONV_SMALL=lib" "FN2_DECONV_GEMM=lib FN2_CONV_SMALL=lib"; do
    echo "== [$cfg] $env"
    env $env python bench.py $cfg --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_p10_p50_p90'])"
  done
done
