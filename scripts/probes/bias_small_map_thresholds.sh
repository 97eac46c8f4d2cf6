python -m pytest tests/test_gpu_parity.py -q -k "bias" -x 2>&1 | tail -2
for v in "C >= 128 \&\& (long long)N * hw <= 20000" "C >= 64 \&\& (long long)N * hw <= 80000" "C >= 64 \&\& (long long)N * hw <= 200000"; do
  sed -i "s/return C >= [0-9]* && (long long)N \* hw <= [0-9]*;/return $v;/" flownet2_amd/csrc/bias_act.hip
  grep -n "static inline bool small_map" flownet2_amd/csrc/bias_act.hip
  python -m flownet2_amd.build > /dev/null 2>&1
  for i in 1 2; do python bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train ms', d['ms_per_step'])"; done
done
