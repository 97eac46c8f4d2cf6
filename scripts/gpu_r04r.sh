set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04r
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/r04r/pytest.txt 2>&1
tail -8 gpurun_out/r04r/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r04r/bench.err > gpurun_out/r04r/bench.json; tail -c 1200 gpurun_out/r04r/bench.json | head -c 1200; echo
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04r/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config'].get('library_conv_fallbacks'))
for k,v in d['extra'].items(): print(k, v['value'], v['ms_per_step'])
P
