export TMPDIR=/tmp
mkdir -p gpurun_out/r04t
timeout 600 python scripts/deconv_bench.py --net C --iters 30 2>&1 | grep -v amdgpu.ids > gpurun_out/r04t/deconv_C.txt
python - <<'P'
import re
best={}
for l in open('gpurun_out/r04t/deconv_C.txt'):
    if l.startswith('deconv'): name=l.split()[0]; print(l.strip()[:200])
    m=re.search(r'variant\s+(\d+):\s+([\d.]+) us', l)
    if m:
        t=float(m.group(2))
        if name not in best or t<best[name][0]: best[name]=(t,int(m.group(1)))
print(best)
P
