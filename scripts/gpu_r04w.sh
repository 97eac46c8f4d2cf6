export TMPDIR=/tmp
mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_conv_plane.py tests/test_tconv.py tests/test_conv_mfma.py -m gpu -q -x 2>&1 | tail -3
bash scripts/train_trace.sh r04w > gpurun_out/r04w_train_kernels.txt 2>&1
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r04w/train/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
stem=[r for r in rows if 'conv_k7s2_relu' in r['Name']]
steps=int(stem[0]['Calls'])
for r in rows:
    n=r['Name']; us=float(r['TotalDurationNs'])/1e3/steps
    if 'pack' in n or 'pad_width' in n:
        print("%8.1f us/step %6.1f calls/step  %s" % (us, int(r['Calls'])/steps, n[:130]))
P
head -3 gpurun_out/r04w_train_kernels.txt
