# predict_flow forward (csrc/flow_head.hip): kernel durations per decoder level under rocprofv3 for several wave targets of the channel split (rebuilt on the box)
export TMPDIR=/tmp
for g in 4096 8192 16384 32768; do
  sed -i "s/blocks \* kHeadG \* nsplit < [0-9]*/blocks * kHeadG * nsplit < $g/" flownet2_amd/csrc/flow_head.hip
  python -m flownet2_amd.build > /dev/null 2>&1
  rm -rf /tmp/hv; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/hv -o hv -- python $GRAFT_REPO_ROOT/scripts/probes/head_microbench.py > /dev/null 2>&1)
  echo "wave target $g"
  python - <<EOF
import csv,glob,collections
f=glob.glob("/tmp/hv/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "conv3x3_c2" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:40], r["Grid_Size_X"], r.get("Grid_Size_Y",""))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0
for k,v in sorted(d.items()):
    v.sort(); print("  ",k, len(v), "median %.1f us"%v[len(v)//2]); tot+=v[len(v)//2]*(2 if len(v)>60 else 1)
print("   sum of medians %.1f us"%tot)
EOF
done
sed -i "s/blocks \* kHeadG \* nsplit < [0-9]*/blocks * kHeadG * nsplit < 4096/" flownet2_amd/csrc/flow_head.hip
