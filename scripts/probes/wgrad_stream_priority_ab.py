import sys, os
sys.path.insert(0, os.getcwd())
prio = int(sys.argv[1])
sys.argv = ["bench.py", "--mode", "train", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
import torch
import flownet2_amd.functional as Fn
try:
    st = torch.cuda.Stream(priority=prio)
    print("stream with priority", prio, "->", st.priority, file=sys.stderr)
    Fn._WGRAD_SIDE["streams"][torch.device("cuda", 0)] = st
except Exception as e:
    print("priority", prio, "refused:", e, file=sys.stderr)
import bench
bench.main()
