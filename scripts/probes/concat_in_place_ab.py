import sys, os
sys.path.insert(0, os.getcwd())
flag = sys.argv[1] == "1"
sys.argv = ["bench.py", "--mode", "train", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
import flownet2_amd.nets as nets
nets.CONCAT_IN_PLACE_TRAINING[0] = flag
import bench
bench.main()
