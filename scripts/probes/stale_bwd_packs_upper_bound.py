import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--mode", "train", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
import flownet2_amd.functional as Fn
orig = Fn._cached
def fake(cache, key, w, make):
    if cache is Fn._PACKED_T:
        hit = cache.get(key)
        if hit is not None and hit[0]() is w:
            return hit[2]
    return orig(cache, key, w, make)
Fn._cached = fake
import bench
bench.main()
