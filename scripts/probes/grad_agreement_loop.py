"""Runs tests/test_gpu_parity.py::test_training_gradients_fused_path_matches_stock_ops 14 times in one process and prints its agreement
figures.  Finding (r01): two modes, ~7e-7 relative L2 over all gradients, or 4.7e-6 with deconv2.w at 5e-4 -- the library picks a
different kernel for one of the two graphs in some runs; the test's criteria hold in both with two orders of magnitude of margin."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
for i in range(14):
    try:
        T.test_training_gradients_fused_path_matches_stock_ops()
    except AssertionError as e:
        print("FAIL", e)
