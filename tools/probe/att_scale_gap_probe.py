"""How far apart may the site scales of the two key blocks be on the planes-only forward?  Runs
tests/test_planes_gpu.py::test_attention_fwd_planes_only_with_site_scales_far_apart over a range of gaps and prints the
measured errors (row 0 = the rows whose output is the small-magnitude block's alone)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_planes_gpu as T

print("   R   err / max|O|   err(row 0) / max|O[0]|   max|O[0]| / max|O|")
for R in (0, 4, 10, 13, 16, 18, 20, 22, 24, 28, 32, -10, -18, -24):
    e_all, e0, share = T._scale_gap_errors(16, 48, 40, 40, 100, R)
    print(f"{R:4d}   {e_all:10.3e}     {e0:10.3e}              {share:10.3e}")
