"""Runs the randomized oracle-vs-reference sweeps of tests/test_oracle_vs_ref.py over a range of seeds (the pytest run keeps 16 + 12 + 8 + 8 + 8 + 6 of
them): python tests/tools/deep_pin.py <first seed> <last seed + 1>  (KT_PIN_FROM=k skips the first k families).  Needs /root/reference (oracle/_ref)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
import test_oracle_vs_ref as T
from oracle import oracle as O, ref as R
R.build(); R.lib(); O.build(); O.lib()
bad = []
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    for fn in (T.test_randomized_sweep, T.test_randomized_sweep_image_and_rgbd, T.test_random_state_integrate, T.test_random_state_raycast_and_extract, T.test_perturbed_maps_icp, T.test_noise_images_rgbd)[int(os.environ.get("KT_PIN_FROM", 0)):]:
        try:
            fn(O, R, seed)
        except AssertionError as e:
            bad.append((fn.__name__, seed, str(e)[:200]))
            print("FAIL", fn.__name__, seed, str(e)[:300].replace("\n", " "))
        except Exception as e:
            print("ERROR", fn.__name__, seed, repr(e)[:200])
print("seeds", lo, hi, "failures", len(bad))
