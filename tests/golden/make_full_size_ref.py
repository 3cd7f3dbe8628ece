"""Writes tests/golden/full_size_ref_v1.npz: the inputs of full_size_scenario.py and the digests of its outputs when run through oracle/_ref,
i.e. through the reference's own kernel sources (needs /root/reference; run in the build container).   python tests/golden/make_full_size_ref.py"""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from kintinuous_amd import synth
from oracle import oracle as O, ref as R
from oracle.oracle import OIntr
from full_size_scenario import make_inputs, scenario
R.build(); R.lib(); O.build(); O.lib()
g = make_inputs()
filtered = [O.bilateral_filter(g["depth%d" % k]) for k in range(3)]
out = scenario(R, g, OIntr, O.mat33_inverse, filtered)
np.savez_compressed(os.path.join(HERE, "full_size_ref_v1.npz"), digests=np.array(json.dumps(out, sort_keys=True)), **g)
print(json.dumps(out, indent=1, sort_keys=True))
