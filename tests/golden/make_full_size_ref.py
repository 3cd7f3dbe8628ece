"""Writes tests/golden/full_size_ref_v1.json: the digests of full_size_scenario.py run through oracle/_ref, i.e. through the reference's
own kernel sources (needs /root/reference; run in the build container).   python tests/golden/make_full_size_ref.py"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from kintinuous_amd import synth
from oracle import oracle as O, ref as R
from oracle.oracle import OIntr
from full_size_scenario import scenario
R.build(); R.lib(); O.build(); O.lib()
cam = synth.Camera.scaled(1)
_, frames, _, _ = synth.sequence("orbit", 3, cam, 1234)
filtered = [O.bilateral_filter(d) for d, _ in frames]
out = scenario(R, OIntr, O.mat33_inverse, filtered)
json.dump(out, open(os.path.join(HERE, "full_size_ref_v1.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
