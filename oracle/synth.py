"""Synthetic weights/inputs: moved to wav2lip_amd/synthetic.py (bench.py must not import from oracle/ outside its cpu_baseline
leg); re-exported here for the tests and golden generators.  TEST INFRASTRUCTURE."""
from wav2lip_amd.synthetic import *  # noqa: F401,F403
from wav2lip_amd.synthetic import _rng  # noqa: F401
