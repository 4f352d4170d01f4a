"""bench.py with a tuning knob set first:  python tools/bench_knob.py <which> <value> [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-mixture-vae_amd"))
from dmvae_hip import _lib as L
L.check(L.lib.dmvae_debug_set_knob(int(sys.argv[1]), int(sys.argv[2])))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name="__main__")
