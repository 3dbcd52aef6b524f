"""Run bench.py against an EXPERIMENTAL library build (tools/lab/build_lab.py) instead of the product library:

    python tools/lab/bench_with_lab.py <experiment> [bench.py arguments ...]

In-situ A/B of a lab variant (the whole aggregator forward, clocks and caches as in the real run). The product library and its
build stamp are not touched; the JSON line gains no field -- the caller labels the runs."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import build as B, lib as L  # noqa: E402

name = sys.argv[1]
L.LIB_PATH = os.path.join(HERE, "_build", name, "libomnivggt_hip.so")
assert os.path.exists(L.LIB_PATH), L.LIB_PATH
B.is_current = lambda: True                     # the lab build has no stamp; never rebuild the product from here
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
