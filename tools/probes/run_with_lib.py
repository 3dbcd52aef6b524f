"""Run a repo script on an alternate build of the library (tools/probes/build_alt.py):

    python tools/probes/run_with_lib.py <name under tools/probes/_build | product> <script.py> [args...]

The alternate .so is bound with the product's ctypes signatures and installed as lib._lib before the script starts, so bench.py /
tests/bench_kernels.py / tools run unchanged (in-situ A/B of lab kernels). Lab use only: nothing in the product imports this."""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import lib as L  # noqa: E402


def main():
    name, script = sys.argv[1], sys.argv[2]
    L.load()
    if name != "product":
        alt = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "_build", name, "libomnivggt_hip.so"))
        for sym, (res, a) in L.SYMBOLS.items():
            f = getattr(alt, sym)
            f.restype, f.argtypes = res, a
        L._lib = alt
        sys.stderr.write("[run_with_lib] %s on %s\n" % (script, name))
    sys.argv = [script] + sys.argv[3:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
