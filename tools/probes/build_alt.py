"""Alternate builds of the product library for in-process A/B runs: the product sources compiled with extra -D flags into
tools/probes/_build/<name>/libomnivggt_hip.so (git-ignored; travels to the GPU box with the snapshot).

    python tools/probes/build_alt.py name=-DFLAG=VALUE[,-DFLAG2=...] ...
    e.g. python tools/probes/build_alt.py pipe1=-DOVG_ATTN_PIPE_LOOP=1 pipe0=-DOVG_ATTN_PIPE_LOOP=0

Only ovg_attn.hip is recompiled per variant when every flag names an OVG_ATTN_* macro (the other objects are shared)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import build as B  # noqa: E402

OUT = os.path.join(ROOT, "tools", "probes", "_build")


def cc(src, obj, extra):
    if os.path.exists(obj) and not extra:
        return obj
    cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(B.CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return obj


def main():
    os.makedirs(os.path.join(OUT, "common"), exist_ok=True)
    for spec in sys.argv[1:]:
        name, flags = spec.split("=", 1)
        flags = flags.split(",")
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        attn_only = all(f.startswith("-DOVG_ATTN_") for f in flags)

        def one(src):
            if attn_only and src != "ovg_attn.hip":
                return cc(src, os.path.join(OUT, "common", src.replace(".hip", ".o")), [])
            return cc(src, os.path.join(d, src.replace(".hip", ".o")), flags)
        with ThreadPoolExecutor(max_workers=4) as ex:
            objs = list(ex.map(one, B.SOURCES))
        so = os.path.join(d, "libomnivggt_hip.so")
        r = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        print(name, "->", os.path.relpath(so, ROOT))


if __name__ == "__main__":
    main()
