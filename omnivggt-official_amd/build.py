"""Build libomnivggt_hip.so (gfx950) in-tree with hipcc.

`python -m omnivggt_official_amd.build` or `__graft_entry__.build()`.  hipcc
cross-compiles without a GPU; the .so lands next to this file so it travels
with the tree (git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libomnivggt_hip.so")
SOURCES = ["ovg_gemm.hip", "ovg_attn.hip", "ovg_elem.hip", "ovg_block.hip", "ovg_head.hip", "ovg_camhead.hip", "ovg_camtab.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# attention: no NaN can occur on valid inputs (masked scores are -inf, never inf-inf), and without
# this hipcc inserts a canonicalising v_max before every fmaxf on an MFMA output (64 VALU / tile)
EXTRA_FLAGS = {"ovg_attn.hip": ["-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files.append(os.path.join(HERE, "..", "include", "omnivggt_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def is_current():
    """True when the in-tree .so was built from exactly the current sources / header / flags."""
    stamp = OUT + ".stamp"
    return os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == _digest()


def have_hipcc():
    try:
        c = _hipcc()
    except RuntimeError:
        return False
    import shutil
    return os.path.isabs(c) or shutil.which(c) is not None


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link the shared library.
    Safe against concurrent callers (one process per GPU all finding a stale library at start-up): an exclusive file lock
    serialises them, the objects of each caller go to a private directory, and the library appears by an atomic rename --
    a reader never maps a half-written .so, and whoever comes second finds the stamp current and returns."""
    import fcntl
    import shutil
    import tempfile
    stamp = OUT + ".stamp"
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            dig = _digest()
            if not force and is_current():
                return OUT
            hipcc = _hipcc()
            os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
            objdir = tempfile.mkdtemp(prefix="obj.", dir=os.path.join(HERE, "build"))

            def cc(src):
                obj = os.path.join(objdir, src.replace(".hip", ".o"))
                cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
                return obj

            try:
                with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
                    objs = list(ex.map(cc, SOURCES))
                tmp_out = os.path.join(objdir, "libomnivggt_hip.so")
                r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_out, *objs], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("link failed:\n" + r.stderr[-4000:])
                if os.path.exists(stamp):
                    os.remove(stamp)                      # never a current stamp next to a library it does not describe
                os.replace(tmp_out, OUT)
                with open(stamp + ".tmp", "w") as fh:
                    fh.write(dig)
                os.replace(stamp + ".tmp", stamp)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
            return OUT
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
