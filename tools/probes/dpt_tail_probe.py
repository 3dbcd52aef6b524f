"""Timing of the one-launch DPT output stage (ovg_dpt_tail) against the three launches it replaces, across library builds
(the product + tools/probes/_build/<name>, e.g. phase-skipping lab builds -DOVG_DT_SKIP=1|2|4), 8 views 296^2 -> 518^2.

    python tools/probes/dpt_tail_probe.py [--names dt_nomfma dt_nointerp dt_noprefetch] [--views 8]"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def load_variant(path):
    lib = C.CDLL(path)
    for name, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def timed(fn, iters=10, rounds=5):
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", nargs="*", default=[])
    ap.add_argument("--views", type=int, default=8)
    args = ap.parse_args()
    build = os.path.join(ROOT, "tools", "probes", "_build")
    libs = {"product": L.load()}
    libs.update({n: load_variant(os.path.join(build, n, "libomnivggt_hip.so")) for n in args.names})
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    n, H, OH, od = args.views, 296, 518, 4
    x = torch.randn(n, H, H, 128, generator=g).to(dt).to(DEV)
    w1 = torch.zeros(128, 1152, dtype=dt)
    w1[:32] = (torch.randn(32, 1152, generator=g) / 34.0).to(dt)
    w1 = w1.to(DEV)
    b1, w2, b2 = (torch.randn(32, generator=g) * 0.3).to(DEV), (torch.randn(od, 32, generator=g) * 0.2).to(DEV), (torch.randn(od, generator=g) * 0.1).to(DEV)
    pos = ((torch.randn(OH, 64, generator=g) * 0.1).to(DEV), (torch.randn(OH, 64, generator=g) * 0.1).to(DEV))

    def three():
        y = ops.upsample(x, OH, OH, dt, pos=pos)
        return ops.dpt_out(ops.conv(y, w1, b1, dt, 32, ksize=3, relu=True, out_f32=True), w2, b2, "inv_log")
    ref = three()
    print("three launches (upsample -> conv -> dpt_out): %.1f us" % timed(three))
    flop = 2.0 * n * OH * OH * 32 * 1152
    for nm, lib in libs.items():
        L._lib = lib
        out = ops.dpt_tail(x, OH, OH, dt, pos, w1, b1, w2, b2, "inv_log")
        err = float((out[0] - ref[0]).abs().max() / ref[0].abs().max())
        us = timed(lambda: ops.dpt_tail(x, OH, OH, dt, pos, w1, b1, w2, b2, "inv_log"))
        print("%-14s %8.1f us  (%.0f TFLOP/s of the 3x3 conv)  max-rel vs three launches %.2e" % (nm, us, flop / us / 1e6, err), flush=True)
    L._lib = libs["product"]


if __name__ == "__main__":
    main()
