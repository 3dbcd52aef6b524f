"""Interleaved A/B of the global-attention launch of EXPERIMENTAL library builds (tools/lab/build_lab.py) on one GPU.

All variants are loaded into ONE process (separate dlopen handles) and timed round-robin on the same tensors, so box-to-box and
minute-to-minute drift cancels. Every variant's output is checked against the baseline kernel (variant 1 of the control build)
and compared BIT FOR BIT with the control build's output of the same launch: the order-pinned bodies keep the arithmetic and the
per-accumulator operation order of the shipped body, so anything but identical bits is a bug (a hazard, a missed wait).

    python tools/lab/run_attn_lab.py [--views 64 8] [--rounds 5] [--names control pipe_v1 ...]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def load_variant(path):
    lib = C.CDLL(path)
    for name, (res, args) in L.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    assert lib.ovg_abi_version() == L.ABI_VERSION
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, nargs="+", default=[64, 8])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--names", nargs="+", default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--mode", default="global", choices=["global", "frame"], help="global: 16 heads x (S x 1374) tokens; frame: S x 16 heads x 1374 tokens")
    ap.add_argument("--solo", action="store_true", help="time only the named builds (no control build beside them): for rocprofv3 passes, where "
                    "kernels of the same name and grid from two builds would be merged")
    ap.add_argument("--zero", action="store_true", help="zero-filled q / k / v: the same instruction stream at far lower switching power -- if the "
                    "kernel speeds up, its clock (the chip's power budget) is what bounds it, not its issue slots")
    ap.add_argument("--variants", type=int, nargs="+", default=[0], help="attention variants to time (0 = the launch plan; 50 = 256-row kernel, 57 = 512-row kernel without tail split)")
    args = ap.parse_args()
    build = os.path.join(HERE, "_build")
    names = args.names or sorted(d for d in os.listdir(build) if os.path.exists(os.path.join(build, d, "libomnivggt_hip.so")))
    if not args.solo:
        if "control" in names:
            names.remove("control")
        names = ["control"] + names
    libs = {n: load_variant(os.path.join(build, n, "libomnivggt_hip.so")) for n in names}

    def use(n):
        L._lib = libs[n]

    use(names[0])
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    for S in args.views:
        BH, n = (16, S * 1374) if args.mode == "global" else (S * 16, 1374)
        q, k, vt = ops.alloc_qkv(BH, n, n, dt, DEV)
        q[:, :n] = (torch.randn(BH, n, 64, generator=g) * 1.3).to(dt).to(DEV)
        k[:, :n] = torch.randn(BH, n, 64, generator=g).to(dt).to(DEV)
        ops.set_vt(vt, torch.randn(BH, 64, n, generator=g).to(dt))
        if args.zero:
            q.zero_(); k.zero_(); vt.zero_()
        flop = 4.0 * BH * n * n * 64
        ref = ops.flash_attn(q, [(k, vt, n)], n, dt, variant=1).float()
        combos = [(nm, v) for v in args.variants for nm in names]
        outs, errs = {}, {}
        for nm, v in list(combos):
            use(nm)
            o = torch.zeros((BH // 16) * n, 1024, device=DEV, dtype=dt)
            try:
                ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v)
            except L.OvgError as e:                     # a variant this build does not have (e.g. the lab kernels in the control build)
                combos.remove((nm, v))
                continue
            torch.cuda.synchronize()
            outs[(nm, v)] = o
            errs[(nm, v)] = float((o.float() - ref).abs().max() / ref.abs().max())
        cref = lambda v: outs[(names[0], v)] if (names[0], v) in outs else outs[(names[0], args.variants[0])]
        same = {c: bool(torch.equal(outs[c].view(torch.int16), cref(c[1]).view(torch.int16))) for c in combos}
        iters = max(2, int(200.0 / max(1e-3, flop / 1200e12 * 1e3)))
        times = {c: [] for c in combos}
        for r in range(args.rounds + 1):
            for nm, v in combos:
                use(nm)
                o = outs[(nm, v)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[(nm, v)].append(e0.elapsed_time(e1) / iters)
        base = statistics.median(times[(names[0], args.variants[0])]) if (names[0], args.variants[0]) in times else float("nan")
        for c in combos:
            ms = statistics.median(times[c])
            print(("ZERO DATA " if args.zero else "") + ("frame " if args.mode == "frame" else "") + "S=%-3d %-14s variant %-2d median %8.4f ms (min %8.4f)  %7.1f TFLOP/s  %5.1f%% of 2.5PF  vs control/plan %+6.2f%%  err_vs_baseline=%.2e  bits==control: %s"
                  % (S, c[0], c[1], ms, min(times[c]), flop / ms / 1e9, flop / ms / 1e9 / 25.0, (base / ms - 1) * 100, errs[c], same[c]), flush=True)
        del q, k, vt, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
