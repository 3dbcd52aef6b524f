"""In-process A/B of attention launches across library builds (the product + tools/probes/_build/*): interleaved timing on the same
tensors, bit comparison against the product, and the speculative-softmax fallback counter of every launch.

    python tools/probes/attn_ab_probe.py [--shapes s8 s16 rank8] [--rounds 4] [--names pipe1 pipe0]"""
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
        fn.restype = res
        fn.argtypes = args
    assert lib.ovg_abi_version() == L.ABI_VERSION
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["s8", "s16", "rank8"])
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--names", nargs="+", default=None)
    ap.add_argument("--variants", type=int, nargs="+", default=[0])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32x"], help="f32x: the split-f16 kernel on (hi, lo) plane pairs (single-segment shapes)")
    args = ap.parse_args()
    build = os.path.join(ROOT, "tools", "probes", "_build")
    names = args.names or sorted(d for d in os.listdir(build) if os.path.exists(os.path.join(build, d, "libomnivggt_hip.so")))
    libs = {"product": L.load()}
    libs.update({n: load_variant(os.path.join(build, n, "libomnivggt_hip.so")) for n in names})
    split = args.dtype == "f32x"
    dt = L.F32X if split else torch.bfloat16
    g = torch.Generator().manual_seed(0)
    for shape in args.shapes:
        if shape.startswith("rank"):            # per-rank launch of the head-parallel sharded form: W sources x 2 heads, W key segments
            W, gs, n = int(shape[4:]), 2, 8 * 1374
            BH, kv_heads, head_major = W * gs, gs, True
            nks = [n] * W
        else:
            S = int(shape[1:])
            BH, n, kv_heads, head_major, nks = 16, S * 1374, 0, False, [S * 1374]
            gs = BH
        q, _, _ = ops.alloc_qkv(BH, n, 64, dt, DEV)
        q32 = torch.randn(BH, n, 64, generator=g) * 1.3
        segs = []
        if split:
            h = ops.to_hilo(q32.to(DEV))
            q.hi[:, :n], q.lo[:, :n] = h.hi, h.lo
        else:
            q[:, :n] = q32.to(dt).to(DEV)
        for nk in nks:
            _, k, vt = ops.alloc_qkv(gs, 64, nk, dt, DEV)
            k32, v32 = torch.randn(gs, nk, 64, generator=g), torch.randn(gs, 64, nk, generator=g)
            if split:
                kh, vh = ops.to_hilo(k32.to(DEV)), ops.to_hilo(v32.to(DEV))
                k.hi[:, :nk], k.lo[:, :nk] = kh.hi, kh.lo
                ops.set_vt(vt.hi, vh.hi)
                ops.set_vt(vt.lo, vh.lo)
            else:
                k[:, :nk] = k32.to(dt).to(DEV)
                ops.set_vt(vt, v32.to(dt))
            segs.append((k, vt, nk))
        flop = 4.0 * BH * n * sum(nks) * 64
        cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
        for v in args.variants:
            outs, fb, times = {}, {}, {nm: [] for nm in libs}

            def run(nm, out):
                L._lib = libs[nm]
                plan = ops.attn_plan(BH, n, nks, dt, v, 1, nq_pad=q.shape[1])          # kv_splits = 1: the unsplit launch
                return ops.flash_attn(q, segs, n, dt, out=out, variant=v, kv_heads=kv_heads, head_major=head_major, kv_splits=0 if split else 1,
                                      fallback_count=None if split else cnt), plan
            for nm in libs:
                cnt.zero_()
                o, plan = run(nm, None)
                torch.cuda.synchronize()
                outs[nm], fb[nm] = o, int(cnt.item())
            iters = max(2, int(100.0 / max(1e-3, flop / 1200e12 * 1e3)))
            for r in range(args.rounds + 1):
                for nm in libs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        run(nm, outs[nm])
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[nm].append(e0.elapsed_time(e1) / iters)
            base = statistics.median(times["product"])
            for nm in libs:
                ms = statistics.median(times[nm])
                valid = (lambda t: t[:, :n]) if head_major else (lambda t: t)      # head-major outputs carry uninitialised padding rows
                bits = lambda t: (t.planes if split else valid(t)).contiguous().view(torch.int16)
                same = bool(torch.equal(bits(outs[nm]), bits(outs["product"])))
                print("%-6s variant %-2d %-8s median %8.4f ms  %7.1f TFLOP/s  vs product %+6.2f%%  fallback workgroups %d  bits==product %s  (q tile %d)"
                      % (shape, v, nm, ms, flop / ms / 1e9, (base / ms - 1) * 100, fb[nm], same, plan["q_tile"]), flush=True)
        L._lib = libs["product"]
        del q, segs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
