"""Interleaved A/B micro-benchmarks of the hot kernels on the bench shapes (GPU box tool).

    python tests/bench_kernels.py attn [--views 8 16] [--variants 1 6 8 21 25] [--rounds 5]
    python tests/bench_kernels.py gemm [--views 8]

Variants are interleaved inside one process (round-robin, median over rounds) on random
data, and every variant's output is checked against variant 1 (the baseline kernel, itself
verified against torch in gpu_selftest.py) before it is timed."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import lib as L, ops  # noqa: E402

DEV = "cuda"


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def attn(args):
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    out = {}
    for mode in args.modes:
        for S in args.views:
            if mode == "global":
                BH, n = 16, S * 1374
            else:
                BH, n = S * 16, 1374
            q, k, vt = ops.alloc_qkv(BH, n, n, dt, DEV)
            q[:, :n] = (torch.randn(BH, n, 64, generator=g) * 1.3).to(dt).to(DEV)
            k[:, :n] = torch.randn(BH, n, 64, generator=g).to(dt).to(DEV)
            vt[:, :, :n] = torch.randn(BH, 64, n, generator=g).to(dt).to(DEV)
            flop = 4.0 * BH * n * n * 64
            ref = ops.flash_attn(q, [(k, vt, n)], n, dt, variant=1).float()
            outs = {}
            for v in args.variants:
                o = torch.empty((BH // 16) * n, 1024, device=DEV, dtype=dt)
                ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v)
                err = float((o.float() - ref).abs().max() / ref.abs().max())
                outs[v] = (o, err)
            times = {v: [] for v in args.variants}
            iters = max(2, int(args.target_ms / max(1e-3, flop / 500e12 * 1e3)))
            for _ in range(args.rounds):
                for v in args.variants:
                    o = outs[v][0]
                    times[v].append(timed(lambda: ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v), iters))
            for v in args.variants:
                ms = statistics.median(times[v])
                tf = flop / ms / 1e9
                print("attn %-6s S=%-3d N=%-6d variant=%d: median %.3f ms (min %.3f)  %.1f TFLOP/s  %.1f%% of 2.5PF  err_vs_v1=%.2e"
                      % (mode, S, n, v, ms, min(times[v]), tf, tf / 25.0, outs[v][1]), flush=True)
                out["attn_%s_S%d_v%d" % (mode, S, v)] = {"ms": ms, "tflops": tf, "err": outs[v][1]}
    return out


def gemm(args):
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    out = {}
    for S in args.views:
        M = S * 1374
        for nm, N, K, epi in (("qkv", 3072, 1024, "qkv"), ("proj", 1024, 1024, L.EPI_RES), ("fc1", 4096, 1024, L.EPI_GELU), ("fc2", 1024, 4096, L.EPI_RES)):
            x = torch.randn(M, K, generator=g).to(dt).to(DEV)
            w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(DEV)
            b = torch.randn(N, generator=g).to(DEV)
            if epi == "qkv":
                q, k, vt = ops.alloc_qkv(16, M, M, dt, DEV)
                qn = [torch.ones(64, device=DEV), torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros(64, device=DEV)]
                from omnivggt_official_amd.aggregator import make_rope_tables
                rope = make_rope_tables(38, DEV)
                fn = lambda: ops.qkv(x, w, b, M, dt, q, k, vt, qk_norm=qn, rope=rope)
            elif epi == L.EPI_RES:
                res = torch.randn(M, N, generator=g).to(DEV)
                gam = torch.ones(N, device=DEV)
                y = torch.empty(M, N, device=DEV)
                fn = lambda: ops.linear(x, w, b, dt, epilogue=L.EPI_RES, out=y, res=res, gamma=gam)
            else:
                y = torch.empty(M, N, device=DEV, dtype=dt)
                fn = lambda: ops.linear(x, w, b, dt, epilogue=epi, out=y)
            fn()
            combos = [(gm, ml) for gm in args.tile_groups for ml in args.mainloops]
            if args.ablate:
                # 2/3 = 128^2 LDS-DMA loop without loads / without MFMAs; 5/6 = the same for the 256^2 ping-pong loop (RES epilogue only)
                combos = [(8, 7), (8, 1), (8, 2), (8, 3), (4, 4), (4, 5), (4, 6)]
            ts = {c: [] for c in combos}
            for _ in range(args.rounds):
                for gm, ml in combos:                  # interleaved A/B of the tuning knobs
                    L.load().ovg_debug_set(0, gm)
                    L.load().ovg_debug_set(2, gm if ml >= 4 else 4)     # 256^2 kernels: their own group size
                    L.load().ovg_debug_set(1, ml)
                    ts[(gm, ml)].append(timed(fn, 20))
            L.load().ovg_debug_set(0, 8)
            L.load().ovg_debug_set(1, 0)
            L.load().ovg_debug_set(2, 4)
            for gm, ml in combos:
                ms = statistics.median(ts[(gm, ml)])
                tf = 2.0 * M * N * K / ms / 1e9
                print("gemm %-5s S=%d M=%d N=%d K=%d group=%d mainloop=%d: median %.3f ms  %.1f TFLOP/s (%.1f%% of 2.5PF)"
                      % (nm, S, M, N, K, gm, ml, ms, tf, tf / 25.0), flush=True)
                out["gemm_%s_S%d_g%d_ml%d" % (nm, S, gm, ml)] = {"ms": ms, "tflops": tf}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "all"])
    ap.add_argument("--views", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--variants", type=int, nargs="+", default=[1, 6, 8, 21, 25], help="see dispatch16 in ovg_attn.hip")
    ap.add_argument("--modes", nargs="+", default=["global", "frame"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tile-groups", type=int, nargs="+", default=[8])
    ap.add_argument("--mainloops", type=int, nargs="+", default=[0], help="0 = automatic, 7 = 128^2 register-staged, 1 = 128^2 LDS-DMA, 4 = 256^2 ping-pong")
    ap.add_argument("--ablate", action="store_true", help="GEMM: also time the LDS-DMA loop without loads / without MFMAs")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--target-ms", type=float, default=20.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    L.require_gpu()
    res = {}
    if args.what in ("attn", "all"):
        res.update(attn(args))
    if args.what in ("gemm", "all"):
        res.update(gemm(args))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
