"""Interleaved A/B micro-benchmarks of the hot kernels on the bench shapes (GPU box tool).

    python tests/bench_kernels.py attn [--views 8 16] [--variants 1 0 50] [--rounds 5]
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
            ops.set_vt(vt, torch.randn(BH, 64, n, generator=g).to(dt))
            flop = 4.0 * BH * n * n * 64
            ref = ops.flash_attn(q, [(k, vt, n)], n, dt, variant=1).float()
            outs = {}
            combos = [(v, s) for v in args.variants for s in args.kv_splits]      # kv_splits: 1 = never, 0 = the library's plan, 2..8 forced
            for v, sp in combos:
                o = torch.empty((BH // 16) * n, 1024, device=DEV, dtype=dt)
                plan = ops.attn_plan(BH, n, [n], dt, v, sp, nq_pad=q.shape[1])
                ws = ops.alloc_split_ws(plan, DEV) if sp != 1 else None
                ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v, kv_splits=sp, split_ws=ws)
                err = float((o.float() - ref).abs().max() / ref.abs().max())
                outs[(v, sp)] = (o, err, ws, plan["splits"] if sp != 1 else 1)
            times = {c: [] for c in combos}
            iters = max(2, int(args.target_ms / max(1e-3, flop / 500e12 * 1e3)))
            for _ in range(args.rounds):
                for v, sp in combos:
                    o, _, ws, _ = outs[(v, sp)]
                    times[(v, sp)].append(timed(lambda: ops.flash_attn(q, [(k, vt, n)], n, dt, out=o, variant=v, kv_splits=sp, split_ws=ws), iters))
            for v, sp in combos:
                ms = statistics.median(times[(v, sp)])
                tf = flop / ms / 1e9
                print("attn %-6s S=%-3d N=%-6d variant=%d kv_splits=%d(->%d): median %.3f ms (min %.3f)  %.1f TFLOP/s  %.1f%% of 2.5PF  err_vs_v1=%.2e"
                      % (mode, S, n, v, sp, outs[(v, sp)][3], ms, min(times[(v, sp)]), tf, tf / 25.0, outs[(v, sp)][1]), flush=True)
                out["attn_%s_S%d_v%d_s%d" % (mode, S, v, sp)] = {"ms": ms, "tflops": tf, "err": outs[(v, sp)][1]}
    return out


def gemm(args):
    """Interleaved A/B of the GEMM workgroup tiles (ovg_linear_params.tile) on the block's four shapes at the bench's M,
    plus square STORE problems (4096^3 / 8192^3, the guide's reference shapes) that isolate the main loop from the fused
    epilogues. Every tile's output is compared with tile 1 (128 x 128, itself verified against torch in gpu_selftest.py)."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    out = {}
    shapes = []
    for S in args.views:
        M = S * 1374
        shapes += [("qkv", M, 3072, 1024, "qkv"), ("proj", M, 1024, 1024, L.EPI_RES), ("fc1", M, 4096, 1024, L.EPI_GELU), ("fc2", M, 1024, 4096, L.EPI_RES)]
    if args.no_qkv:
        shapes = [sh for sh in shapes if sh[4] != "qkv"]
    for n in args.square:
        shapes.append(("sq%d" % n, n, n, n, L.EPI_STORE))
    for nm, M, N, K, epi in shapes:
        x = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        if args.rotate > 1:
            # in the forward a block's weights are touched once per forward (1.9 GB of bf16 weights cycle through the caches): time the GEMM on
            # `rotate` copies of W in turn, so that every launch streams its weights from HBM like the in-situ launch does
            ws = [w] + [w.clone() for _ in range(args.rotate - 1)]
            turn = [0]

            class _Rot:
                def data_ptr(self_inner):
                    turn[0] = (turn[0] + 1) % len(ws)
                    return ws[turn[0]].data_ptr()

                def __getattr__(self_inner, k):
                    return getattr(ws[0], k)
            w = _Rot()
        if epi == "qkv":
            q, k, vt = ops.alloc_qkv(16, M, M, dt, DEV)
            qn = [torch.ones(64, device=DEV), torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros(64, device=DEV)]
            from omnivggt_official_amd.aggregator import make_rope_tables
            rope = make_rope_tables(38, DEV)
            fn = lambda t: ops.qkv(x, w, b, M, dt, q, k, vt, qk_norm=qn, rope=rope, tile=t)
            result = lambda: torch.cat([q.flatten(), k.flatten(), vt.flatten()]).float()
        elif epi == L.EPI_RES:
            res = torch.randn(M, N, generator=g).to(DEV)
            gam = torch.ones(N, device=DEV)
            y = torch.empty(M, N, device=DEV)
            fn = lambda t: ops.linear(x, w, b, dt, epilogue=L.EPI_RES, out=y, res=res, gamma=gam, tile=t)
            result = lambda: y.float().clone()
        else:
            y = torch.empty(M, N, device=DEV, dtype=dt)
            fn = lambda t: ops.linear(x, w, b, dt, epilogue=epi, out=y, tile=t)
            result = lambda: y.float().clone()
        fn(1)
        ref = result()
        errs = {}
        for t in args.tiles:
            fn(t)
            errs[t] = float((result() - ref).abs().max() / ref.abs().max())
        ts = {t: [] for t in args.tiles}
        flop = 2.0 * M * N * K
        iters = max(3, int(args.target_ms / max(1e-3, flop / 800e12 * 1e3)))
        for _ in range(args.rounds):
            for t in args.tiles:                       # interleaved A/B
                ts[t].append(timed(lambda: fn(t), iters))
        for t in args.tiles:
            ms = statistics.median(ts[t])
            tf = flop / ms / 1e9
            print("gemm %-7s M=%-6d N=%-5d K=%-5d tile=%d: median %.4f ms (min %.4f)  %.1f TFLOP/s (%.1f%% of 2.5PF)  err_vs_tile1=%.2e"
                  % (nm, M, N, K, t, ms, min(ts[t]), tf, tf / 25.0, errs[t]), flush=True)
            out["gemm_%s_M%d_tile%d" % (nm, M, t)] = {"ms": ms, "tflops": tf, "err": errs[t]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "all"])
    ap.add_argument("--views", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--variants", type=int, nargs="+", default=[1, 0, 50], help="see dispatch16 in ovg_attn.hip (history variants need a -DOVG_AB_VARIANTS build)")
    ap.add_argument("--modes", nargs="+", default=["global", "frame"])
    ap.add_argument("--kv-splits", type=int, nargs="+", default=[1], help="attn: split-KV factors to compare (1 = off, 0 = library plan, 2..8 forced)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tiles", type=int, nargs="+", default=[1, 2], help="GEMM: ovg_linear_params.tile values to compare (1 = 128^2, 2 = 256^2 ping-pong, ...)")
    ap.add_argument("--square", type=int, nargs="*", default=[], help="GEMM: also time n^3 STORE problems (e.g. 4096 8192)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--target-ms", type=float, default=20.0)
    ap.add_argument("--no-qkv", action="store_true", help="GEMM: skip the QKV shape (lab tiles that exist for ovg_linear only)")
    ap.add_argument("--rotate", type=int, default=1, help="GEMM: cycle over this many copies of the weight matrix (weights from HBM, as in the forward)")
    ap.add_argument("--out", default="")
    ap.add_argument("--alt-lib", default="", help="name of an alternate build under tools/probes/_build/ (build_alt.py) to run on instead of the product library")
    args = ap.parse_args()
    L.require_gpu()
    if args.alt_lib:
        import ctypes
        alt = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "_build", args.alt_lib, "libomnivggt_hip.so"))
        for name, (res, a) in L.SYMBOLS.items():
            f = getattr(alt, name)
            f.restype, f.argtypes = res, a
        L.load()
        L._lib = alt
        print("running on the alternate build", args.alt_lib, flush=True)
    res = {}
    if args.what in ("attn", "all"):
        res.update(attn(args))
    if args.what in ("gemm", "all"):
        res.update(gemm(args))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
