"""Where a GEMM workgroup's time goes: per-workgroup wall-clock stamps (entry / first k-stage visible / main loop done / epilogue done)
from a lab build of the product library (-DOVG_GEMM_TIMELINE, tools/probes/build_alt.py tl=-DOVG_GEMM_TIMELINE), on the block's four
GEMM shapes at the bench's M, with and without a start-up stagger of the first round of workgroups.

    python tools/probes/build_alt.py tl=-DOVG_GEMM_TIMELINE
    python tools/probes/gemm_timeline.py [--views 8 64] [--tiles 1 2] [--stagger 0 450]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
from omnivggt_official_amd import lib as L, ops  # noqa: E402
from attn_ab_probe import load_variant  # noqa: E402

DEV = "cuda"
TICK_US = 0.01      # wall_clock64: 100 MHz


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pct(a, q):
    return float(np.percentile(a, q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, nargs="+", default=[8, 64])
    ap.add_argument("--tiles", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--stagger", type=int, nargs="+", default=[0, 450], help="ticks (10 ns) per stagger step, 8 steps")
    ap.add_argument("--name", default="tl")
    ap.add_argument("--square", type=int, nargs="*", default=[4096])
    args = ap.parse_args()
    L.require_gpu()
    prod = L.load()
    lab = load_variant(os.path.join(ROOT, "tools", "probes", "_build", args.name, "libomnivggt_hip.so"))
    lab.ovg_lab_gemm_timeline.restype = C.c_int
    lab.ovg_lab_gemm_timeline.argtypes = [C.c_void_p, C.c_int]
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    shapes = []
    for S in args.views:
        M = S * 1374
        shapes += [("qkv", M, 3072, 1024, "qkv"), ("proj", M, 1024, 1024, L.EPI_RES), ("fc1", M, 4096, 1024, L.EPI_GELU), ("fc2", M, 1024, 4096, L.EPI_RES)]
    for n in args.square:
        shapes.append(("sq%d" % n, n, n, n, L.EPI_STORE))
    for nm, M, N, K, epi in shapes:
        x = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        if epi == "qkv":
            q, k, vt = ops.alloc_qkv(16, M, M, dt, DEV)
            qn = [torch.ones(64, device=DEV), torch.zeros(64, device=DEV), torch.ones(64, device=DEV), torch.zeros(64, device=DEV)]
            from omnivggt_official_amd.aggregator import make_rope_tables
            rope = make_rope_tables(38, DEV)
            fn = lambda t: ops.qkv(x, w, b, M, dt, q, k, vt, qk_norm=qn, rope=rope, tile=t)
        elif epi == L.EPI_RES:
            res = torch.randn(M, N, generator=g).to(DEV)
            gam = torch.ones(N, device=DEV)
            y = torch.empty(M, N, device=DEV)
            fn = lambda t: ops.linear(x, w, b, dt, epilogue=L.EPI_RES, out=y, res=res, gamma=gam, tile=t)
        else:
            y = torch.empty(M, N, device=DEV, dtype=dt)
            fn = lambda t: ops.linear(x, w, b, dt, epilogue=epi, out=y, tile=t)
        flop = 2.0 * M * N * K
        for t in args.tiles:
            tb = 128 if t == 1 else 256
            nwg = ((M + tb - 1) // tb) * (N // tb)
            L._lib = prod
            fn(t)
            ms_prod = statistics.median(timed(lambda: fn(t), 5) for _ in range(5))
            for stg in args.stagger:
                L._lib = lab
                buf = torch.zeros(nwg * 16, dtype=torch.int64, device=DEV)
                assert lab.ovg_lab_gemm_timeline(buf.data_ptr(), stg) == 0
                fn(t)
                ms = statistics.median(timed(lambda: fn(t), 5) for _ in range(5))
                buf.zero_()
                torch.cuda.synchronize()
                fn(t)
                torch.cuda.synchronize()
                a = buf.cpu().numpy().reshape(nwg, 2, 8).astype(np.int64)
                assert lab.ovg_lab_gemm_timeline(0, 0) == 0
                t0 = a[:, :, 0].min()
                w0, w1 = a[:, 0, :4] - t0, a[:, 1, :4] - t0            # first / last wave stamps, ticks from the first entry
                span = max(w0[:, 3].max(), w1[:, 3].max()) * TICK_US
                pro = (w0[:, 1] - w0[:, 0]) * TICK_US
                loop = (w0[:, 2] - w0[:, 1]) * TICK_US
                ep0 = (w0[:, 3] - w0[:, 2]) * TICK_US
                ep1 = (w1[:, 3] - w1[:, 2]) * TICK_US
                tot = (np.maximum(w0[:, 3], w1[:, 3]) - w0[:, 0]) * TICK_US
                mhz = (a[:, 0, 7] - a[:, 0, 6]) / np.maximum(1, (a[:, 0, 3] - a[:, 0, 0])) * 100.0
                # how many workgroups sit in their epilogue at the same time (sampled every 0.5 us)
                ts = np.arange(0, int(span / TICK_US), 50)
                e_beg, e_end = np.minimum(w0[:, 2], w1[:, 2]), np.maximum(w0[:, 3], w1[:, 3])
                conc = np.array([np.count_nonzero((e_beg <= tt) & (e_end > tt)) for tt in ts[:: max(1, len(ts) // 400)]])
                # gaps between consecutive workgroups on the same CU (hw id word: cu / se / xcc identify it)
                cu_key = ((a[:, 0, 4] >> 8) & 0xff) | ((a[:, 0, 5] & 0xf) << 8)
                gaps = []
                for key in np.unique(cu_key)[:64]:
                    idx = np.where(cu_key == key)[0]
                    if len(idx) > 1:
                        o = idx[np.argsort(w0[idx, 0])]
                        ends = np.maximum(w0[o, 3], w1[o, 3])
                        gaps.extend(((w0[o[1:], 0] - ends[:-1]) * TICK_US).tolist())
                print("tl %-5s M=%-6d N=%-5d K=%-5d tile=%d stagger=%-4d wgs=%-5d | product %.4f ms (%.0f TF) lab %.4f ms (%.0f TF) span %.1f us | per WG us: prologue %.2f/%.2f loop %.2f/%.2f "
                      "epilogue(w0) %.2f/%.2f epilogue(wlast) %.2f/%.2f total %.2f/%.2f (median/p90) | shader MHz %.0f | epilogue concurrency mean %.0f max %d | same-CU gap median %.2f us (n=%d)"
                      % (nm, M, N, K, t, stg, nwg, ms_prod, flop / ms_prod / 1e9, ms, flop / ms / 1e9, span, pct(pro, 50), pct(pro, 90), pct(loop, 50), pct(loop, 90),
                         pct(ep0, 50), pct(ep0, 90), pct(ep1, 50), pct(ep1, 90), pct(tot, 50), pct(tot, 90), float(np.median(mhz)), conc.mean(), conc.max(),
                         float(np.median(gaps)) if gaps else -1.0, len(gaps)), flush=True)
                # round structure: start times of the first 3 rounds on the busiest CU
                if stg == args.stagger[0]:
                    key = np.unique(cu_key)[0]
                    idx = np.where(cu_key == key)[0]
                    o = idx[np.argsort(w0[idx, 0])][:6]
                    print("   one CU, first workgroups (us from launch): " + "  ".join("[%.1f %.1f %.1f %.1f]" % tuple(w0[i] * TICK_US) for i in o), flush=True)
        L._lib = prod


if __name__ == "__main__":
    main()
