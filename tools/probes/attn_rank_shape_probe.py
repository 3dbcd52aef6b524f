"""Per-rank global-attention launch of the 8-GPU view-sharded run (head-parallel form), timed on one GPU: 16 (source rank, head) entries of
10 992 query rows each against 8 key segments of 10 992 keys (kv_heads = 2) -- the shape every rank runs 24 times per forward at 64 views."""
import sys, statistics, torch
sys.path.insert(0, "/root/repo")
from omnivggt_official_amd import ops
dt, DEV = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(0)
# usage: attn_rank_shape_probe.py [ranks heads_per_launch views_per_rank]   (default 8 2 8; 4 ranks: 4 4 16; 2 ranks, one of two head groups: 2 4 32)
W, gs, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) * 1374) if len(sys.argv) > 3 else (8, 2, 8 * 1374)
q, _, _ = ops.alloc_qkv(W * gs, n, n, dt, DEV)
q[:, :n] = (torch.randn(W * gs, n, 64, generator=g) * 1.3).to(dt).to(DEV)
segs = []
for r in range(W):
    _, k, vt = ops.alloc_qkv(gs, 64, n, dt, DEV)
    k[:, :n] = torch.randn(gs, n, 64, generator=g).to(dt).to(DEV)
    ops.set_vt(vt, torch.randn(gs, 64, n, generator=g).to(dt))
    segs.append((k, vt, n))
flop = 4.0 * W * gs * n * (W * n) * 64
out = torch.empty(W * gs, q.shape[1], 64, device=DEV, dtype=dt)
ref = None
for variant, sp in ((50, 1), (0, 0), (50, 4), (57, 1), (57, 3), (57, 4), (57, 5), (57, 6), (57, 8)):
    plan = ops.attn_plan(W * gs, n, [n] * W, dt, variant, sp, nq_pad=q.shape[1])
    ws = ops.alloc_split_ws(plan, DEV) if plan["splits"] > 1 else None
    f = lambda: ops.flash_attn(q, segs, n, dt, out=out, variant=variant, kv_heads=gs, head_major=True, kv_splits=sp, split_ws=ws)
    f(); torch.cuda.synchronize()
    if ref is None: ref = out.float().clone()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    print("variant %d kv_splits %d (-> %d, q tile %d): %.3f ms  %.1f TFLOP/s  %.1f%% of 2.5PF  max-rel vs first %.1e" % (variant, sp, plan["splits"], plan["q_tile"], ms, flop / ms / 1e9, flop / ms / 1e9 / 25, err), flush=True)
